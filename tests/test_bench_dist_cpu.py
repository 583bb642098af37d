"""The N > 1 side of bench.py on the CPU: two processes over gloo run the bench's own sharding and reduction code
(bench.shard_seed / reduce_job / job_rates).  The data path has no collective (reads are independent, SURVEY.md
section 8e); what must be right is that ranks work on different shards, that the timed region is the MAX over ranks
and that the whole-job value sums every rank's units."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch.distributed as dist
    import bench
    import simbatch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = simbatch.SimRef(2_000_000, n_contigs=2, seed=5, threads=1)
    B = simbatch.SimBatch(ref, 8 + rank, 2000, "ont2d", seed=bench.shard_seed(rank), threads=1)   # ragged shards
    n_bases = int(B.read_off[-1])
    dt_local = 0.5 + 0.25 * rank
    dt, tot = bench.reduce_job(dt_local, [B.n_reads, n_bases, n_bases, rank], world)
    dist.barrier()
    q.put((rank, dt, tot.tolist(), B.read_seq[:64].tobytes(), B.n_reads, n_bases))
    dist.destroy_process_group()


def test_two_ranks_shard_and_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, dt0, tot0, head0, n0, b0), (r1, dt1, tot1, head1, n1, b1) = out
    assert head0 != head1                                   # different reads on the two ranks
    assert dt0 == dt1 == 0.75                                # MAX over ranks
    assert tot0 == tot1 == [n0 + n1, b0 + b1, b0 + b1, 1]    # SUM over ranks
    sys.path.insert(0, ROOT)
    import bench
    rps, gbps = bench.job_rates(dt0, np.array(tot0), steps=3)
    assert abs(rps - (n0 + n1) * 3 / 0.75) < 1e-9 and abs(gbps - (b0 + b1) * 3 / 0.75 / 1e9) < 1e-12


def test_single_rank_is_identity():
    sys.path.insert(0, ROOT)
    import bench
    dt, tot = bench.reduce_job(1.5, [4, 3, 2, 1], 1)
    assert dt == 1.5 and tot.tolist() == [4, 3, 2, 1]


def test_simulator_refuses_a_reference_without_room_for_the_read():
    """tools/simhits.c draws a read's locus by rejection; with no contig longer than twice the read it used to draw for ever (a GPU-box run
    was killed for silence over exactly that).  It now says so."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pytest
    import simbatch
    simbatch.build()
    ref = simbatch.SimRef(400_000, n_contigs=4, seed=3, threads=2, repeats=False)          # four contigs of 100 kbp
    with pytest.raises(ValueError, match="no contig"):
        simbatch.SimBatch(ref, 4, 60_000, "ont2d", seed=1, threads=2)
    B = simbatch.SimBatch(ref, 4, 20_000, "ont2d", seed=1, threads=2)                        # twice 20 kbp fits
    assert B.n_reads == 4
