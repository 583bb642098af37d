"""GPU parity of the hot path proper through the C-ABI (lamsa_hp_align_batch): chaining + gap-fill /
split extension on the MI355X against the oracle and -- through the oracle's pinned SAM -- the reference.
Integer / index work: result streams must be identical word for word."""
import numpy as np
import pytest

import goldenlib
import reflib

pytestmark = pytest.mark.gpu


def _handle(B, rt, over):
    from lamsa_amd import hp
    return hp.LamsaHp(hp.make_para(rt, **over), ref=(B.pac, B.l_pac, B.seq_off, B.seq_len))


@pytest.mark.parametrize("name", goldenlib.SCENARIOS)
def test_hip_read_kernel_matches_oracle_on_fixtures(name, tmp_path):
    ref, reads, args, _ = goldenlib.stage_scenario(name, str(tmp_path))
    rt, over = goldenlib.para_from_args(args)
    lp = reflib.lo_para(rt, **over)
    B = reflib.Batch(ref, reads, lp)
    want = reflib.oracle_streams(B, lp)
    h = _handle(B, rt, over)
    got, st = h.align_batch(B)
    h.close()
    assert [i for i in range(B.n_reads) if want[i] != got[i]] == []
    assert (st == 0).all()


def test_hip_batch_edge_cases(tmp_path):
    """Empty batch, single read, repeated runs of a resident batch, reads in a different order."""
    ref, reads, args, _ = goldenlib.stage_scenario("c2_pacbio", str(tmp_path))
    rt, over = goldenlib.para_from_args(args)
    lp = reflib.lo_para(rt, **over)
    B = reflib.Batch(ref, reads, lp)
    want = reflib.oracle_streams(B, lp)
    h = _handle(B, rt, over)
    got, st = h.align_batch(B.take([]))
    assert got == [] and len(st) == 0
    got, _ = h.align_batch(B.take([3]))
    assert got == [want[3]]
    perm = list(reversed(range(B.n_reads)))
    got, _ = h.align_batch(B.take(perm))
    assert got == [want[i] for i in perm]
    h.upload_batch(B)
    a = h.run_uploaded()[0]
    h.run_uploaded(fetch=False)
    b = h.run_uploaded()[0]
    assert a == want and b == want          # idempotent on a resident batch
    h.close()


def test_hip_rejects_malformed_batches(tmp_path):
    ref, reads, args, _ = goldenlib.stage_scenario("c1_perfect", str(tmp_path))
    rt, over = goldenlib.para_from_args(args)
    lp = reflib.lo_para(rt, **over)
    B = reflib.Batch(ref, reads, lp).take([0, 1])
    h = _handle(B, rt, over)
    bad = B.take([0, 1]); bad.h_chr = bad.h_chr.copy(); bad.h_chr[0] = 99
    with pytest.raises(RuntimeError):
        h.align_batch(bad)
    bad = B.take([0, 1]); bad.seed_id = bad.seed_id.copy(); bad.seed_id[1] = bad.seed_id[0]
    with pytest.raises(RuntimeError):
        h.align_batch(bad)
    h.close()
