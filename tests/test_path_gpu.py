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
    # seed geometry that is not the read's own (a stale or foreign hit stream): the kernels would cut read windows from it
    bad = B.take([0, 1]); bad.seed_all = bad.seed_all.copy(); bad.seed_all[0] += 7
    with pytest.raises(RuntimeError, match="seed_all"):
        h.align_batch(bad)
    bad = B.take([0, 1]); bad.last_len = bad.last_len.copy(); bad.last_len[1] += 1
    with pytest.raises(RuntimeError, match="last_len"):
        h.align_batch(bad)
    # a field the device keeps in fewer bits than the boundary's type: never wrapped -- that read comes back unaligned with
    # LAMSA_HP_ST_UNSUPPORTED, the rest of the batch is aligned as ever
    bad = B.take([0, 1]); bad.h_len_dif = bad.h_len_dif.copy(); bad.h_len_dif[0] = 300
    got, st = h.align_batch(bad)
    assert int(st[0]) == 4 and list(got[0]) == [4, 0, 0]
    assert int(st[1]) == 0 and got[1] == reflib.oracle_streams(B.take([1]), lp)[0]
    got, st = h.align_batch(B)                                    # the handle is still usable
    assert got == reflib.oracle_streams(B, lp) and (st == 0).all()
    h.close()


def test_hip_compact_cigar_form(tmp_path):
    """The compact boundary form (cig8: one byte per seed-CIGAR element, h_cig_off = NULL: offsets summed up on the device
    from the lengths) gives the streams of the word form; a length table that does not add up is refused."""
    from lamsa_amd import hp
    ref, reads, args, _ = goldenlib.stage_scenario("c3_ont", str(tmp_path))
    rt, over = goldenlib.para_from_args(args)
    lp = reflib.lo_para(rt, **over)
    B = reflib.Batch(ref, reads, lp)
    want = reflib.oracle_streams(B, lp)
    h = _handle(B, rt, over)
    Bc = hp.compact_batch(B)
    got, st = h.align_batch(Bc)
    assert got == want and (st == 0).all()
    Bp = hp.pinned_batch(Bc)
    got, st = h.align_batch(Bp)
    Bp.release()
    assert got == want and (st == 0).all()
    bad = hp.compact_batch(B); bad.cig8 = bad.cig8[:-3]
    with pytest.raises(RuntimeError, match="add up"):
        h.align_batch(bad)
    h.close()


def test_hip_batch_beyond_2_31_cigar_elements():
    """147 456 reads x 10 kbp in ONE batch: 2.3 G seed-CIGAR elements, more than 32-bit offsets can address.  The compact form of the
    boundary (one byte per element, offsets summed up on the device) takes it; every read comes back with status 0 and a sample
    equals the oracle word for word.  The word form with its 32-bit offsets is refused with a message, not wrapped."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import simbatch
    from lamsa_amd import hp
    n = 147456
    ref = simbatch.SimRef(1_000_000_000, n_contigs=12, seed=5, threads=16)
    B = simbatch.SimBatch(ref, n, 10000, "ont2d", seed=77, threads=16)
    assert B.n_cig > 0x7fffffff
    P = hp.make_para("ont2d")
    h = hp.LamsaHp(P, ref=(ref.pac, ref.l_pac, ref.seq_off, ref.seq_len))
    with pytest.raises(RuntimeError, match="2\\^31"):
        h.upload_batch(B)
    Bc = hp.compact_batch(B)
    h.upload_batch(Bc)
    raw = h.run_uploaded(fetch=True, raw=True)
    stream, r_off, r_len, status = raw
    assert (np.asarray(status) == 0).all()
    idx = list(range(0, n, 4099))[:32] + [n - 1]
    lp = reflib.lo_para("ont2d")
    # the sample for the oracle: its seed CIGARs re-based (in B they lie beyond what B's 32-bit offsets can say)
    cn = np.asarray(B.h_cig_n, np.int64); off = np.concatenate([[0], np.cumsum(cn)])
    sub = simbatch.take(B, idx)
    hits = np.concatenate([np.arange(B.hit_off[B.seed_off[i]], B.hit_off[B.seed_off[i + 1]]) for i in idx])
    sub.cig = np.concatenate([B.cig[off[k]:off[k + 1]] for k in hits] + [np.zeros(4, np.int32)])
    sub.h_cig_off = np.concatenate([np.concatenate([[0], np.cumsum(cn[hits])[:-1]]), np.zeros(4, np.int64)]).astype(np.int32)
    want = reflib.oracle_streams(sub, lp, 8)
    for k, i in enumerate(idx):
        got = stream[int(r_off[i]):int(r_off[i]) + int(r_len[i])].tolist()
        assert got == want[k], i
    h.close()


def test_hip_leaves_a_megabase_read_unaligned():
    """A read with more than 32767 seeds (1.2 Mbp at the 25-bp step): the device keeps seed ids in 16 bits, and wrapped ids would chain
    into wrong alignments with status 0.  The read is not aligned -- LAMSA_HP_ST_UNSUPPORTED, an empty result -- and a caller's other
    reads are (the batch is not refused: one unusual read in a user's map must not end the run)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import simbatch
    from lamsa_amd import hp
    ref = simbatch.SimRef(8_000_000, n_contigs=1, seed=3, threads=4)         # the simulator wants a contig of more than twice the read length
    B = simbatch.SimBatch(ref, 1, 1_200_000, "ont2d", seed=5, threads=1)
    h = hp.LamsaHp(hp.make_para("ont2d"), ref=(ref.pac, ref.l_pac, ref.seq_off, ref.seq_len), device=0)
    got, st = h.align_batch(B)
    assert int(st[0]) == 4 and list(got[0]) == [4, 0, 0]
    B2 = simbatch.SimBatch(ref, 6, 3000, "ont2d", seed=6, threads=1)
    import reflib
    got, st = h.align_batch(B2)
    assert (st == 0).all() and got == reflib.oracle_streams(B2, reflib.lo_para("ont2d"))
    h.close()


def test_hip_second_pass_after_scratch_overflow(tmp_path):
    """With the first pass's scratch capped far too low every read is flagged, nothing is written out of bounds, and
    the second pass (8x capacities) delivers the same streams as the oracle."""
    ref, reads, args, _ = goldenlib.stage_scenario("c3_ont", str(tmp_path))
    rt, over = goldenlib.para_from_args(args)
    lp = reflib.lo_para(rt, **over)
    B = reflib.Batch(ref, reads, lp)
    want = reflib.oracle_streams(B, lp)
    h = _handle(B, rt, over)
    h.set_scratch_limit(600 << 10)
    got, st = h.align_batch(B)
    assert h.last_kernel_ms(1) > 0                      # the retry kernel ran
    assert got == want and (st == 0).all()
    h.set_scratch_limit(0)
    got, st = h.align_batch(B)
    assert h.last_kernel_ms(1) == 0 and got == want
    h.close()


def test_hip_streaming_submit_collect(tmp_path):
    """lamsa_hp_submit_batch / lamsa_hp_collect_batch: chunks of different sizes two-deep in flight come back in
    submission order with the streams of the one-call boundary; queue-discipline errors; page-locked input arrays;
    a second pass needed while the next chunk is already queued."""
    from lamsa_amd import hp
    ref, reads, args, _ = goldenlib.stage_scenario("c3_ont", str(tmp_path))
    rt, over = goldenlib.para_from_args(args)
    lp = reflib.lo_para(rt, **over)
    B = reflib.Batch(ref, reads, lp)
    want = reflib.oracle_streams(B, lp)
    n = B.n_reads
    parts = [list(range(0, n // 2)), list(range(n // 2, n)), [], [n - 1, 0, 1], list(range(n))]
    h = _handle(B, rt, over)
    with pytest.raises(RuntimeError):
        h.collect_batch()                                # nothing in flight
    got = []
    h.submit_batch(B.take(parts[0]))
    for k in range(1, len(parts)):
        h.submit_batch(B.take(parts[k]))
        if k == 1:
            with pytest.raises(RuntimeError):
                h.submit_batch(B)                        # two already in flight
            with pytest.raises(RuntimeError):
                h.align_batch(B)                         # the one-call form is refused while chunks are in flight
        got.append(h.collect_batch())
    got.append(h.collect_batch())
    for k, idx in enumerate(parts):
        assert got[k][0] == [want[i] for i in idx], "chunk %d" % k
        assert (got[k][1] == 0).all()
    # the one-call form works again once the queue is empty
    assert h.align_batch(B)[0] == want
    # page-locked arrays
    Bp = hp.pinned_batch(B)
    h.submit_batch(Bp); h.submit_batch(Bp)
    a = h.collect_batch()[0]; b = h.collect_batch()[0]
    Bp.release()
    assert a == want and b == want
    # every read of the first chunk needs the second pass while the second chunk's kernel is already queued
    h.set_scratch_limit(600 << 10)
    h.submit_batch(B); h.submit_batch(B.take(parts[0]))
    a, st = h.collect_batch()
    assert h.last_kernel_ms(1) > 0 and a == want and (st == 0).all()
    b, st = h.collect_batch()
    assert b == [want[i] for i in parts[0]] and (st == 0).all()
    # runs of a resident batch, two deep
    h.set_scratch_limit(0)
    h.upload_batch(B)
    h.start_uploaded(); h.start_uploaded()
    with pytest.raises(RuntimeError):
        h.start_uploaded()                               # two already in flight
    with pytest.raises(RuntimeError):
        h.submit_batch(B)                                # not while resident runs are in flight
    a = h.finish_uploaded()[0]
    h.start_uploaded()
    b = h.finish_uploaded()[0]; c = h.finish_uploaded()[0]
    assert a == want and b == want and c == want
    with pytest.raises(RuntimeError):
        h.finish_uploaded()
    assert h.run_uploaded()[0] == want
    h.close()
    # a handle destroyed with a chunk still in flight
    h = _handle(B, rt, over)
    h.submit_batch(B)
    h.close()


def _records(stream):
    """(line, offset, chr, nstrand, score, NM, cigar words) of every record of one read's stream."""
    out, i = [], 3
    for ln in range(stream[1] + stream[2]):
        n_res = stream[i + 3]; i += 4
        for _ in range(n_res):
            off = (stream[i] & 0xffffffff) | (stream[i + 1] << 32)
            cn = stream[i + 6]
            out.append((ln, off, stream[i + 2], stream[i + 3], stream[i + 4], stream[i + 5], stream[i + 7:i + 7 + cn]))
            i += 7 + cn
    assert i == len(stream)
    return out


def test_hip_full_size_batch_properties():
    """A bench-sized batch (8192 x 10 kbp ONT-like reads, repeat-rich 400 Mbp reference): properties that do not
    need the oracle on every read -- CIGARs consume exactly the read, records lie inside their contig, AS and NM
    recomputed from CIGAR + sequences agree, almost every read finds its true locus, running twice or in another
    order changes nothing -- plus word-for-word equality with the oracle on a sample."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import simbatch
    from lamsa_amd import hp
    n, L = 8192, 10000
    ref = simbatch.SimRef(400_000_000, n_contigs=8, seed=5, threads=8)
    B = simbatch.SimBatch(ref, n, L, "ont2d", seed=4242, threads=8)
    P = hp.make_para("ont2d")
    h = hp.LamsaHp(P, ref=(ref.pac, ref.l_pac, ref.seq_off, ref.seq_len))
    h.upload_batch(B)
    got, st = h.run_uploaded()
    again, _ = h.run_uploaded()
    assert got == again                                                  # idempotent on a resident batch
    assert (st == 0).all()
    mapped = 0
    rng = np.random.default_rng(1)
    check_as = set(rng.choice(n, 96, replace=False).tolist())
    lp = reflib.lo_para("ont2d")
    for r in range(n):
        recs = _records(got[r])
        mapped += bool(recs)
        read = B.read_seq[int(B.read_off[r]):int(B.read_off[r + 1])]
        for ln, off, chrom, nstrand, score, NM, cig in recs:
            ops = [(w & 0xf, w >> 4) for w in cig]
            assert all(o in (0, 1, 2, 4) and l > 0 for o, l in ops)
            assert sum(l for o, l in ops if o in (0, 1, 4)) == L          # M, I, S consume the whole read
            rl = sum(l for o, l in ops if o in (0, 2))
            assert 1 <= chrom <= len(ref.seq_len) and off >= 1 and off - 1 + rl <= int(ref.seq_len[chrom - 1])
            if r in check_as:                                             # lamsa_res_aux, src/frag_check.c:793-853
                q = read if nstrand == 1 else np.where(read[::-1] < 4, 3 - read[::-1], 4)
                k0 = int(ref.seq_off[chrom - 1]) + off - 1
                ks = np.arange(k0, k0 + rl, dtype=np.int64)
                t = (ref.pac[ks >> 2] >> ((~ks & 3) << 1)) & 3
                qi = ti = mm = m = io = ie = do = de = 0
                for o, l in ops:
                    if o == 0:
                        d = int((q[qi:qi + l] != t[ti:ti + l]).sum()); mm += d; m += l - d; qi += l; ti += l
                    elif o == 1:
                        qi += l; io += 1; ie += l
                    elif o == 2:
                        ti += l; do += 1; de += l
                    else:
                        qi += l
                assert NM == mm + ie + de
                assert score == m * lp.match - mm * lp.mis - io * lp.ins_gapo - ie * lp.ins_gape - do * lp.del_gapo - de * lp.del_gape
    assert mapped >= 0.98 * n
    # another order, a subset: per-read streams must not depend on what else is in the batch
    sub = rng.choice(n, 512, replace=False).tolist()
    got_sub, _ = h.align_batch(simbatch.take(B, sub))
    assert got_sub == [got[i] for i in sub]
    # the oracle on a sample
    import bench
    first = bench.take_first(B, 96)
    want = reflib.oracle_streams(first, lp, 8)
    assert [i for i in range(96) if want[i] != got[i]] == []
    h.close()


OPTION_SETS = [
    dict(match=2, mis=5, ins_gapo=4, del_gapo=6, ins_gape=1, del_gape=2, ins_ext_o=4, del_ext_o=6, ins_ext_e=1, del_ext_e=2, band_w=50, end_bonus=3, ovlp_rat=0.5, ske_max=5),
    dict(split_len=50, res_mul_max=3, SV_len_thd=5000, per_aln_m=100, band_w=300),          # -g 50 -r 3 -V 5000 -p 100 -w 300 (wide band: HBM rows)
]


@pytest.mark.parametrize("name", ["c2_pacbio", "c3_ont", "c5_sv"])
@pytest.mark.parametrize("k", [0, 1])
def test_hip_scoring_and_limit_options(name, k, tmp_path):
    """Command-line options other than the presets (-m -M -O -E -w -b -v -s / -g -r -V -p -w) reach the kernels through
    lamsa_hp_para and give the oracle's result."""
    ref, reads, args, _ = goldenlib.stage_scenario(name, str(tmp_path))
    rt, over = goldenlib.para_from_args(args)
    over = dict(over); over.update(OPTION_SETS[k])
    lp = reflib.lo_para(rt, **over)
    B = reflib.Batch(ref, reads, lp)
    want = reflib.oracle_streams(B, lp)
    h = _handle(B, rt, over)
    got, st = h.align_batch(B)
    h.close()
    assert [i for i in range(B.n_reads) if want[i] != got[i]] == []
    assert (st == 0).all()
