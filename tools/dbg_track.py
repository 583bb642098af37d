import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import goldenlib, reflib, tempfile
from lamsa_amd import hp
d = tempfile.mkdtemp()
ref, reads, args, _ = goldenlib.stage_scenario("c1_perfect", d)
rt, over = goldenlib.para_from_args(args)
lp = reflib.lo_para(rt, **over)
B = reflib.Batch(ref, reads, lp)
want = reflib.oracle_streams(B, lp)
h = hp.LamsaHp(hp.make_para(rt, **over), ref=(B.pac, B.l_pac, B.seq_off, B.seq_len))
got, st = h.align_batch(B)
work = h.last_work
print('work', [ (i, int(work[2*i+1])) for i in (7, 51, 0, 1) ])
for i in range(B.n_reads):
    if want[i] != got[i]:
        print("read", i, "seeds", int(B.seed_all[i]), "slots", int(B.seed_off[i+1]-B.seed_off[i]), "hits", int(B.hit_off[B.seed_off[i+1]]-B.hit_off[B.seed_off[i]]), "want hdr", want[i][:3], "got hdr", got[i][:3], "len", len(want[i]), len(got[i]))
        w, g = want[i], got[i]
        print("  want first line", w[3:7], "got", g[3:7])
