import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "tools")]
import numpy as np
import reflib, goldenlib
from lamsa_amd import hp
lp = reflib.lo_para("ont2d")
H = hp.LamsaHp(hp.make_para("ont2d"))
rng = np.random.default_rng(5)
e = np.zeros(0, np.uint8)
t200 = rng.integers(0, 4, 200, dtype=np.uint8)
jobs = [(t200, t200), (t200[:129], t200[:128]), (t200[:128], t200[:128]), (t200[:127], t200[:127]), (t200[:140], t200[:160])]
for kind, w, h0 in ((1, 2, 1), (1, 100, 1), (1, 2, 50), (1, 100, 3), (1,100,8)):
    got = H.dp_batch(jobs, kind, w, h0)
    want = reflib.oracle_dp(jobs, lp, kind, w, h0)
    for i in range(len(jobs)):
        print(kind, w, h0, i, "score", want["score"][i], got["score"][i], "qle", want["qle"][i], got["qle"][i], "tle", want["tle"][i], got["tle"][i], "status", got["status"][i], "cig", want["cigars"][i][:6], got["cigars"][i][:6])
