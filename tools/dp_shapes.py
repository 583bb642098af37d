"""Development aid (CPU only): shapes of the DP calls the fill makes on a simulated batch of a bench profile, from the device sources
under the lane emulation (HP_DPLOG in hp_ksw.h).  `python tools/dp_shapes.py [profile] [n_reads] [read_len]`"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import reflib  # noqa: E402
import simbatch  # noqa: E402
from lamsa_amd import hp  # noqa: E402


def main():
    prof = sys.argv[1] if len(sys.argv) > 1 else "ont2d"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
    rt = {"ont2d": "ont2d", "pacbio": "pacbio", "default": "default"}[prof]
    ref = simbatch.SimRef(3_100_000_000, n_contigs=24, seed=5, threads=8)
    B = simbatch.SimBatch(ref, n, L, prof, seed=77, threads=8)
    E = reflib.emu()
    E.emu_dplog_on(1)
    reflib.emu_streams(B, hp.make_para(rt), lane_dp=False)
    E.emu_dplog.restype = C.c_longlong
    E.emu_dplog.argtypes = [C.c_void_p, C.c_longlong]
    cnt = E.emu_dplog(None, 0)
    buf = np.zeros(cnt, np.int64)
    E.emu_dplog(buf.ctypes.data, cnt)
    E.emu_dplog_on(0)
    rec = buf.reshape(-1, 5)
    for kind, name in ((0, "ksw_extend"), (1, "ksw_global")):
        r = rec[rec[:, 0] == kind]
        print("%s: %d calls (%.1f per read)" % (name, len(r), len(r) / n))
        q, t, w, cells = r[:, 1], r[:, 2], r[:, 3], r[:, 4]
        edges = [0, 16, 32, 62, 126, 160, 256, 512, 1024, 1 << 30]
        for a, b in zip(edges[:-1], edges[1:]):
            m = (q > a) & (q <= b)
            if not m.any():
                continue
            rows = (cells[m] / np.maximum(1, np.minimum(q[m], 2 * w[m] + 1))).sum() if kind == 0 else t[m].sum()
            print("  query %5d..%-6s %7d calls  mean q %6.1f  t %6.1f  w %6.1f | cells %11d (%.1f %%)  q*t %12d  ~rows %9d" % (
                a + 1, b if b < 1 << 30 else "", m.sum(), q[m].mean(), t[m].mean(), w[m].mean(), cells[m].sum(), 100.0 * cells[m].sum() / max(1, cells.sum()),
                (q[m] * t[m]).sum(), rows))


if __name__ == "__main__":
    main()
