"""Development check (CPU only): the device sources under the lane emulation against the oracle on simulated batches
of every bench profile.  `python tools/emu_vs_oracle.py [n_reads_scale]`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import reflib  # noqa: E402
import simbatch  # noqa: E402
from lamsa_amd import hp  # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    ref = simbatch.SimRef(3_100_000_000, n_contigs=24, seed=5, threads=8)
    bad = 0
    for prof, rt, n, L, over in (("ont2d", "ont2d", 384, 10000, {}), ("pacbio", "pacbio", 256, 5000, {}), ("default", "default", 256, 5000, {}),
                                 ("pb20k", "pacbio", 64, 20000, {"band_w": 200})):
        n = max(8, int(n * scale))
        B = simbatch.SimBatch(ref, n, L, prof, seed=77, threads=8)
        stats = []
        s = reflib.emu_streams(B, hp.make_para(rt, **over), stats=stats)
        s = s[0] if isinstance(s, tuple) else s
        want = reflib.oracle_streams(B, reflib.lo_para(rt, **over), 8)
        same = sum(1 for i in range(n) if list(want[i]) == list(s[i]))
        bad += n - same
        print("%-8s emulated kernels == oracle on %d / %d reads; path counters %s" % (prof, same, n, stats[:16]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
