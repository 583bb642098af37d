"""Local cross-check (needs /root/reference built into oracle/_ref, so it runs in the build container only):
simulated reads -> reference `lamsa aln` (default run, stage 4 included) vs this repo's host CLI (tests/_build/lamsa_emu by default, or the
product binary given as argv[1] on a GPU box holding the staged inputs).  Also covers FASTQ input and the output options.

    python tools/crosscheck_cli.py [binary] [n_reads]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import simdata  # noqa: E402
import goldenlib as G  # noqa: E402
import make_golden_reads as M  # noqa: E402

LAMSA = os.path.join(ROOT, "oracle", "_ref", "lamsa")


def main():
    binary = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "_build", "lamsa_emu")
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.default_rng(M.REF["seed"])
    contigs = simdata.make_reference(rng, M.REF["contigs"], M.REF["repeats"])
    tmp = tempfile.mkdtemp(prefix="lamsa_xc_")
    ref = os.path.join(tmp, "ref.fa")
    simdata.write_fasta(ref, [("chr%d" % (i + 1), c) for i, c in enumerate(contigs)])
    subprocess.run([LAMSA, "index", ref], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    bad = 0
    cases = [("pacbio", ["-T", "pacbio"], 5000, "pacbio", {}), ("ont", ["-T", "ont2d"], 8000, "ont", {}), ("sv", [], 10000, "lowerr", {"sv_frac": 0.67}),
             ("soft", ["-T", "ont2d", "-S"], 4000, "ont", {}), ("split", ["-T", "pacbio", "-g", "50", "-r", "3"], 6000, "pacbio", {"sv_frac": 0.5}),
             ("score", ["-m", "2", "-M", "5", "-O", "4,6", "-E", "1,2", "-w", "50", "-b", "3", "-v", "0.5", "-s", "5"], 5000, "lowerr", {"sv_frac": 0.3}),
             ("fastq", ["-T", "ont2d"], 3000, "ont", {"fastq": True}), ("fa_C", ["-C"], 3000, "lowerr", {}),
             ("rescue", [], 4000, "lowerr", {"rescue": True}), ("rescueP", ["-T", "pacbio"], 5000, "pacbio", {"rescue": True}), ("rescueR", ["-R", "150", "-k", "15"], 4000, "lowerr", {"rescue": True})]
    # (`-C` together with FASTQ input is not compared: the reference's reverse-strand QUAL loop, src/lamsa_aln.c:1043, runs off the array.)
    for si, (name, args, length, prof, extra) in enumerate(cases):
        rng = np.random.default_rng(7000 + si)
        sub, ins, dele = simdata.PROFILES[prof]
        if extra.get("rescue"):
            reads = M.rescue_reads(rng, contigs, n_reads, length, sub, ins, dele)
        else:
            reads = simdata.simulate_reads(rng, contigs, n_reads, length, sub, ins, dele, extra.get("sv_frac", 0.0))
        rd = os.path.join(tmp, name + (".fq" if extra.get("fastq") else ".fa"))
        if extra.get("fastq"):
            with open(rd, "w") as f:
                for nm, s in reads:
                    q = "".join(chr(33 + int(x)) for x in rng.integers(2, 40, len(s)))
                    f.write("@%s some comment\n%s\n+\n%s\n" % (nm, "".join("ACGTN"[int(c)] for c in s), q))
        else:
            simdata.write_fasta(rd, reads, width=70)
        want = subprocess.run([LAMSA, "aln"] + args + ["-t", "1", ref, rd], check=True, capture_output=True, text=True).stdout      # default -R: stage 4 on
        got = subprocess.run([binary, "aln", "-N"] + args + [ref, rd], check=True, capture_output=True, text=True).stdout
        same = G.strip_pg(got) == G.strip_pg(want)
        print("%-8s %s  %d lines" % (name, "identical" if same else "DIFFERENT", len(want.splitlines())))
        if not same:
            bad += 1
            for a, b in zip(G.strip_pg(got).splitlines(), G.strip_pg(want).splitlines()):
                if a != b:
                    print("  got : " + a[:400]); print("  want: " + b[:400]); break
    # the seeding front end itself: our binary cuts the seeds and runs the bundled GEM mapper (container only)
    gem_dir = os.path.join(os.path.dirname(LAMSA), "gem")
    if os.path.exists(os.path.join(gem_dir, "gem-mapper")):
        rd = os.path.join(tmp, "ont.fa")
        want = subprocess.run([LAMSA, "aln", "-T", "ont2d", "-t", "1", "-R", "0", ref, rd], check=True, capture_output=True, text=True).stdout
        for f in (rd + ".seed", rd + ".seed.gem.map", rd + ".seed.info"):
            if os.path.exists(f):
                os.remove(f)
        got = subprocess.run([binary, "aln", "-T", "ont2d", "-R", "0", "--gem-dir", gem_dir, ref, rd], check=True, capture_output=True, text=True).stdout
        same = G.strip_pg(got) == G.strip_pg(want)
        print("%-8s %s  (seeds cut and GEM run by our binary)" % ("seeding", "identical" if same else "DIFFERENT"))
        bad += 0 if same else 1
    print("tmp:", tmp)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
