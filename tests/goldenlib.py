"""Readers for the committed golden fixtures (tests/golden/)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
READ_TYPES = ("default", "pacbio", "ont2d")


def dp_vectors():
    """Yield (read_type, jobs, calls) with calls = list of (kind, w, h0, expected dict)."""
    z = np.load(os.path.join(GOLD, "dp_vectors.npz"))
    for rt in READ_TYPES:
        ql, tl = z[rt + "_qlen"], z[rt + "_tlen"]
        qo = np.concatenate([[0], np.cumsum(ql)]); to = np.concatenate([[0], np.cumsum(tl)])
        jobs = [(z[rt + "_q"][qo[i]:qo[i + 1]].copy(), z[rt + "_t"][to[i]:to[i + 1]].copy()) for i in range(len(ql))]
        calls = []
        ci = 0
        while "%s_c%d_kwh" % (rt, ci) in z:
            k = "%s_c%d_" % (rt, ci)
            kind, w, h0 = [int(x) for x in z[k + "kwh"]]
            cn = z[k + "cign"]; co = np.concatenate([[0], np.cumsum(cn)])
            cig = [z[k + "cig"][co[i]:co[i + 1]].tolist() for i in range(len(cn))]
            calls.append((kind, w, h0, dict(score=z[k + "score"], qle=z[k + "qle"], tle=z[k + "tle"], cigars=cig)))
            ci += 1
        yield rt, jobs, calls


def same_dp(exp, got, kind):
    """Bit-exact comparison of two DP result dicts; returns list of mismatching job indices."""
    bad = []
    for i in range(len(exp["cigars"])):
        ok = int(exp["score"][i]) == int(got["score"][i]) and list(exp["cigars"][i]) == list(got["cigars"][i])
        if kind == 1:
            ok = ok and int(exp["qle"][i]) == int(got["qle"][i]) and int(exp["tle"][i]) == int(got["tle"][i])
        if "status" in got and int(got["status"][i]) != 0:
            ok = False
        if not ok:
            bad.append(i)
    return bad


SCENARIOS = ("c1_perfect", "c2_pacbio", "c3_ont", "c4_pb20k", "c5_sv", "c6_edge", "c7_rescue", "c8_rescue_ont", "c9_rearr", "c10_rearr_ont")
RESCUE_SCENARIOS = tuple(n for n in SCENARIOS if os.path.exists(os.path.join(GOLD, n, "golden_full.sam.gz")))      # stage 4 changes their output: golden_full.sam is the default run


def stage_scenario(name, tmpdir):
    """Materialise a whole-path fixture in tmpdir (the GEM map is stored gzipped).
    Returns (ref_prefix, reads_path, args list, golden SAM text without @PG)."""
    import gzip
    import shutil
    d = os.path.join(GOLD, name)
    for ext in (".ann", ".amb", ".pac", ".bwt", ".sa"):
        shutil.copy(os.path.join(GOLD, "ref", "ref.fa" + ext), os.path.join(tmpdir, "ref.fa" + ext))
    for src, dst in (("reads.fa.gz", "reads.fa"), ("reads.fa.seed.gem.map.gz", "reads.fa.seed.gem.map")):
        with gzip.open(os.path.join(d, src), "rb") as f, open(os.path.join(tmpdir, dst), "wb") as g:
            g.write(f.read())
    args = open(os.path.join(d, "args.txt")).read().split()
    gold = gzip.open(os.path.join(d, "golden_R0.sam.gz"), "rt").read()
    return os.path.join(tmpdir, "ref.fa"), os.path.join(tmpdir, "reads.fa"), args, gold


def golden_full(name):
    """SAM of the reference's default run (stage 4 on) for the scenarios where it differs from -R 0."""
    import gzip
    return gzip.open(os.path.join(GOLD, name, "golden_full.sam.gz"), "rt").read()


def strip_pg(text):
    return "".join(l + "\n" for l in text.split("\n") if l and not l.startswith("@PG"))


def para_from_args(args):
    """(read_type, overrides) for the fixture's command-line options."""
    rt, over, i = "default", {}, 0
    while i < len(args):
        if args[i] == "-T":
            rt = args[i + 1]; i += 2
        elif args[i] == "-w":
            over["band_w"] = int(args[i + 1]); i += 2
        elif args[i] == "-V":
            over["SV_len_thd"] = int(args[i + 1]); i += 2
        else:
            i += 1
    return rt, over
