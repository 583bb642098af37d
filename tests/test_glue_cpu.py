"""INTEGRATION.md, compiled: the binding (lamsa_amd/glue/lamsa_hp_glue.c) and the edits of tools/apply_glue.py applied to a
scratch copy of the reference, built with -DLAMSA_HP and linked against the CPU emulation of the C-ABI instead of
liblamsa_hp.so.  The reference binary -- its own file IO, GEM parsing, stage (4), ranking and SAM writer, with stages
(2),(3),(2'),(3') coming through include/lamsa_hp.h -- must write the SAM it wrote before.

Needs the reference's sources (only in the build container): skipped elsewhere.  Nothing of the reference is kept: the
scratch copy lives in pytest's tmp_path."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

import goldenlib as G
import reflib

REF_SRC = "/root/reference/src"
ROOT = reflib.ROOT

pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference's sources are not on this machine")


@pytest.fixture(scope="module")
def glued(tmp_path_factory):
    d = tmp_path_factory.mktemp("glue")
    src = str(d / "src")
    shutil.copytree(REF_SRC, src)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "apply_glue.py"), src], check=True)
    lib = reflib.emu_capi_lib()
    exe = str(d / "lamsa_glued")
    # the reference's own recipe (oracle/Makefile `ref`: -fcommon, -lm -lz -lpthread) plus the two additions of INTEGRATION.md section 1
    p = subprocess.run(["gcc", "-O2", "-fcommon", "-w", "-DLAMSA_HP", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "lamsa_amd", "glue"),
                        "-o", exe] + sorted(glob.glob(src + "/*.c")) + [lib, "-Wl,-rpath," + os.path.dirname(lib), "-lm", "-lz", "-lpthread", "-lstdc++"],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return exe


@pytest.mark.parametrize("name", ("c3_ont", "c5_sv", "c6_edge", "c7_rescue"))
def test_reference_with_the_binding_writes_the_same_sam(glued, name, tmp_path):
    ref, reads, args, gold_r0 = G.stage_scenario(name, str(tmp_path))
    gold = G.golden_full(name) if name in G.RESCUE_SCENARIOS else gold_r0          # default run: stage (4) on
    out = str(tmp_path / "out.sam")
    p = subprocess.run([glued, "aln"] + args + ["-t", "3", "-N", ref, reads, "-o", out], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(open(out).read()) == G.strip_pg(gold)
