"""The product binary lamsa_amd/bin/lamsa (host C++ + liblamsa_hp.so, HIP kernels) on the MI355X: SAM byte-identical
to the reference's (`-R 0` goldens)."""
import os
import subprocess

import pytest

import goldenlib as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "lamsa_amd", "bin", "lamsa")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", G.SCENARIOS)
def test_sam_identical_to_reference(name, tmp_path):
    assert os.path.exists(BIN), "lamsa_amd/bin/lamsa is not built (python -c 'import __graft_entry__ as g; g.build()')"
    ref, reads, args, gold = G.stage_scenario(name, str(tmp_path))
    p = subprocess.run([BIN, "aln", "-R", "0"] + args + [ref, reads], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(p.stdout) == G.strip_pg(gold)


def test_small_batches(tmp_path):
    ref, reads, args, gold = G.stage_scenario("c2_pacbio", str(tmp_path))
    p = subprocess.run([BIN, "aln", "-R", "0", "--batch", "4"] + args + [ref, reads], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(p.stdout) == G.strip_pg(gold)
