"""The option sets of tools/crosscheck_cli.py as a test: freshly simulated reads, seeded by the reference's own GEM bundle and aligned
by the compiled reference (`oracle/_ref/lamsa aln`, default run, stage 4 on) against this repository's host program on the emulated
C-ABI -- twelve cases: three read types, `-S`, `-g/-r`, a full set of scoring options, FASTQ, `-C`, three stage-4 rescues, and seeds
cut + GEM driven by our own binary.  SAM must be identical in every case.  Needs the reference build and its GEM bundle: build
container only (skipped elsewhere)."""
import os
import shutil
import subprocess
import sys

import pytest

import reflib

ROOT = reflib.ROOT
NEED = [os.path.join(ROOT, "oracle", "_ref", "lamsa"), "/root/reference/gem"]


@pytest.mark.skipif(not all(os.path.exists(p) for p in NEED), reason="needs the compiled reference and its GEM bundle")
def test_option_sets_against_the_reference_binary():
    cli = reflib.emu_cli()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "crosscheck_cli.py"), cli, "40"], capture_output=True, text=True, timeout=1500)
    tmp = [l.split(":", 1)[1].strip() for l in p.stdout.splitlines() if l.startswith("tmp:")]
    for d in tmp:
        shutil.rmtree(d, ignore_errors=True)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert p.stdout.count("identical") == 12 and "DIFFERENT" not in p.stdout, p.stdout
