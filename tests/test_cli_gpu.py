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
    p = subprocess.run([BIN, "aln", "-N", "-R", "0"] + args + [ref, reads], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(p.stdout) == G.strip_pg(gold)


@pytest.mark.parametrize("shape", ["0", "1", "2"])
def test_every_shape_of_the_chaining_kernels(shape, tmp_path):
    """The chaining kernels exist with 2 432 / 3 392 / 5 120 LDS words per wave at 4 / 3 / 2 waves per SIMD and a batch picks one by its mean
    read length (chain_shape, hp_align_api.hip); LAMSA_HP_CHAIN_SHAPE forces one.  What runs out of LDS and what through HBM differs between
    them (cluster capacity, line sets, gap tables); the SAM must not."""
    for name in ("c3_ont", "c4_pb20k", "c5_sv", "c10_rearr_ont"):
        d = tmp_path / name; d.mkdir()
        ref, reads, args, gold = G.stage_scenario(name, str(d))
        p = subprocess.run([BIN, "aln", "-N", "-R", "0"] + args + [ref, reads], capture_output=True, text=True, env=dict(os.environ, LAMSA_HP_CHAIN_SHAPE=shape))
        assert p.returncode == 0, p.stderr[-2000:]
        assert G.strip_pg(p.stdout) == G.strip_pg(gold), (name, shape)


def test_small_batches(tmp_path):
    ref, reads, args, gold = G.stage_scenario("c2_pacbio", str(tmp_path))
    p = subprocess.run([BIN, "aln", "-N", "-R", "0", "--batch", "4"] + args + [ref, reads], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(p.stdout) == G.strip_pg(gold)


def test_end_to_end_files_at_scale(tmp_path):
    """1024 simulated 6-kbp ONT-like reads written as the files `lamsa aln` reads (FASTA + GEM map text, tools/simfiles.py),
    aligned by the product binary in several GPU batches with all host threads, against the oracle's CLI on the same
    files: SAM byte-identical."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import reflib
    import simbatch
    import simfiles
    ref = simbatch.SimRef(200_000_000, n_contigs=6, seed=5, threads=8)
    B = simbatch.SimBatch(ref, 1024, 6000, "ont2d", seed=31, threads=8)
    d = str(tmp_path)
    simfiles.write_index(d + "/ref.fa", ref)
    simfiles.write_reads(d + "/reads.fa", B)
    reflib.build_oracle()
    want = subprocess.run([os.path.join(ROOT, "oracle", "lamsa_oracle"), "aln", "-T", "ont2d", "-t", "16", "-R", "0", d + "/ref.fa", d + "/reads.fa"],
                          capture_output=True, text=True)
    assert want.returncode == 0, want.stderr[-2000:]
    got = subprocess.run([BIN, "aln", "-N", "-T", "ont2d", "-R", "0", "-t", "16", "--batch", "300", d + "/ref.fa", d + "/reads.fa"], capture_output=True, text=True)
    assert got.returncode == 0, got.stderr[-2000:]
    a, b = G.strip_pg(got.stdout), G.strip_pg(want.stdout)
    assert len(a.splitlines()) >= 1024 + 6
    assert a == b
    # the binary hit stream on the MI355X: written beside a run, then aligned from (no GEM text, no parse) -- the same SAM both times
    one = subprocess.run([BIN, "aln", "-N", "-T", "ont2d", "-R", "0", "-t", "16", "--batch", "300", "--save-hits", d + "/hits.bin", d + "/ref.fa", d + "/reads.fa"], capture_output=True, text=True)
    assert one.returncode == 0 and G.strip_pg(one.stdout) == b, one.stderr[-2000:]
    two = subprocess.run([BIN, "aln", "-N", "-T", "ont2d", "-R", "0", "-t", "4", "--hits", d + "/hits.bin", d + "/ref.fa", d + "/reads.fa"], capture_output=True, text=True)
    assert two.returncode == 0 and G.strip_pg(two.stdout) == b, two.stderr[-2000:]
    assert os.path.getsize(d + "/hits.bin") < os.path.getsize(d + "/reads.fa.seed.gem.map")            # compact form (one byte per seed-CIGAR element): smaller than the text
    # --shard i/3 (one process after the other on this GPU), from the text and from the hit stream, each into a file of its own (-o: the
    # chunk text goes through a mapping of the file, LAMSA_MAP_OUT_MIN lowered so that these small chunks take that path): the files written
    # one after the other are the unsharded SAM
    for src in ([], ["--hits", d + "/hits.bin"]):
        text = ""
        for i in range(3):
            out = d + "/shard%d.sam" % i
            p = subprocess.run([BIN, "aln", "-N", "-T", "ont2d", "-R", "0", "-t", "8", "--batch", "200", "--shard", "%d/3" % i, "-o", out] + src + [d + "/ref.fa", d + "/reads.fa"],
                               capture_output=True, text=True, env=dict(os.environ, LAMSA_MAP_OUT_MIN="4096"))
            assert p.returncode == 0, p.stderr[-2000:]
            text += open(out).read()
        assert G.strip_pg(text) == b, src


@pytest.mark.parametrize("name", G.RESCUE_SCENARIOS)
def test_stage4_bwt_rescue_matches_reference_default_run(name, tmp_path):
    """Default -R on the MI355X: stage 4 plans on the host, runs its DP jobs as one lamsa_hp_dp_batch on a second handle,
    and the SAM equals the reference's default run."""
    ref, reads, args, _ = G.stage_scenario(name, str(tmp_path))
    p = subprocess.run([BIN, "aln", "-N", "--batch", "5"] + args + [ref, reads], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(p.stdout) == G.strip_pg(G.golden_full(name))


def test_chunks_dealt_over_several_devices(tmp_path):
    """--devices: one handle per listed device, chunks round-robin, output in input order (two handles on device 0 here)."""
    ref, reads, args, _ = G.stage_scenario("c7_rescue", str(tmp_path))
    p = subprocess.run([BIN, "aln", "-N", "--batch", "2", "--devices", "0,0,0"] + args + [ref, reads], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(p.stdout) == G.strip_pg(G.golden_full("c7_rescue"))


@pytest.mark.parametrize("name", G.SCENARIOS)
def test_default_run_equals_reference_default_run(name, tmp_path):
    """No -R on the command line (stage 4 on, -R 300): the reference's default output -- golden_full.sam where stage 4
    changed something, golden_R0.sam where the reference's two runs were identical."""
    ref, reads, args, want_r0 = G.stage_scenario(name, str(tmp_path))
    want = G.golden_full(name) if name in G.RESCUE_SCENARIOS else want_r0
    p = subprocess.run([BIN, "aln", "-N"] + args + [ref, reads], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(p.stdout) == G.strip_pg(want)


def test_map_is_read_while_the_mapper_writes_it_on_the_gpu(tmp_path):
    """The product binary starts the mapper (a stand-in that writes the fixture's map in slow pieces) BEFORE its first GPU call and reads
    the map while it grows: same SAM as with `-N` on the finished map, and with `--seed-first`."""
    import shutil
    import test_cli_cpu
    ref, reads, args, gold = G.stage_scenario("c3_ont", str(tmp_path))
    keep = str(tmp_path / "map.keep")
    shutil.move(reads + ".seed.gem.map", keep)
    gem = test_cli_cpu._fake_mapper(str(tmp_path / "gem"), keep, pieces=8, delay=0.2)
    for extra in ([], ["--seed-first"]):
        p = subprocess.run([BIN, "aln", "-R", "0", "-t", "4", "--batch", "16", "--gem-dir", gem] + extra + args + [ref, reads], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        assert "gem-mapper done!" in p.stderr
        assert G.strip_pg(p.stdout) == G.strip_pg(gold)
        os.remove(reads + ".seed.gem.map")


REF_BIN = os.path.join(ROOT, "oracle", "_ref", "lamsa")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="the compiled reference (oracle/_ref/lamsa, built where /root/reference exists) did not travel")
@pytest.mark.parametrize("workload,n_reads", [("ont10k", 500), ("pb5k", 500), ("pb20k", 200), ("sv10k", 500)])
def test_sam_identical_to_the_reference_binary_at_bench_shapes(workload, n_reads, tmp_path):
    """At-scale parity against the reference ITSELF (not the oracle): reads and seed hits of a bench workload (bench.WORKLOADS: read length,
    error profile, options) written as the files `lamsa aln` reads, aligned by the compiled reference (`oracle/_ref/lamsa aln -N -I -R 0`,
    from /root/reference/src, CPU) and by the product binary on the MI355X -- the same SAM, record for record."""
    import shutil
    import sys
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import simbatch
    import simfiles
    wl = bench.WORKLOADS[workload]
    ref = simbatch.SimRef(600_000_000, n_contigs=12, seed=5, threads=16)
    B = simbatch.SimBatch(ref, n_reads, wl["length"], wl["profile"], seed=4242, threads=16)
    d = str(tmp_path)
    simfiles.write_index(d + "/ref.fa", ref)
    for ext in ("bwt", "sa"):                       # loaded by the reference at start-up, never searched with -R 0 (src/lamsa_aln.c:1233)
        shutil.copy(os.path.join(ROOT, "tests", "golden", "ref", "ref.fa." + ext), d + "/ref.fa." + ext)
    p = simbatch.PROFILES[wl["profile"]]
    simfiles.write_reads(d + "/reads.fa", B, seed_len=50, seed_step=p["seed_step"], workers=16)
    with open(d + "/reads.fa.seed.info", "w") as f:
        for r in range(n_reads):
            f.write("r%d %d %d %d\n" % (r, int(B.seed_all[r]), int(B.last_len[r]), int(B.read_off[r + 1] - B.read_off[r])))
    args = [] if wl["read_type"] == "default" else ["-T", wl["read_type"]]
    for k, v in wl["over"].items():
        args += [{"band_w": "-w", "SV_len_thd": "-V"}[k], str(v)]
    want = subprocess.run([REF_BIN, "aln"] + args + ["-t", "16", "-N", "-I", "-R", "0", "-o", d + "/out.sam", d + "/ref.fa", d + "/reads.fa"], capture_output=True, text=True, timeout=900)
    assert want.returncode == 0, want.stderr[-2000:]
    res = bench.compare_with_product(d, args, 16, n_reads)
    assert res == "%d/%d reads" % (n_reads, n_reads), res


GLUED = os.path.join(ROOT, "oracle", "_ref", "lamsa_glued")


@pytest.mark.parametrize("name", ("c3_ont", "c5_sv", "c6_edge", "c7_rescue", "c9_rearr"))
def test_reference_with_the_binding_on_the_gpu(name, tmp_path):
    """The drop-in boundary on hardware: the REFERENCE's own `lamsa aln` (its file IO, GEM parsing, stage (4), ranking, SAM writer) with the
    binding of INTEGRATION.md compiled in (oracle/Makefile `glued`: lamsa_amd/glue/lamsa_hp_glue.c + tools/apply_glue.py, linked against
    liblamsa_hp.so), stages (2),(3),(2'),(3') on the MI355X through include/lamsa_hp.h: the SAM the reference writes without it (default run:
    stage (4) on).  The binary is built in the build container and travels like oracle/_ref/lamsa."""
    if not os.path.exists(GLUED):
        pytest.skip("oracle/_ref/lamsa_glued was not built (needs the reference's sources: build container only)")
    ref, reads, args, gold_r0 = G.stage_scenario(name, str(tmp_path))
    gold = G.golden_full(name) if name in G.RESCUE_SCENARIOS else gold_r0
    out = str(tmp_path / "out.sam")
    p = subprocess.run([GLUED, "aln"] + args + ["-t", "3", "-N", ref, reads, "-o", out], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(open(out).read()) == G.strip_pg(gold)


@pytest.mark.parametrize("workload", ["ont10k", "sv10k"])
def test_default_run_with_an_index_built_here(workload):
    """The reference's DEFAULT run (no -R 0: stage 4, the BWT rescue, searches the FM index) against the product binary's on bench-shaped files:
    a 60 Mbp stand-in whose .bwt / .sa the product's own `lamsa index --from-pac` builds on the spot (the suffixes sorted block by block,
    lamsa_amd/host/index.cpp), 400 simulated reads with their seed hits as GEM map text.  The same SAM, read by read.  bench.py does this
    after every default run (cpu_baseline.reference_binary.default_run), tools/default_run.py at 300 Mbp for every workload."""
    import sys
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import simbatch
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "lamsa")):
        pytest.skip("the compiled reference did not travel")
    simbatch.build()
    n = 400
    r = bench.default_run_check(workload, bench.WORKLOADS[workload], min(os.cpu_count() or 8, 32), n, 60_000_000)
    assert "error" not in r, r
    assert r["gpu_equals_reference"] == "%d/%d reads" % (n, n), r
