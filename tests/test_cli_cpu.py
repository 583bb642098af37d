"""The `lamsa aln` host program (lamsa_amd/host): FASTA/FASTQ + GEM-map parsing, result ranking / MAPQ / XA and
the SAM writer, byte-for-byte against the reference's SAM (tests/golden/*/golden_R0.sam, made by the reference
binary with `-R 0`; tools/make_golden_reads.py).  On the CPU the host program is linked against the emulated
C-ABI (tests/emu); tests/test_cli_gpu.py runs the product binary on the MI355X."""
import gzip
import os
import re
import shutil
import subprocess

import pytest

import goldenlib as G
import reflib


@pytest.fixture(scope="module")
def cli():
    return reflib.emu_cli()


@pytest.mark.parametrize("name", G.SCENARIOS)
def test_sam_identical_to_reference(cli, name, tmp_path):
    ref, reads, args, gold = G.stage_scenario(name, str(tmp_path))
    p = subprocess.run([cli, "aln", "-N", "-R", "0"] + args + [ref, reads], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "@PG\tID:lamsa" in p.stdout
    assert G.strip_pg(p.stdout) == G.strip_pg(gold)


def test_output_file_small_batches_and_gz_reads(cli, tmp_path):
    """-o FILE; reads given gzipped; --batch smaller than the file (several chunks) -- same SAM."""
    ref, reads, args, gold = G.stage_scenario("c5_sv", str(tmp_path))
    gz = reads + ".gz"
    with open(reads, "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    shutil.move(reads + ".seed.gem.map", gz + ".seed.gem.map")
    out = str(tmp_path / "out.sam")
    p = subprocess.run([cli, "aln", "-N", "-R", "0", "--batch", "3", "-o", out] + args + [ref, gz], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(open(out).read()) == G.strip_pg(gold)


def test_error_paths(cli, tmp_path):
    ref, reads, args, gold = G.stage_scenario("c1_perfect", str(tmp_path))
    assert subprocess.run([cli], capture_output=True).returncode == 1                                  # usage
    assert subprocess.run([cli, "aln", ref], capture_output=True).returncode == 1                       # one positional argument only
    assert subprocess.run([cli, "aln", "-T", "nanopore", ref, reads], capture_output=True).returncode == 1
    assert subprocess.run([cli, "aln", "-v", "1.5", ref, reads], capture_output=True).returncode == 1
    assert subprocess.run([cli, "aln", str(tmp_path / "nope.fa"), reads], capture_output=True).returncode != 0
    os.remove(reads + ".seed.gem.map")                                                                 # no seeding results: must say so, not run
    p = subprocess.run([cli, "aln", "-N", ref, reads], capture_output=True, text=True)
    assert p.returncode != 0 and "gem" in p.stderr.lower()
    p = subprocess.run([cli, "aln", "--gem-dir", str(tmp_path / "nowhere"), ref, reads], capture_output=True, text=True)    # seeding wanted, no mapper there
    assert p.returncode != 0 and "gem mapper not found" in p.stderr.lower()
    seeds = open(reads + ".seed").read().split("\n")                                                     # but the seeds were cut as the reference cuts them
    assert seeds[0].endswith("_0:0") and len(seeds[1]) == 50 and seeds[2].endswith("_1:100")


def test_output_file_written_by_all_threads_side_by_side(cli, tmp_path):
    """A regular output file (-o FILE, or stdout redirected into one): the SAM text of a chunk is written by the threads side by side, each at
    its own offset (chunks of 8 MB and more; LAMSA_MAP_OUT_MIN lowers the limit); the file is the same as the text written to a pipe, header
    and chunk order included -- and the parallel path really ran."""
    ref, reads, args, want = G.stage_scenario("c2_pacbio", str(tmp_path))
    out = str(tmp_path / "out.sam")
    sizes = []
    for limit, t in (("1", "3"), ("1", "5"), ("100000000", "3"), ("mixed", "3")):
        if limit == "mixed":                            # some chunks side by side, the smaller ones through the stream behind them
            limit = str(sorted(sizes)[len(sizes) // 2]) if sizes else "1"
        p = subprocess.run([cli, "aln", "-N", "-R", "0", "-t", t, "--batch", "7", "-o", out] + args + [ref, reads], capture_output=True, text=True, env=dict(os.environ, LAMSA_MAP_OUT_MIN=limit, LAMSA_TRACE="1"))
        assert p.returncode == 0, p.stderr[-2000:]
        assert G.strip_pg(open(out).read()) == G.strip_pg(want), (limit, t)
        n_par = int(re.search(r"\[write\] (\d+) chunks written by all threads side by side", p.stderr).group(1))
        if limit in ("1", "100000000"):
            assert (n_par > 0) == (limit == "1"), (limit, t, n_par)
        if not sizes:                                   # bytes of SAM text per chunk of 7 reads
            recs = [l for l in open(out).read().splitlines(True) if not l.startswith("@")]
            names = []
            for l in recs:
                if not names or names[-1][0] != l.split("\t")[0]: names.append([l.split("\t")[0], 0])
                names[-1][1] += len(l)
            sizes = [sum(x[1] for x in names[i:i + 7]) for i in range(0, len(names), 7)]
    with open(out, "w") as fo:                          # a shell's `>`
        p = subprocess.run([cli, "aln", "-N", "-R", "0", "-t", "3", "--batch", "7"] + args + [ref, reads], stdout=fo, stderr=subprocess.PIPE, text=True, env=dict(os.environ, LAMSA_MAP_OUT_MIN="1", LAMSA_TRACE="1"))
    assert p.returncode == 0 and "[write] 0 chunks written" not in p.stderr, p.stderr[-2000:]
    assert G.strip_pg(open(out).read()) == G.strip_pg(want)
    with open(out, "a") as fo:                          # `>>`: an append-mode descriptor ignores offsets: one after the other
        p = subprocess.run([cli, "aln", "-N", "-R", "0", "-t", "3", "--batch", "7"] + args + [ref, reads], stdout=fo, stderr=subprocess.PIPE, text=True, env=dict(os.environ, LAMSA_MAP_OUT_MIN="1", LAMSA_TRACE="1"))
    assert p.returncode == 0 and "[write] 0 chunks written" in p.stderr, p.stderr[-2000:]
    assert G.strip_pg(open(out).read()) == G.strip_pg(want) + G.strip_pg(want)


def test_seed_cigars_in_words_when_an_element_does_not_fit_a_byte(cli, tmp_path):
    """The parser writes the seed CIGARs in the boundary's compact form (a byte per element) and parses a chunk again into 32-bit words
    when an element is longer than 63 (seeds longer than that); LAMSA_WIDE_CIGARS=1 sends every chunk that way: same SAM, and a hit
    stream saved from it replays to the same SAM."""
    ref, reads, args, want = G.stage_scenario("c3_ont", str(tmp_path))
    env = dict(os.environ, LAMSA_WIDE_CIGARS="1")
    hits = str(tmp_path / "hits_w.bin")
    p = subprocess.run([cli, "aln", "-N", "-R", "0", "--batch", "7", "--save-hits", hits] + args + [ref, reads], capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(p.stdout) == G.strip_pg(want)
    q = subprocess.run([cli, "aln", "-R", "0", "--hits", hits] + args + [ref, reads], capture_output=True, text=True)
    assert q.returncode == 0 and G.strip_pg(q.stdout) == G.strip_pg(want), q.stderr[-2000:]


def test_binary_hit_stream_round_trip(cli, tmp_path):
    """--save-hits writes the parsed seed hits chunk by chunk; --hits replays them (no GEM text, no parse) -- same SAM;
    a stream written with other seeding options, or for other reads, is refused."""
    ref, reads, args, want = G.stage_scenario("c3_ont", str(tmp_path))
    hits = str(tmp_path / "hits.bin")
    p = subprocess.run([cli, "aln", "-N", "-R", "0", "--batch", "5", "--save-hits", hits] + args + [ref, reads], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(p.stdout) == G.strip_pg(want)
    os.remove(reads + ".seed.gem.map")                         # the replay must not need it
    q = subprocess.run([cli, "aln", "-R", "0", "--hits", hits] + args + [ref, reads], capture_output=True, text=True)
    assert q.returncode == 0, q.stderr[-2000:]
    assert G.strip_pg(q.stdout) == G.strip_pg(want)
    # the same reads as FASTA lines of 60 columns with DOS line ends (records handed out as spans of the mapped file, parsed by all
    # threads), and as FASTQ (the sequential reader): the replay does not care
    recs = [(l.split("\n", 1)[0], "".join(l.split("\n")[1:])) for l in open(reads).read().split(">")[1:]]
    folded, fq = str(tmp_path / "folded.fa"), str(tmp_path / "reads.fq")
    with open(folded, "w", newline="") as f:
        for name, seq in recs:
            f.write(">" + name + "\r\n" + "".join(seq[i:i + 60] + "\r\n" for i in range(0, len(seq), 60)))
    with open(fq, "w") as f:
        for name, seq in recs:
            f.write("@" + name + "\n" + seq + "\n+\n" + "@" * len(seq) + "\n")
    for other in (folded, fq):
        q = subprocess.run([cli, "aln", "-R", "0", "-t", "3", "--hits", hits] + args + [ref, other], capture_output=True, text=True)
        assert q.returncode == 0, q.stderr[-2000:]
        got = [l.split("\t") for l in G.strip_pg(q.stdout).splitlines() if not l.startswith("@")]
        exp = [l.split("\t") for l in G.strip_pg(want).splitlines() if not l.startswith("@")]
        assert [g[:10] for g in got] == [e[:10] for e in exp] and [g[11:] for g in got] == [e[11:] for e in exp], other      # (the FASTQ run prints qualities)
    bad = subprocess.run([cli, "aln", "-R", "0", "--hits", hits, "-T", "pacbio", ref, reads], capture_output=True, text=True)
    assert bad.returncode != 0 and "hit stream" in bad.stderr
    os.makedirs(str(tmp_path / "other"))
    ref2, reads2, args2, _ = G.stage_scenario("c2_pacbio", str(tmp_path / "other"))
    bad = subprocess.run([cli, "aln", "-R", "0", "--hits", hits] + args + [ref, reads2], capture_output=True, text=True)
    assert bad.returncode != 0 and "hit stream" in bad.stderr


@pytest.mark.parametrize("name", G.RESCUE_SCENARIOS)
def test_stage4_bwt_rescue_matches_reference_default_run(cli, name, tmp_path):
    """Default -R: regions the first rounds left uncovered are searched in the FM index (<ref>.bwt/.sa), chained and
    aligned (lamsa_amd/host/rescue.cpp + one DP batch) -- SAM identical to the reference's default run; without the index
    files the stage is skipped with a note and the output is the -R 0 one."""
    ref, reads, args, want_r0 = G.stage_scenario(name, str(tmp_path))
    p = subprocess.run([cli, "aln", "-N", "--batch", "5"] + args + [ref, reads], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(p.stdout) == G.strip_pg(G.golden_full(name))
    assert G.strip_pg(G.golden_full(name)) != G.strip_pg(want_r0)
    os.remove(ref + ".sa")
    q = subprocess.run([cli, "aln", "-N"] + args + [ref, reads], capture_output=True, text=True)
    assert q.returncode == 0 and "stage 4" in q.stderr
    assert G.strip_pg(q.stdout) == G.strip_pg(want_r0)


def test_parse_only_ends(cli, tmp_path):
    """--parse-only (read + parse, no alignment: the host-side rate of tools/cli_bench.py) must come back -- it once
    left the per-device submit threads waiting -- and must write no alignments."""
    ref, reads, args, _ = G.stage_scenario("c2_pacbio", str(tmp_path))
    out = str(tmp_path / "out.sam")
    p = subprocess.run([cli, "aln", "-N", "-t", "3", "--batch", "7", "--parse-only", "-o", out] + args + [ref, reads], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "100 reads" in p.stderr
    assert all(l.startswith("@") for l in open(out).read().splitlines())


def _fake_mapper(d, map_src, pieces=6, delay=0.25, rc=0, stop_after=None):
    """A stand-in for the bundle's gem-mapper: writes the prepared map to `<-o PREFIX>.map` in pieces, slowly."""
    os.makedirs(d, exist_ok=True)
    exe = os.path.join(d, "gem-mapper")
    with open(exe, "w") as f:
        f.write("""#!/usr/bin/env python3
import sys, time
a = sys.argv[1:]
out = a[a.index("-o") + 1] + ".map"
data = open(%r, "rb").read()
n = %d; cut = [len(data) * i // n for i in range(n + 1)]
with open(out, "wb") as g:
    for i in range(n):
        if %r is not None and i >= %r:
            break
        g.write(data[cut[i]:cut[i + 1]]); g.flush(); time.sleep(%f)
sys.exit(%d)
""" % (map_src, pieces, stop_after, stop_after, delay, rc))
    os.chmod(exe, 0o755)
    return d


def test_map_is_read_while_the_mapper_writes_it(cli, tmp_path):
    """Seeding overlapped with the rest: the mapper is started and left running, its map is read as it
    grows -- here a stand-in that writes the fixture's map in six slow pieces, cutting lines in the middle.  Same SAM; `--seed-first`
    (wait for the mapper, as the reference does) too; a mapper that dies half way is an error, not a short output."""
    ref, reads, args, gold = G.stage_scenario("c2_pacbio", str(tmp_path))
    keep = str(tmp_path / "map.keep")
    shutil.move(reads + ".seed.gem.map", keep)
    gem = _fake_mapper(str(tmp_path / "gem"), keep)
    for extra in ([], ["--seed-first"]):
        p = subprocess.run([cli, "aln", "-R", "0", "-t", "2", "--batch", "9", "--gem-dir", gem] + extra + args + [ref, reads], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        assert "gem-mapper done!" in p.stderr
        assert G.strip_pg(p.stdout) == G.strip_pg(gold)
        os.remove(reads + ".seed.gem.map")
    gem = _fake_mapper(str(tmp_path / "gem2"), keep, rc=3, stop_after=3)
    p = subprocess.run([cli, "aln", "-R", "0", "--batch", "9", "--gem-dir", gem] + args + [ref, reads], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and ("exit abnormally" in p.stderr or "does not match" in p.stderr)


def test_an_error_exit_does_not_leave_the_mapper_running(cli, tmp_path):
    """run_aln leaves through an error (here: the index does not load) while the mapper started before it is still writing: the mapper is
    told to stop and waited for -- no orphan burning -t cores and growing <reads>.seed.gem.map -- and its end is reported."""
    import time
    ref, reads, args, _ = G.stage_scenario("c2_pacbio", str(tmp_path))
    keep = str(tmp_path / "map.keep")
    shutil.move(reads + ".seed.gem.map", keep)
    gem = _fake_mapper(str(tmp_path / "gem"), keep, pieces=40, delay=0.5)          # would run for 20 s
    os.remove(ref + ".pac")                                                          # load_index fails after the mapper has started
    t0 = time.time()
    p = subprocess.run([cli, "aln", "-R", "0", "--gem-dir", gem] + args + [ref, reads], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "gem-mapper stopped" in p.stderr, p.stderr[-1500:]
    assert time.time() - t0 < 15
    size = os.path.getsize(reads + ".seed.gem.map") if os.path.exists(reads + ".seed.gem.map") else 0
    time.sleep(1.5)
    assert (os.path.getsize(reads + ".seed.gem.map") if os.path.exists(reads + ".seed.gem.map") else 0) == size, "the mapper is still writing"


def test_chunks_dealt_over_several_devices(cli, tmp_path):
    """--devices: one handle per listed device, chunks round-robin, output in input order (two handles on device 0 here)."""
    ref, reads, args, _ = G.stage_scenario("c7_rescue", str(tmp_path))
    p = subprocess.run([cli, "aln", "-N", "--batch", "2", "--devices", "0,0,0"] + args + [ref, reads], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(p.stdout) == G.strip_pg(G.golden_full("c7_rescue"))


@pytest.mark.parametrize("name", G.SCENARIOS)
def test_default_run_equals_reference_default_run(cli, name, tmp_path):
    """No -R on the command line (stage 4 on, -R 300): the reference's default output -- golden_full.sam where stage 4
    changed something, golden_R0.sam where the reference's two runs were identical."""
    ref, reads, args, want_r0 = G.stage_scenario(name, str(tmp_path))
    want = G.golden_full(name) if name in G.RESCUE_SCENARIOS else want_r0
    p = subprocess.run([cli, "aln", "-N"] + args + [ref, reads], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert G.strip_pg(p.stdout) == G.strip_pg(want)


def test_fm_index_queries_against_brute_force(tmp_path):
    """The FM-index code of stage 4 (occurrence counts, backward search, suffix-array lookup on the reference's .bwt/.sa):
    hit counts and positions of 3 000 k-mers (k = 19 and k = 12, present and absent) equal a brute-force table of both strands."""
    exe = str(tmp_path / "fm_check")
    host = os.path.join(G.ROOT, "lamsa_amd", "host")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", host, "-I", os.path.join(G.ROOT, "include"), "-o", exe,
                    os.path.join(G.ROOT, "tests", "host", "fm_check.cpp"), os.path.join(host, "rescue.cpp"), os.path.join(host, "lamsa_host.cpp"),
                    os.path.join(G.ROOT, "tests", "emu", "emu_api.cpp"), os.path.join(G.ROOT, "tests", "emu", "emu_capi.cpp"),
                    "-I", os.path.join(G.ROOT, "tests", "emu"), "-I", os.path.join(G.ROOT, "lamsa_amd", "csrc"), "-lz", "-lpthread"], check=True)
    ref, _, _, _ = G.stage_scenario("c7_rescue", str(tmp_path))
    for k, n in ((19, 2000), (12, 1000)):
        p = subprocess.run([exe, ref, str(k), str(n)], capture_output=True, text=True)
        assert p.returncode == 0 and p.stdout.startswith("ok"), p.stdout + p.stderr


def test_index_files_identical_to_the_reference(cli, tmp_path):
    """`lamsa index --no-gem`: .pac/.ann/.amb/.bwt/.sa byte-identical to the reference's `lamsa index` -- on the fixture
    reference (regenerated from its recipe; expected files = tests/golden/ref) and on a small FASTA with runs of N, other
    ambiguity codes, lower case, wrapped lines and header comments (tests/golden/index_nrich, expected files made by the
    reference); then `lamsa aln` runs on the index just built."""
    import sys
    sys.path.insert(0, os.path.join(G.ROOT, "tools"))
    import numpy as np
    import simdata
    import make_golden_reads as M
    rng = np.random.default_rng(M.REF["seed"])
    contigs = simdata.make_reference(rng, M.REF["contigs"], M.REF["repeats"])
    ref = str(tmp_path / "ref.fa")
    simdata.write_fasta(ref, [("chr%d" % (i + 1), c) for i, c in enumerate(contigs)])
    p = subprocess.run([cli, "index", "--no-gem", ref], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    for ext in (".pac", ".ann", ".amb", ".bwt", ".sa"):
        assert open(ref + ext, "rb").read() == open(os.path.join(G.GOLD, "ref", "ref.fa" + ext), "rb").read(), ext
    d = os.path.join(G.GOLD, "index_nrich")
    small = str(tmp_path / "small.fa")
    shutil.copy(os.path.join(d, "ref.fa"), small)
    p = subprocess.run([cli, "index", "--no-gem", small], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    for ext in ("pac", "ann", "amb", "bwt", "sa"):
        assert open(small + "." + ext, "rb").read() == open(os.path.join(d, "expected." + ext), "rb").read(), ext
    # the suffixes sorted in many small blocks by several threads, and from the .pac alone: the same bytes
    many = str(tmp_path / "many.fa")
    shutil.copy(ref, many)
    p = subprocess.run([cli, "index", "--no-gem", many], capture_output=True, text=True, env=dict(os.environ, LAMSA_INDEX_BLOCK="3000", LAMSA_INDEX_THREADS="3"))
    assert p.returncode == 0, p.stderr
    os.remove(many); os.remove(many + ".bwt"); os.remove(many + ".sa")
    p = subprocess.run([cli, "index", "--from-pac", many], capture_output=True, text=True, env=dict(os.environ, LAMSA_INDEX_BLOCK="70000", LAMSA_INDEX_THREADS="5"))
    assert p.returncode == 0, p.stderr
    for ext in (".bwt", ".sa"):
        assert open(many + ext, "rb").read() == open(os.path.join(G.GOLD, "ref", "ref.fa" + ext), "rb").read(), ext
    # the index just built serves `lamsa aln` (default run, stage 4 on)
    os.makedirs(str(tmp_path / "s"))
    _, reads, args, _ = G.stage_scenario("c7_rescue", str(tmp_path / "s"))
    shutil.copy(reads, str(tmp_path / "reads.fa")); shutil.copy(reads + ".seed.gem.map", str(tmp_path / "reads.fa.seed.gem.map"))
    q = subprocess.run([cli, "aln", "-N"] + args + [ref, str(tmp_path / "reads.fa")], capture_output=True, text=True)
    assert q.returncode == 0, q.stderr[-2000:]
    assert G.strip_pg(q.stdout) == G.strip_pg(G.golden_full("c7_rescue"))


REF_BIN = os.path.join(reflib.ROOT, "oracle", "_ref", "lamsa")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="the compiled reference (oracle/_ref/lamsa) is only built where /root/reference exists")
def test_index_of_a_repeat_rich_text_against_the_reference_binary(cli, tmp_path):
    """`lamsa index` block by block against the reference's own builder (BWT-SW, src/bwt_gen.c) on a fresh 6 Mbp text with planted repeat
    families, long exact repeats and a run of one base (deep suffix comparisons): .bwt and .sa byte for byte.  (The same comparison on a
    120 Mbp text -- 240 M suffixes -- is recorded in profiles/r04_index.txt.)"""
    import sys
    sys.path.insert(0, os.path.join(G.ROOT, "tools"))
    import numpy as np
    import simdata
    rng = np.random.default_rng(77)
    contigs = simdata.make_reference(rng, [2_000_000, 2_500_000, 1_500_000], [(300, 1500), (1000, 300), (6000, 40)])
    contigs[1][100_000:130_000] = contigs[0][500_000:530_000]                    # a 30 kbp exact copy
    contigs[2][200_000:205_000] = 0                                             # 5 000 x A
    contigs[2][300_000:304_000] = np.tile(np.array([0, 3], np.uint8), 2000)     # (AT)n
    for d in ("ours", "theirs"):
        os.makedirs(str(tmp_path / d))
        simdata.write_fasta(str(tmp_path / d / "ref.fa"), [("chr%d" % (i + 1), c) for i, c in enumerate(contigs)])
    p = subprocess.run([cli, "index", "--no-gem", str(tmp_path / "ours" / "ref.fa")], capture_output=True, text=True, env=dict(os.environ, LAMSA_INDEX_BLOCK="2000000"))
    assert p.returncode == 0, p.stderr
    q = subprocess.run([REF_BIN, "index", str(tmp_path / "theirs" / "ref.fa")], capture_output=True, text=True, timeout=1500)
    assert os.path.exists(str(tmp_path / "theirs" / "ref.fa.sa")), q.stderr[-2000:]
    for ext in ("pac", "ann", "amb", "bwt", "sa"):
        assert open(str(tmp_path / "ours" / "ref.fa.") + ext, "rb").read() == open(str(tmp_path / "theirs" / "ref.fa.") + ext, "rb").read(), ext


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="the compiled reference (oracle/_ref/lamsa) is only built where /root/reference exists")
@pytest.mark.parametrize("workload,n_reads", [("ont10k", 24), ("pb5k", 32), ("sv10k", 32), ("mol5k", 32)])
def test_bench_shaped_reads_against_the_reference_binary(cli, workload, n_reads, tmp_path):
    """The bench's own inputs -- simulated reads and seed hits of a bench workload against a repeat-planted stand-in, written as the files
    `lamsa aln` reads -- through the REFERENCE binary and through the product's host program over the (emulated) device sources: the same
    SAM.  tests/test_cli_gpu.py does the same with the HIP library at 200-500 reads per workload, bench.py on its CPU-baseline sample."""
    import shutil
    import sys
    sys.path.insert(0, reflib.ROOT); sys.path.insert(0, os.path.join(reflib.ROOT, "tools"))
    import bench
    import simbatch
    import simfiles
    simbatch.build()
    wl = bench.WORKLOADS[workload]
    ref = simbatch.SimRef(120_000_000, n_contigs=6, seed=5, threads=8)
    B = simbatch.SimBatch(ref, n_reads, wl["length"], wl["profile"], seed=99, threads=8)
    d = str(tmp_path)
    simfiles.write_index(d + "/ref.fa", ref)
    for ext in ("bwt", "sa"):
        shutil.copy(os.path.join(reflib.ROOT, "tests", "golden", "ref", "ref.fa." + ext), d + "/ref.fa." + ext)
    p = simbatch.PROFILES[wl["profile"]]
    simfiles.write_reads(d + "/reads.fa", B, seed_len=50, seed_step=p["seed_step"])
    with open(d + "/reads.fa.seed.info", "w") as f:
        for r in range(n_reads):
            f.write("r%d %d %d %d\n" % (r, int(B.seed_all[r]), int(B.last_len[r]), int(B.read_off[r + 1] - B.read_off[r])))
    args = [] if wl["read_type"] == "default" else ["-T", wl["read_type"]]
    for k, v in wl["over"].items():
        args += [{"band_w": "-w", "SV_len_thd": "-V"}[k], str(v)]
    want = subprocess.run([REF_BIN, "aln"] + args + ["-t", "4", "-N", "-I", "-R", "0", "-o", d + "/out.sam", d + "/ref.fa", d + "/reads.fa"], capture_output=True, text=True, timeout=900)
    assert want.returncode == 0, want.stderr[-2000:]
    res = bench.compare_with_product(d, args, 4, n_reads, exe=cli)
    assert res == "%d/%d reads" % (n_reads, n_reads), res


@pytest.mark.parametrize("name,n", [("c3_ont", 2), ("c2_pacbio", 3), ("c5_sv", 5)])
def test_shards_concatenate_to_the_unsharded_output(cli, name, n, tmp_path):
    """--shard i/N: every process aligns one contiguous part of the read stream with its own parser (the read file cut by bytes, its
    part of the GEM map found through the first read's name; a hit stream cut by chunks; compressed reads by record count); the
    outputs of the shards written one after the other are the unsharded SAM, byte for byte (the reference prints in input order,
    src/lamsa_aln.c:1102-1110)."""
    ref, reads, args, gold = G.stage_scenario(name, str(tmp_path))
    base = [cli, "aln", "-N", "-R", "0", "--batch", "16"] + args
    whole = subprocess.run(base + [ref, reads], capture_output=True, text=True)
    assert whole.returncode == 0 and G.strip_pg(whole.stdout) == G.strip_pg(gold)
    parts = []
    for i in range(n):
        p = subprocess.run(base + ["--shard", "%d/%d" % (i, n), ref, reads], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        assert (i == 0) == p.stdout.startswith("@"), "only shard 0 writes the header"
        parts.append(p.stdout)
    assert all(len(x) > 0 for x in parts)
    assert G.strip_pg("".join(parts)) == G.strip_pg(whole.stdout)
    # the hit stream, cut by chunks
    hits = str(tmp_path / "hits.bin")
    one = subprocess.run(base + ["--save-hits", hits, ref, reads], capture_output=True, text=True)
    assert one.returncode == 0
    parts = [subprocess.run(base + ["--hits", hits, "--shard", "%d/%d" % (i, n), ref, reads], capture_output=True, text=True) for i in range(n)]
    assert all(p.returncode == 0 for p in parts), [p.stderr[-500:] for p in parts]
    assert G.strip_pg("".join(p.stdout for p in parts)) == G.strip_pg(whole.stdout)
    # compressed reads, cut by record count
    gz = str(tmp_path / "reads.fa.gz")
    with open(reads, "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    parts = [subprocess.run(base + ["--seed-result", reads + ".seed.gem.map", "--shard", "%d/%d" % (i, n), ref, gz], capture_output=True, text=True) for i in range(n)]
    assert all(p.returncode == 0 for p in parts), [p.stderr[-500:] for p in parts]
    assert G.strip_pg("".join(p.stdout for p in parts)) == G.strip_pg(whole.stdout)
    bad = subprocess.run(base + ["--shard", "3/3", ref, reads], capture_output=True, text=True)
    assert bad.returncode != 0


@pytest.mark.parametrize("qfirst", [">", "@"])
def test_shards_of_a_fastq_whose_quality_lines_begin_like_headers(cli, qfirst, tmp_path):
    """--shard on an uncompressed FASTQ: '>' (Phred 29) and '@' (Phred 31) are legal first characters of a quality line, and the byte cut of
    the read file must not take such a line for a record start (kseq, src/kseq.h:179-225, consumes the quality by length).  Without -C the
    SAM of a mapped read prints '*' for QUAL, so the shards concatenate to the FASTA run's output for every read that maps; unmapped
    reads print their quality, so the comparison is against the unsharded FASTQ run."""
    ref, reads, args, gold = G.stage_scenario("c3_ont", str(tmp_path))
    recs, name, seq = [], None, []
    for line in open(reads):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(seq)))
            name, seq = line[1:], []
        else:
            seq.append(line)
    recs.append((name, "".join(seq)))
    fq = str(tmp_path / "reads.fq")
    with open(fq, "w") as f:
        for nm, sq in recs:
            f.write("@%s\n%s\n+\n%s\n" % (nm, sq, qfirst + "I" * (len(sq) - 1)))
    shutil.copy(reads + ".seed.gem.map", fq + ".seed.gem.map")
    base = [cli, "aln", "-N", "-R", "0", "--batch", "16"] + args
    whole = subprocess.run(base + [ref, fq], capture_output=True, text=True)
    assert whole.returncode == 0, whole.stderr[-2000:]
    for n in (2, 3, 7):
        parts = [subprocess.run(base + ["--shard", "%d/%d" % (i, n), ref, fq], capture_output=True, text=True) for i in range(n)]
        assert all(p.returncode == 0 for p in parts), [p.stderr[-500:] for p in parts]
        assert G.strip_pg("".join(p.stdout for p in parts)) == G.strip_pg(whole.stdout), n


def test_shard_entry_is_not_fooled_by_a_longer_read_name(cli, tmp_path):
    """The shard's first line in the GEM map is found by the first read's name; a read whose name has that name plus '_' as a prefix
    ("r7_x" before "r7") must not be taken for it, and a shard whose first read is shorter than a seed takes the next read's lines."""
    ref, reads, args, gold = G.stage_scenario("c3_ont", str(tmp_path))
    text = open(reads).read().split(">")[1:]
    names = [t.split("\n", 1)[0] for t in text]
    mid = len(names) // 2
    # rename: the read just before the middle gets the middle read's name plus "_x"; a 20-base read without seeds is put at the middle
    new_name = {names[mid - 1]: names[mid] + "_x"}
    with open(reads, "w") as f:
        for k, t in enumerate(text):
            nm, body = t.split("\n", 1)
            if k == mid:
                f.write(">tiny\nACGTACGTACGTACGTACGT\n")
            f.write(">%s\n%s" % (new_name.get(nm, nm), body))
    lines = open(reads + ".seed.gem.map").read().split("\n")
    with open(reads + ".seed.gem.map", "w") as f:
        for ln in lines:
            if ln:
                nm, rest = ln.rsplit("_", 1) if "\t" not in ln.split("_")[-1][:0] else (None, None)
                head, tail = ln.split("\t", 1)
                rn, seed = head.rsplit("_", 1)
                f.write("%s_%s\t%s\n" % (new_name.get(rn, rn), seed, tail))
    base = [cli, "aln", "-N", "-R", "0", "--batch", "16"] + args
    whole = subprocess.run(base + [ref, reads], capture_output=True, text=True)
    assert whole.returncode == 0, whole.stderr[-2000:]
    size = os.path.getsize(reads)
    pos = open(reads).read().index(">tiny")
    # choose shard counts whose cut falls at or just before the tiny read, so that it is a shard's first record
    tried = 0
    for n in range(2, 40):
        cuts = [size // n * i for i in range(1, n)]
        if any(pos - 200 <= c <= pos for c in cuts):
            parts = [subprocess.run(base + ["--shard", "%d/%d" % (i, n), ref, reads], capture_output=True, text=True) for i in range(n)]
            assert all(p.returncode == 0 for p in parts), (n, [p.stderr[-500:] for p in parts])
            assert G.strip_pg("".join(p.stdout for p in parts)) == G.strip_pg(whole.stdout), n
            tried += 1
            if tried >= 2:
                break
    parts = [subprocess.run(base + ["--shard", "%d/2" % i, ref, reads], capture_output=True, text=True) for i in range(2)]
    assert all(p.returncode == 0 for p in parts) and G.strip_pg("".join(p.stdout for p in parts)) == G.strip_pg(whole.stdout)
