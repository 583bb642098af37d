#!/bin/bash
# Which lines of the device sources (lamsa_amd/csrc/hp_*.h) does the CPU test suite execute?  The tests' lane-emulation build
# (tests/emu) compiles those very sources with g++, so gcov answers it:
#   tools/device_coverage.sh [out.txt]        (default profiles/r03_device_coverage.txt)
# builds the emulation with --coverage into tests/_build_cov, runs `pytest -m "not gpu"`, and writes per file the line
# coverage and the never-executed line ranges (grouped by the function they belong to).
set -e
cd "$(dirname "$0")/.."
out=${1:-profiles/r03_device_coverage.txt}
rm -rf tests/_build_cov
LAMSA_EMU_COVERAGE=1 python -m pytest tests -q -m "not gpu" -x -k "emu or kernel or cli or lane or sort or compact or chaining or split or four or edge" > /tmp/cov_pytest.log 2>&1 || { tail -20 /tmp/cov_pytest.log; exit 1; }
cd tests/_build_cov
gcov -o . *.gcda > /dev/null 2>&1 || true        # one invocation: the counts of a header compiled into several objects are summed
cd ../..
python3 - "$out" <<'PY'
import glob, os, re, sys
out = sys.argv[1]
rows = []
for f in sorted(glob.glob("tests/_build_cov/hp_*.h.gcov")):
    name = os.path.basename(f)[:-5]
    src = open(os.path.join("lamsa_amd/csrc", name)).read().split("\n") if os.path.exists(os.path.join("lamsa_amd/csrc", name)) else []
    # a line of a template is listed once per instantiation as well: a line counts once, and as run when any listing of it ran
    seen = {}
    for line in open(f, errors="replace"):
        m = re.match(r"\s*([^:]+):\s*(\d+):(.*)", line)
        if not m:
            continue
        cnt, ln = m.group(1).strip(), int(m.group(2))
        if ln == 0 or cnt == "-":
            continue
        ran = not (cnt.startswith("#") or cnt.startswith("="))
        seen[ln] = seen.get(ln, False) or ran
    execd = sum(1 for v in seen.values() if v); miss = sum(1 for v in seen.values() if not v); missing = sorted(l for l, v in seen.items() if not v)
    # merge gcov files of the same header coming from several objects: keep the best (gcov writes one per object; take union)
    rows.append((name, execd, miss, missing, src))
# union over duplicates
by = {}
for name, e, m, missing, src in rows:
    if name not in by or m < by[name][1]:
        by[name] = (e, m, missing, src)
with open(out, "w") as w:
    w.write("# device-source line coverage of the CPU test suite (tests/emu build, gcov); produced by tools/device_coverage.sh\n")
    te = tm = 0
    for name, (e, m, missing, src) in sorted(by.items()):
        te += e; tm += m
        w.write("%-18s %5d of %5d executable lines run (%.1f%%)\n" % (name, e, e + m, 100.0 * e / max(1, e + m)))
    w.write("%-18s %5d of %5d (%.1f%%)\n\n" % ("TOTAL", te, te + tm, 100.0 * te / max(1, te + tm)))
    for name, (e, m, missing, src) in sorted(by.items()):
        if not missing:
            continue
        w.write("## %s: lines never executed\n" % name)
        # group into ranges, label with the nearest preceding function header
        rng = []; s0 = p = None
        for ln in missing:
            if p is not None and ln <= p + 2:
                p = ln; continue
            if s0 is not None: rng.append((s0, p))
            s0 = p = ln
        if s0 is not None: rng.append((s0, p))
        for a, b in rng:
            fn = ""
            for k in range(a - 1, 0, -1):
                t = src[k - 1] if k - 1 < len(src) else ""
                mm = re.match(r"^(?:template <[^>]*>\s*)?(?:HP_\w+|static|template)[^;]*?\b(\w+)\s*\(", t)
                if mm and not t.startswith(" "):
                    fn = mm.group(1); break
            first = src[a - 1].strip()[:90] if a - 1 < len(src) else ""
            w.write("  %4d-%-4d %-22s | %s\n" % (a, b, fn, first))
        w.write("\n")
print(open(out).read()[:1500])
PY
