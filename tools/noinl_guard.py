#!/usr/bin/env python3
"""Guard for the HP_NOINL hazard ("Known hazards"): hipcc 7.2 once miscompiled a non-inlined device call whose caller had
handed it the address of a local -- four per-lane scratch slots written through a generic pointer in the callee and read back with
scratch_load in the caller.  The pattern is banned from the device sources: results of non-inlined routines come back by value.

    python tools/noinl_guard.py [--asm] [--report FILE]

* Source check (always; __graft_entry__.build() runs it): every call of an HP_NOINL routine in lamsa_amd/csrc whose argument list holds
  an address-of expression.  Exit status 1 when one is found that tools/noinl_allow.txt does not list: the file holds the accepted ones --
  context structures of the calling frame that the callee fills (FlStore, Trig, Regs) and members of the record being built -- and no
  scalar local (LOCAL in the report) is accepted any more.
* --asm: per non-inlined device function of the last `make -C lamsa_amd/csrc asm`: scratch bytes per lane, scratch stores / loads and
  calls in its body (a call frame goes through scratch on every call), from lamsa_amd/lib/asm.
"""
import argparse
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lamsa_amd", "csrc")
ALLOW = os.path.join(ROOT, "tools", "noinl_allow.txt")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: re.sub(r"[^\n]", " ", m.group(0)), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")))


def noinl_names(texts):
    names = set()
    for t in texts.values():
        for m in re.finditer(r"HP_NOINL\s+[\w:<>\s\*&]+?\b(\w+)\s*\(", t):
            names.add(m.group(1))
    return names


def call_args(text, pos):
    """Argument list starting at the '(' at text[pos]; returns (list of argument strings, end index)."""
    depth, args, cur, i = 0, [], "", pos
    while i < len(text):
        c = text[i]
        if c in "([{":
            depth += 1
            if depth > 1:
                cur += c
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                args.append(cur.strip())
                return args, i
            cur += c
        elif c == "," and depth == 1:
            args.append(cur.strip()); cur = ""
        else:
            cur += c
        i += 1
    return args, i


def local_scalars_before(text, pos):
    """Names declared as plain scalars (int / bool / long long / int64_t ...) between the start of the enclosing function and pos."""
    start = text.rfind("\n{", 0, pos)
    body = text[start if start >= 0 else 0:pos]
    names = set()
    for m in re.finditer(r"\b(?:const\s+)?(?:int|bool|long long|int32_t|int64_t|unsigned|size_t|float|double)\s+((?:\w+\s*(?:=[^;,]*)?,\s*)*\w+\s*(?:=[^;]*)?);", body):
        for part in re.split(r",(?![^()]*\))", m.group(1)):
            nm = re.match(r"\s*(\w+)", part)
            if nm:
                names.add(nm.group(1))
    return names


def source_check():
    texts = {f: strip_comments(open(f).read()) for f in sources()}
    names = noinl_names(texts)
    found = []
    for f, t in texts.items():
        for m in re.finditer(r"\b(\w+)\s*(?:<[^<>;(){}]*>)?\s*\(", t):
            callee = m.group(1)
            if callee not in names:
                continue
            before = t[max(0, m.start() - 80):m.start()]
            if re.search(r"HP_NOINL[^;{}]*$", before):          # the definition itself
                continue
            args, _ = call_args(t, m.end() - 1)
            line = t.count("\n", 0, m.start()) + 1
            locs = None
            for a in args:
                am = re.match(r"&\s*(\w+)\s*$", a)
                if am:
                    if locs is None:
                        locs = local_scalars_before(t, m.start())
                    kind = "LOCAL" if am.group(1) in locs else "other"
                    found.append((kind, os.path.relpath(f, ROOT), line, callee, a))
                elif a.startswith("&"):
                    found.append(("other", os.path.relpath(f, ROOT), line, callee, a))
    return names, found


def asm_report():
    sfile = os.path.join(ROOT, "lamsa_amd", "lib", "asm", "hp_align_api-hip-amdgcn-amd-amdhsa-gfx950.s")
    rfile = os.path.join(ROOT, "lamsa_amd", "lib", "asm", "resource_usage.txt")
    if not os.path.exists(sfile):
        return ["(no ISA: run `make -C lamsa_amd/csrc asm` first)"]
    st, ld, calls, order = {}, {}, {}, []
    fn = None
    for line in open(sfile):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn = m.group(1); order.append(fn)
            continue
        if fn is None:
            continue
        if "scratch_store" in line:
            st[fn] = st.get(fn, 0) + 1
        elif "scratch_load" in line:
            ld[fn] = ld.get(fn, 0) + 1
        elif "s_swappc" in line:
            calls[fn] = calls.get(fn, 0) + 1
    scratch = {}
    if os.path.exists(rfile):
        cur = None
        for line in open(rfile):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and cur:
                scratch[cur] = int(m.group(1))
    try:
        dem = subprocess.run(["c++filt"], input="\n".join(order), capture_output=True, text=True).stdout.split("\n")
    except OSError:
        dem = order
    out = ["%-7s %-6s %-6s %-6s %s" % ("scratch", "stores", "loads", "calls", "function (device, gfx950)")]
    rows = []
    for k, fn in enumerate(order):
        if not (st.get(fn) or ld.get(fn) or calls.get(fn)):
            continue
        rows.append((st.get(fn, 0) + ld.get(fn, 0), "%-7s %-6d %-6d %-6d %s" % (scratch.get(fn, "-"), st.get(fn, 0), ld.get(fn, 0), calls.get(fn, 0), dem[k][:120] if k < len(dem) else fn)))
    out += [r for _, r in sorted(rows, key=lambda x: -x[0])]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", action="store_true")
    ap.add_argument("--report")
    a = ap.parse_args()
    names, found = source_check()
    allow = set()
    if os.path.exists(ALLOW):
        allow = {l.strip() for l in open(ALLOW) if l.strip() and not l.startswith("#")}
    lines = ["HP_NOINL routines: %d" % len(names), "", "address-of arguments at their call sites (LOCAL = a scalar of the calling frame); every one must be listed in tools/noinl_allow.txt:"]
    bad = []
    for kind, f, line, callee, arg in sorted(found):
        key = "%s %s %s" % (f, callee, arg)
        ok = key in allow
        lines.append("  %-5s %s:%d  %s(... %s ...)%s" % (kind, f, line, callee, arg, "" if ok else "   <-- NOT ALLOWED"))
        if not ok:
            bad.append(key)
    if not found:
        lines.append("  none")
    if a.asm:
        lines += ["", "call frames (lamsa_amd/lib/asm, `make -C lamsa_amd/csrc asm`):"] + ["  " + l for l in asm_report()]
    text = "\n".join(lines) + "\n"
    if a.report:
        open(a.report, "w").write(text)
    else:
        sys.stdout.write(text)
    if bad:
        sys.stderr.write("noinl_guard: %d call(s) hand the address of a caller's local to a non-inlined device routine:\n  %s\n" % (len(bad), "\n  ".join(bad)))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
