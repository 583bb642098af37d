#!/usr/bin/env python3
"""Apply the edits of INTEGRATION.md section 3 to a SCRATCH COPY of the reference's src/ directory.

    python tools/apply_glue.py <scratch>/src

The edits are located by short anchors (a function name, a comment the reference carries) and everything between
anchors is kept from the scratch copy itself, so no reference text lives in this script.  Each anchor must match exactly
once; otherwise the reference has moved and the script stops.  After it ran, `gcc -DLAMSA_HP -I<repo>/include
-I<repo>/lamsa_amd/glue ...` builds a `lamsa` whose stages (2),(3),(2'),(3') go through include/lamsa_hp.h.
Used by tests/test_glue_cpu.py (which links the CPU emulation of the C-ABI) -- the reference itself is never modified.
"""
import re
import sys


def once(text, pattern, what):
    m = list(re.finditer(pattern, text, re.M))
    if len(m) != 1:
        sys.exit("apply_glue: anchor for %s matched %d times" % (what, len(m)))
    return m[0]


def main(src):
    path = src + "/lamsa_aln.c"
    t = open(path).read()

    # (1) the binding, right after the private typedefs it needs
    m = once(t, r"^\} thread_aux_t;.*$", "thread_aux_t")
    t = t[:m.end()] + '\n#ifdef LAMSA_HP\n#include "lamsa_hp_glue.c"\n#endif\n' + t[m.end():]

    # (2) the worker: everything from the map_msg parsing to the second frag_check is already done for the chunk
    a = once(t, r"^[ \t]*// set map_msg[ \t]*$", "worker: start of the replaced block")
    b = once(t, r"^[ \t]*// bwt aln[ \t]*$", "worker: end of the replaced block")
    old = t[a.start():b.start()]
    new = ("#ifdef LAMSA_HP\n"
           "        /* a_res[0], a_res[1] were filled by lamsa_hp_glue_chunk; what is left of the block below: */\n"
           "        aln_reg *a_reg = aln_init_reg(seqs->seq.l);\n"
           "        uint8_t *bseq = (uint8_t*)malloc(seqs->seq.l * sizeof(uint8_t)), *rbseq = NULL;\n"
           "        for (j = 0; j < (int)seqs->seq.l; ++j) bseq[j] = nst_nt4_table[(int)(seqs->seq.s[j])];\n"
           "        get_reg(la_seqs->a_res, a_reg); get_reg(la_seqs->a_res+1, a_reg);\n"
           "#else\n" + old + "#endif\n")
    t = t[:a.start()] + new + t[b.start():]

    # (3) lamsa_aln_core: open / per chunk / close
    m = once(t, r"^[ \t]*pthread_rwlock_init\(&RWLOCK, NULL\);[ \t]*$", "lamsa_aln_core: after the aux set-up")
    t = t[:m.end()] + "\n#ifdef LAMSA_HP\n    if (lamsa_hp_glue_open(AP, bns, pac, 0) != 0) exit(1);\n#endif\n" + t[m.end():]
    m = once(t, r"^[ \t]*THREAD_READ_I = 0;[ \t]*$", "lamsa_aln_core: top of the chunk loop")
    t = t[:m.end()] + "\n#ifdef LAMSA_HP\n        lamsa_hp_glue_chunk(lamsa_seqs, read_seq_t, n_seqs, AP, bns);\n#endif\n" + t[m.end():]
    m = once(t, r"^[ \t]*pthread_rwlock_destroy\(&RWLOCK\);[ \t]*$", "lamsa_aln_core: tear-down")
    t = t[:m.end()] + "\n#ifdef LAMSA_HP\n    lamsa_hp_glue_close();\n#endif\n" + t[m.end():]
    open(path, "w").write(t)

    # (4) a chunk large enough to fill the GPU (the reference reads 128 reads per chunk)
    path = src + "/lamsa_aln.h"
    h = open(path).read()
    m = once(h, r"^#define CHUNK_READ_N[ \t]+\d+[ \t]*$", "CHUNK_READ_N")
    h = h[:m.start()] + "#ifdef LAMSA_HP\n#define CHUNK_READ_N 4096\n#else\n" + m.group(0) + "\n#endif" + h[m.end():]
    open(path, "w").write(h)
    print("apply_glue: 4 edits applied in", src)


if __name__ == "__main__":
    if len(sys.argv) != 2:
        sys.exit(__doc__)
    main(sys.argv[1])
