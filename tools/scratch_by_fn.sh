#!/bin/bash
# Static count of scratch stores / loads and calls per device function in the ISA that `make -C lamsa_amd/csrc asm` leaves under
# lamsa_amd/lib/asm (a call frame costs its callee-saved registers through scratch on every call): tools/scratch_by_fn.sh [file.s]
f=${1:-lamsa_amd/lib/asm/hp_align_api-hip-amdgcn-amd-amdhsa-gfx950.s}
awk '/^(_ZN2hp|_Z[0-9])[A-Za-z0-9_]*:/{fn=$1; sub(":","",fn)} /scratch_store/{st[fn]++} /scratch_load/{ld[fn]++} /s_swappc/{call[fn]++} END{for(f in st) printf "%6d st %6d ld %4d calls  %s\n", st[f], ld[f], call[f], f}' $f | sort -rn | head -${2:-40}
