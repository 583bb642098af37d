#!/bin/bash
# Where do k_chain1's time and HBM traffic come from?  Builds of the library that leave chain_first after the sort index and node records
# (s0), the clusters (s1: their cut only -- since round 3 the MIN pass runs inside the cluster pass of s2), the main pass with the MIN pass
# and the son lists (s2), branch tracking (s3), the line loop (s4) --
# lamsa_amd/lib/var/lib_s*.so, made with `make OUT=../lib/var/lib_s<n>.so EXTRA=-DHP_CHAIN_STOP=<n>` -- and the full library, each profiled
# for kernel time, FETCH_SIZE and WRITE_SIZE (run on the GPU box):  tools/chain_stops.sh <tag> <bench args>
tag=$1; shift
for v in s0 s1 s2 s3 s4 full; do
  if [ $v = full ]; then unset LAMSA_HP_LIB; else export LAMSA_HP_LIB=$PWD/lamsa_amd/lib/var/lib_$v.so; fi
  PMC_SETS="FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" bash tools/prof_bench.sh ${tag}_$v "$@" > /dev/null 2>&1 || exit 1
  python3 tools/summarize_prof.py gpurun_out/prof/${tag}_$v gpurun_out/${tag}_$v "chain_stops $v" > /dev/null 2>&1
  python3 - gpurun_out/${tag}_${v}_pmc.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]["k_chain1"]; p = d["per_dispatch"]
print("%-5s k_chain1 %8.3f ms  fetch %6.1f GB  write %6.1f GB  vmem rd %.4g wr %.4g  valu %.4g salu %.4g lds %.4g" % (sys.argv[2], d["avg_ms_rocprof"], p["FETCH_SIZE"] * 1024 / 1e9, p["WRITE_SIZE"] * 1024 / 1e9,
      p["SQ_INSTS_VMEM_RD"], p["SQ_INSTS_VMEM_WR"], p["SQ_INSTS_VALU"], p["SQ_INSTS_SALU"], p["SQ_INSTS_LDS"]))
PY
done
