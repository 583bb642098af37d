#!/bin/bash
# Where does k_chain1's HBM traffic come from?  Builds of the library that leave chain_first after the MIN pass (s1), the main pass and
# the son lists (s2), branch tracking (s3), the line loop (s4) -- lamsa_amd/lib/var/lib_s*.so, made with -DHP_CHAIN_STOP=n -- and the
# full library, each profiled for kernel time, FETCH_SIZE and WRITE_SIZE (run on the GPU box):  tools/chain_stops.sh <tag> <bench args>
tag=$1; shift
for v in s0 s1 s2 s3 s4 full; do
  if [ $v = full ]; then unset LAMSA_HP_LIB; else export LAMSA_HP_LIB=$PWD/lamsa_amd/lib/var/lib_$v.so; fi
  PMC_SETS="FETCH_SIZE;WRITE_SIZE" bash tools/prof_bench.sh ${tag}_$v "$@" || exit 1
  echo "variant $v done"
done
