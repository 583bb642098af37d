#!/bin/bash
# experiment (GPU box): the -DHP_PROF -DHP_PROF_FILL library's cycle report of one step (lamsa_amd/lib/liblamsa_hp_prof.so).   tools/exp_prof.sh [bench.py arguments]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LAMSA_HP_LIB=$PWD/lamsa_amd/lib/liblamsa_hp_prof.so timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --bare --sequential "$@" > gpurun_out/prof_run.json 2> gpurun_out/prof_run.err
grep "HP_PROF" gpurun_out/prof_run.err | grep -v "read [0-9]" | tail -40
