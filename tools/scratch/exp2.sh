#!/bin/bash
bash tools/quick_gpu.sh $1 "test_path_gpu" || exit 1
VARIANTS="nocopy" bash tools/scratch/exp_nodp.sh
