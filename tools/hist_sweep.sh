#!/bin/bash
# usage (GPU box, repo root): bash tools/hist_sweep.sh <variant> ...  -- k_qv_hist alone (tools/microbench/hist_time.py) under each library variant, twice
for rep in 1 2; do
  for v in "$@"; do
    lib=$PWD/dextractor_amd/libdexgpu.so
    [ "$v" != main ] && lib=$PWD/tools/variants/libdexgpu_$v.so
    DEXGPU_LIB=$lib python tools/microbench/hist_time.py 2>&1 | tail -1
  done
done
