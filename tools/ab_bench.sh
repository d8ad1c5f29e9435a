#!/bin/bash
# Same-box A/B of two library builds: alternates bench runs (T=200) between DIDI_HIP_LIB=$1 and the in-tree library.
# usage: tools/ab_bench.sh <other.so> [precision] [diffusion steps] [timed steps]
P=${2:-fp16x2}
T=${3:-200}
K=${4:-2}
for i in 1 2 3; do
  for lib in "$1" ""; do
    DIDI_HIP_LIB=$lib timeout -k 10 300 python -u bench.py --diffusion-steps $T --steps $K --warmup 1 --no-cpu-baseline --precision $P > gpurun_out/ab_tmp.log 2>&1 || { echo failed; tail -3 gpurun_out/ab_tmp.log; exit 1; }
    echo "${lib:-in-tree} $P $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_tmp.log)"
  done
done
