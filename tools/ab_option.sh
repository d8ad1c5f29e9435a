#!/bin/bash
# Same-box A/B of one tuning switch: alternates bench runs between DIDI_<OPTION>=<value> and the default.
# usage: tools/ab_option.sh <OPTION>=<value> [precision] [diffusion steps] [timed steps]
P=${2:-fp16}
T=${3:-200}
K=${4:-2}
for i in 1 2 3; do
  for opt in "DIDI_$1" "DIDI_NOOP=0"; do
    env "$opt" timeout -k 10 300 python -u bench.py --diffusion-steps $T --steps $K --warmup 1 --no-cpu-baseline --precision $P > gpurun_out/ab_tmp.log 2>&1 || { echo failed; tail -3 gpurun_out/ab_tmp.log; exit 1; }
    echo "$opt $P T=$T $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_tmp.log)"
  done
done
