#!/bin/bash
# usage: ab_env.sh VAR=val
for i in 1 2 3; do
  for opt in "$1" "DIDI_NOOP=0"; do
    env "$opt" timeout -k 10 300 python -u bench.py --diffusion-steps 200 --steps 2 --warmup 1 --no-cpu-baseline --precision fp16 > gpurun_out/ab_tmp.log 2>&1 || { echo failed; tail -3 gpurun_out/ab_tmp.log; exit 1; }
    echo "$opt $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_tmp.log)"
  done
done
