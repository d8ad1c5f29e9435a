#!/bin/bash
# Polls the GPU clock and power while a short bench runs (is the MFMA loop power-throttled?).
python -u bench.py --diffusion-steps 200 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/clock_bench.log 2>&1 &
BP=$!
for i in $(seq 1 14); do
  sleep 0.7
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr '\n' ' '
  echo
done
wait $BP
echo bench rc=$?
