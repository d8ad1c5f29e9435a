#!/bin/bash
# ms per denoiser call at several batch sizes under the GEMM-form knobs given in the environment.
for b in "$@"; do
  python -u bench.py --batch $b --diffusion-steps 100 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/sweep_tmp.log 2>&1 || { echo "B=$b failed"; tail -3 gpurun_out/sweep_tmp.log; exit 1; }
  python3 -c "
import json,sys
r=json.loads(open('gpurun_out/sweep_tmp.log').read().strip().splitlines()[-1])
print('B=%4d  %.3f ms/call  %.1f dialogs/s-at-T2000' % ($b, r['ms_per_step']/100, r['value']/20))"
done
