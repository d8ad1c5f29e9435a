#!/bin/bash
# Per-kernel evidence table (duration, HBM bytes, MFMA busy share) of the fp16 regime: four separate rocprofv3 passes over the same
# short command.  usage: bash tools/per_kernel_profile.sh <tag> [precision]
set -e
TAG=${1:-r03}
P=${2:-fp16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
CMD="bench.py --diffusion-steps 3 --steps 1 --warmup 0 --no-cpu-baseline --precision $P"
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_pk_trace -o t -- python3 $CMD > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pk_fetch -o p -- python3 $CMD > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pk_write -o p -- python3 $CMD > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pk_sq -o p -- python3 $CMD > /dev/null 2>&1
python3 tools/per_kernel_counters.py $(find $O/${TAG}_pk_trace -name "*kernel_trace.csv" | head -1) $(find $O/${TAG}_pk_fetch -name "*counter_collection.csv" | head -1) \
  $(find $O/${TAG}_pk_write -name "*counter_collection.csv" | head -1) $(find $O/${TAG}_pk_sq -name "*counter_collection.csv" | head -1) $O/${TAG}_per_kernel_counters_${P}.json
rm -rf $O/${TAG}_pk_trace $O/${TAG}_pk_fetch $O/${TAG}_pk_write $O/${TAG}_pk_sq
