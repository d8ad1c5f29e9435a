#!/bin/bash
# Round artifacts on the 1-GPU box: default bench line, kernel-trace summary of the same command, HBM-traffic PMC passes per
# precision regime.  usage: bash tools/round_profile.sh <tag>   (outputs under gpurun_out/<tag>_*)
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 400 python3 bench.py > $O/${TAG}_bench_default.log 2> $O/${TAG}_bench_default.err
tail -c 600 $O/${TAG}_bench_default.log; echo
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o t -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/${TAG}_rocprof_bench_stdout.log 2>&1
cp $(find $O/${TAG}_trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
python3 tools/trace_shapes.py $(find $O/${TAG}_trace -name "*kernel_trace.csv" | head -1) | tee $O/${TAG}_kernel_shapes.txt
rm -rf $O/${TAG}_trace
for P in fp16 fp16x2 fp16x3; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pmc_${P}_${C} -o p -- python3 bench.py --diffusion-steps 3 --steps 1 --warmup 0 --no-cpu-baseline --precision $P > /dev/null 2>&1
  done
  python3 tools/pmc_traffic.py $(find $O/${TAG}_pmc_${P}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/${TAG}_pmc_${P}_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/gemm_traffic_${P}.json
  rm -rf $O/${TAG}_pmc_${P}_FETCH_SIZE $O/${TAG}_pmc_${P}_WRITE_SIZE
done
# L2 hit rate / fabric request mix per kernel (tools/pmc_l2.py): what the counters can say about the re-streamed operands
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $O/${TAG}_pmc_l2 -o p -- python3 bench.py --diffusion-steps 3 --steps 1 --warmup 0 --no-cpu-baseline --precision fp16 > /dev/null 2>&1
python3 tools/pmc_l2.py $(find $O/${TAG}_pmc_l2 -name "*counter_collection.csv" | head -1) $O/${TAG}_l2_counters.json | tee $O/${TAG}_l2_counters.txt
rm -rf $O/${TAG}_pmc_l2
# the other BASELINE configurations as bench lines (single GPU): configs[2] at its per-GPU shape, configs[3] shortened to T = 100
# (x 20 for T = 2000), configs[4] = the training step
timeout -k 10 300 python3 bench.py --preset configs2 --no-cpu-baseline > $O/${TAG}_bench_configs2.log 2>&1
timeout -k 10 400 python3 bench.py --preset configs3 --diffusion-steps 100 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_configs3_T100.log 2>&1
timeout -k 10 400 python3 bench.py --preset configs4 --steps 5 --warmup 2 > $O/${TAG}_bench_configs4.log 2>&1
tail -c 300 $O/${TAG}_bench_configs2.log; echo; tail -c 300 $O/${TAG}_bench_configs4.log; echo
