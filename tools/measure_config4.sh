#!/bin/bash
# Config 4 (ARZ2020 + birefringence on the 35-station array) on the GPU box: HBM traffic of one step (two PMC passes, all launches
# summed), kernel statistics, the priced bench line, and a full 1.25e6-event shard of BASELINE configs[3] (1e7 events over 8 GPUs).
# Everything lands in gpurun_out/measure4/; copy what is to be kept into profiles/.   usage: bash tools/measure_config4.sh [noshard]
set -x
cd "$(dirname "$0")/.."
OUT=gpurun_out/measure4
R=${ROUND:-r06}
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --config 4 --no-cpu-baseline --steps 1 --warmup 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o x -- $B > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o x -- $B > /dev/null 2> $OUT/pmc_write.log
F=$(find $OUT/pmc_fetch -name 'x_counter_collection.csv' | head -1)
W=$(find $OUT/pmc_write -name 'x_counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py --per-step 1 $F $W $OUT/${R}_pmc_traffic_config4 > $OUT/pmc_traffic.log 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o x -- $B > $OUT/stats_bench.json 2> $OUT/stats.log
cp $(find $OUT/stats -name 'x_kernel_stats.csv' | head -1) $OUT/${R}_rocprofv3_config4_array_kernel_stats.csv
rm -rf $OUT/stats; find $OUT -name '*.db' -delete
NRHIP_PMC_JSON=$OUT/${R}_pmc_traffic_config4.json python3 bench.py --config 4 --cpu-budget 120 > $OUT/bench_config4.json 2> $OUT/bench_config4.log
if [ "$1" != "noshard" ]; then
  python3 bench.py --config 4 --events 1250000 --warmup 0 --steps 1 --no-cpu-baseline > $OUT/bench_config4_shard_1250000.json 2> $OUT/bench_config4_shard.log
fi
ls -la $OUT
head -8 $OUT/${R}_pmc_traffic_config4.csv
tail -c 700 $OUT/bench_config4.json
tail -c 1500 $OUT/bench_config4_shard_1250000.json
