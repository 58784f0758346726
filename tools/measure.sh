#!/bin/bash
# The round's measurement set on the GPU box (one MI355X): kernel statistics and HBM-traffic counters of the default bench
# command, then the bench lines of every configuration.  Everything lands in gpurun_out/measure/; copy what is to be kept into
# profiles/.   usage: bash tools/measure.sh [tag]
set -x
cd "$(dirname "$0")/.."
OUT=gpurun_out/measure
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o x -- $B --steps 5 --warmup 1 > $OUT/stats_bench.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o x -- $B --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o x -- $B --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_write.log
F=$(find $OUT/pmc_fetch -name 'x_counter_collection.csv' | head -1)
W=$(find $OUT/pmc_write -name 'x_counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py $F $W $OUT/r02_pmc_traffic > $OUT/pmc_traffic.log 2>&1
cp $OUT/r02_pmc_traffic.json profiles/r02_pmc_traffic.json   # so that the bench line below quotes it (same sources)
cp $(find $OUT/stats -name 'x_kernel_stats.csv' | head -1) $OUT/r02_rocprofv3_kernel_stats.csv
rm -rf $OUT/pmc_fetch $OUT/pmc_write; find $OUT -name '*.db' -delete; find $OUT -name '*_kernel_trace.csv' -delete
python3 bench.py > $OUT/bench_config2.json 2> $OUT/bench_config2.log
if [ "$1" = "prof" ]; then ls -la $OUT; exit 0; fi
python3 bench.py --flavour mixed > $OUT/bench_config2_mixed.json 2>> $OUT/bench_config2.log
python3 bench.py --scaling strong --no-cpu-baseline > $OUT/bench_config2_strong.json 2>> $OUT/bench_config2.log
python3 bench.py --config 3 > $OUT/bench_config3.json 2> $OUT/bench_config3.log
python3 bench.py --config 5 > $OUT/bench_config5.json 2> $OUT/bench_config5.log
python3 bench.py --config 4 > $OUT/bench_config4.json 2> $OUT/bench_config4.log
# the general path (ARZ2020 + birefringence) on the 5-channel station: wall time and kernel statistics
python3 tools/config4_probe.py 100000 20000 > $OUT/config4_probe.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4stats -o x -- python3 tools/config4_probe.py 100000 20000 > /dev/null 2> $OUT/c4stats.log
cp $(find $OUT/c4stats -name 'x_kernel_stats.csv' | head -1) $OUT/r02_rocprofv3_config4_kernel_stats.csv
rm -rf $OUT/c4stats
# the digitised phased array on the 35-station array (chirp-z digitiser)
python3 tools/config3_probe.py 250000 35 pa_adc > $OUT/config3_pa_adc.log 2>&1
find $OUT -size +8M -delete
ls -la $OUT
tail -c 600 $OUT/bench_config2.json
