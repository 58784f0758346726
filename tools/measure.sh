#!/bin/bash
# The round's measurement set on the GPU box (one MI355X): kernel statistics and HBM-traffic counters of the default bench
# command, then the bench lines of every configuration.  Everything lands in gpurun_out/measure/ ONLY; copy what is to be kept into
# profiles/ by hand (tools/keep_profiles.sh).   usage: bash tools/measure.sh [prof|all]
# The profiled runs use --warmup 3: the convolution kernel's per-station auto-tune (calls 2 and 3, pipeline.hip) is then decided
# before the profiled step, so the profiled instantiation is the one the 40-step headline settles on.
set -x
cd "$(dirname "$0")/.."
OUT=gpurun_out/measure
R=${ROUND:-r06}
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-end-to-end"   # (the profiled command makes the timed steps only: the end-to-end object's extra passes launch other instantiations of the convolution kernel, and the PMC summary keeps the largest launch per kernel NAME)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o x -- $B --steps 5 --warmup 3 > $OUT/stats_bench.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o x -- $B --steps 1 --warmup 3 > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o x -- $B --steps 1 --warmup 3 > /dev/null 2> $OUT/pmc_write.log
F=$(find $OUT/pmc_fetch -name 'x_counter_collection.csv' | head -1)
W=$(find $OUT/pmc_write -name 'x_counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py $F $W $OUT/${R}_pmc_traffic > $OUT/pmc_traffic.log 2>&1
cp $(find $OUT/stats -name 'x_kernel_stats.csv' | head -1) $OUT/${R}_rocprofv3_kernel_stats.csv
rm -rf $OUT/pmc_fetch $OUT/pmc_write; find $OUT -name '*.db' -delete; find $OUT -name '*_kernel_trace.csv' -delete
# the bench line quotes the traffic of THIS measurement (same kernel sources): point it at the fresh JSON
NRHIP_PMC_JSON=$OUT/${R}_pmc_traffic.json python3 bench.py > $OUT/bench_config2.json 2> $OUT/bench_config2.log
NRHIP_PMC_JSON=$OUT/${R}_pmc_traffic.json python3 bench.py --no-traces --no-cpu-baseline > $OUT/bench_config2_pass1_only.json 2>> $OUT/bench_config2.log
if [ "$1" = "prof" ]; then ls -la $OUT; exit 0; fi
python3 bench.py --flavour mixed > $OUT/bench_config2_mixed.json 2>> $OUT/bench_config2.log
python3 bench.py --scaling strong --no-cpu-baseline --write-expected-sha > $OUT/bench_config2_strong.json 2>> $OUT/bench_config2.log   # (records the one-rank mask hash an N-rank run must gather)
cp profiles/expected_mask_sha16.json $OUT/ 2>/dev/null
python3 bench.py --events 125000 --no-cpu-baseline > $OUT/bench_config2_125k_shard.json 2>> $OUT/bench_config2.log
python3 bench.py --config 3 --trigger pa --cpu-budget 40 > $OUT/bench_config3_pa.json 2>> $OUT/bench_config3.log
python3 bench.py --config 3 --trigger pa_adc_noise --cpu-budget 60 --events 200000 > $OUT/bench_config3_pa_adc_noise.json 2>> $OUT/bench_config3.log
# arrays: HBM traffic of one step (all launches summed) so that their bench lines carry roofline.traffic
for C in 3 5; do
  BA="python3 bench.py --config $C --no-cpu-baseline --steps 1 --warmup 1"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pf$C -o x -- $BA > /dev/null 2> $OUT/pf$C.log
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pw$C -o x -- $BA > /dev/null 2> $OUT/pw$C.log
  python3 tools/pmc_traffic.py --per-step 2 $(find $OUT/pf$C -name 'x_counter_collection.csv' | head -1) $(find $OUT/pw$C -name 'x_counter_collection.csv' | head -1) $OUT/${R}_pmc_traffic_config$C > $OUT/pmc_traffic_config$C.log 2>&1
  rm -rf $OUT/pf$C $OUT/pw$C
done
  # kernel statistics of the arrays (two passes over the list)
for C in 3 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$C -o x -- python3 bench.py --config $C --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2> $OUT/s$C.log
  cp $(find $OUT/s$C -name 'x_kernel_stats.csv' | head -1) $OUT/${R}_rocprofv3_config${C}_kernel_stats.csv
  rm -rf $OUT/s$C
done
NRHIP_PMC_JSON=$OUT/${R}_pmc_traffic_config3.json python3 bench.py --config 3 > $OUT/bench_config3.json 2> $OUT/bench_config3.log
NRHIP_PMC_JSON=$OUT/${R}_pmc_traffic_config5.json python3 bench.py --config 5 > $OUT/bench_config5.json 2> $OUT/bench_config5.log
# the shard rank 3 of 8 would get of the headline list: contiguous (shard_range) and interleaved chunks of 1000 (shard_chunks)
python3 bench.py --emulate-shard 3/8 --steps 20 > $OUT/bench_config2_shard_3of8_contiguous.json 2>> $OUT/bench_config2.log
python3 bench.py --emulate-shard 3/8/1000 --steps 20 > $OUT/bench_config2_shard_3of8_chunks1000.json 2>> $OUT/bench_config2.log
# (config 4: tools/measure_config4.sh -- priced line with its own PMC traffic, kernel statistics, the 1.25e6-event shard)
python3 bench.py --config 4 --trigger pa_adc_noise --no-cpu-baseline > $OUT/bench_config4_pa_adc_noise.json 2>> $OUT/bench_config4.log
# the general path (ARZ2020 + birefringence) on the 5-channel station: wall time and kernel statistics
python3 tools/config4_probe.py 100000 20000 > $OUT/config4_probe.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4stats -o x -- python3 tools/config4_probe.py 100000 20000 > /dev/null 2> $OUT/c4stats.log
cp $(find $OUT/c4stats -name 'x_kernel_stats.csv' | head -1) $OUT/${R}_rocprofv3_config4_kernel_stats.csv
rm -rf $OUT/c4stats
# round 4: the whole drop-in (host list -> output tables), the two-rank launch on this one GPU, the convolution kernel's phases
# (the whole drop-in, host list -> output tables, is the `end_to_end` object of the default line since round 5)
python3 bench.py --scaling strong --events 200000 --steps 3 --no-cpu-baseline --no-end-to-end --write-expected-sha > /dev/null 2>> $OUT/bench_config2.log   # (the one-rank hash of the list the two ranks share)
cp profiles/expected_mask_sha16.json $OUT/ 2>/dev/null
python3 bench.py --gpus 2 --allow-tcp --scaling strong --events 200000 --steps 3 --no-cpu-baseline > $OUT/bench_config2_two_ranks_one_gpu.json 2>> $OUT/bench_config2.log
[ -f nuradiomc_amd/lib/libnrhip_ct.so ] && { python3 tools/conv_phase_probe.py; python3 tools/conv_phase_probe.py --no-traces; python3 tools/conv_phase_probe.py --config 5 --steps 1 --events 300000; python3 tools/conv_pair_probe.py 200 5296; } > $OUT/conv_phases.log 2>&1
# where a 125 k-event shard spends its time (kernel sum vs step)
{ ROWS=14 bash tools/rocprof_quick.sh --events 125000 --steps 10; python3 bench.py --events 125000 --no-cpu-baseline --steps 20 | tail -c 900; } > $OUT/shard125k_kernels.log 2>&1
python3 tools/att_dense_probe.py 40000 > $OUT/att_dense_probe.log 2>&1
find $OUT -size +8M -delete
ls -la $OUT
tail -c 600 $OUT/bench_config2.json
