#!/bin/bash
# copy the round's measurement set from gpurun_out/measure (scratch) into profiles/ (tracked): usage  bash tools/keep_profiles.sh [r03]
cd "$(dirname "$0")/.."
R=${1:-r06}
M=gpurun_out/measure
cp $M/${R}_rocprofv3_kernel_stats.csv $M/${R}_pmc_traffic.csv $M/${R}_pmc_traffic.json $M/${R}_rocprofv3_config4_kernel_stats.csv $M/${R}_pmc_traffic_config3.* $M/${R}_pmc_traffic_config5.* $M/${R}_rocprofv3_config3_kernel_stats.csv $M/${R}_rocprofv3_config5_kernel_stats.csv profiles/ 2>/dev/null
for c in config2 config2_pass1_only config2_mixed config2_strong config2_125k_shard config2_shard_3of8_contiguous config2_shard_3of8_chunks1000 config2_two_ranks_one_gpu config3 config3_pa config3_pa_adc_noise config4 config4_pa_adc_noise config5; do
  [ -s $M/bench_$c.json ] && tail -1 $M/bench_$c.json > profiles/${R}_bench_${c}_1gpu.json
done
[ -s $M/config4_probe.log ] && cp $M/config4_probe.log profiles/${R}_config4_probe.txt
[ -s $M/conv_phases.log ] && cp $M/conv_phases.log profiles/${R}_conv_phases.txt
[ -s $M/att_dense_probe.log ] && cp $M/att_dense_probe.log profiles/${R}_att_dense_probe.txt
[ -s $M/shard125k_kernels.log ] && cp $M/shard125k_kernels.log profiles/${R}_shard125k_kernels.txt
[ -s gpurun_out/measure_sq/sq_a.csv ] && { echo "# rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes, tools/measure_sq.sh) -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 3; largest launch per kernel, summed over XCDs / SEs"; echo "# pass a"; cat gpurun_out/measure_sq/sq_a.csv; echo "# pass b"; cat gpurun_out/measure_sq/sq_b.csv; } > profiles/${R}_pmc_sq_counters.csv
[ -s $M/expected_mask_sha16.json ] && cp $M/expected_mask_sha16.json profiles/
M4=gpurun_out/measure4
cp $M4/${R}_pmc_traffic_config4.csv $M4/${R}_pmc_traffic_config4.json $M4/${R}_rocprofv3_config4_array_kernel_stats.csv profiles/ 2>/dev/null
[ -s $M4/bench_config4.json ] && tail -1 $M4/bench_config4.json > profiles/${R}_bench_config4_1gpu.json
[ -s $M4/bench_config4_shard_1250000.json ] && tail -1 $M4/bench_config4_shard_1250000.json > profiles/${R}_bench_config4_shard_1250000_1gpu.json
ls -la profiles | grep $R
