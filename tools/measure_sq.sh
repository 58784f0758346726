#!/bin/bash
# Shader-sequencer counters of the default bench command (one MI355X): two rocprofv3 --pmc passes, summarised per kernel (largest
# launch) into gpurun_out/measure_sq/r03_pmc_sq_counters.csv.  Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
# per wave (MI355X_MICROARCH.md), SQ_BUSY_CYCLES per SE.
cd "$(dirname "$0")/.."
OUT=gpurun_out/measure_sq
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --steps 1 --warmup 3 $SQ_BENCH_ARGS"   # SQ_BENCH_ARGS: another workload (e.g. --config 3 --trigger pa_adc_noise --events 200000)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/a -o x -- $B > /dev/null 2> $OUT/a.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -o x -- $B > /dev/null 2> $OUT/b.log
python3 tools/pmc_counters.py $(find $OUT/a -name 'x_counter_collection.csv' | head -1) > $OUT/sq_a.csv 2> $OUT/sq_a.err
python3 tools/pmc_counters.py $(find $OUT/b -name 'x_counter_collection.csv' | head -1) > $OUT/sq_b.csv 2> $OUT/sq_b.err
rm -rf $OUT/a $OUT/b
ls -la $OUT; head -c 3000 $OUT/sq_a.csv
