#!/bin/bash
# Instruction-cache counters of the bench kernels (one MI355X): channel_conv_kernel's channel loop with its out-of-line transforms is
# ~55-60 KB of code, about the size of the instruction cache two CUs share -- is it fetch bound?  Round 5: no.  Config 2:
# 8.3e8 requests, 1.8e5 misses + 1.1e6 duplicate misses (0.16 %); config 5 the same rate.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/icache
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/a -o x -- python3 bench.py --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1 > /dev/null 2> $OUT/a.log
python3 tools/pmc_counters.py $(find $OUT/a -name 'x_counter_collection.csv' | head -1) > $OUT/icache_config2.csv 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/b -o x -- python3 bench.py --config 5 --events 200000 --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1 > /dev/null 2> $OUT/b.log
python3 tools/pmc_counters.py $(find $OUT/b -name 'x_counter_collection.csv' | head -1) > $OUT/icache_config5.csv 2> $OUT/b.err
rm -rf $OUT/a $OUT/b
head -c 1500 $OUT/icache_config2.csv
