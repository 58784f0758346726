#!/bin/bash
# HBM traffic per kernel of the default bench command (two PMC passes) -> gpurun_out/pmc_quick/traffic.csv   usage: bash tools/pmc_quick.sh [bench args]
cd "$(dirname "$0")/.."
OUT=gpurun_out/pmc_quick
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --steps 1 --warmup 3 $@"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o x -- $B > /dev/null 2> $OUT/f.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o x -- $B > /dev/null 2> $OUT/w.log
python3 tools/pmc_traffic.py $(find $OUT/f -name 'x_counter_collection.csv' | head -1) $(find $OUT/w -name 'x_counter_collection.csv' | head -1) $OUT/traffic > /dev/null 2>&1
rm -rf $OUT/f $OUT/w
head -12 $OUT/traffic.csv
