#!/bin/bash
# per-kernel times of the default bench command (GPU box)   usage: bash tools/rocprof_quick.sh [bench args]; NRHIP_LIB_NAME picks a build variant
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp -o x -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 2 "$@" > /dev/null 2> gpurun_out/rp.log
python3 - <<'PY'
import pandas as pd, glob
d=pd.read_csv(glob.glob('gpurun_out/rp/**/x_kernel_stats.csv', recursive=True)[0])
d['Name']=d['Name'].str.replace(r'\(.*','',regex=True).str.slice(0,60)
print(d[['Name','Calls','AverageNs','MaxNs','Percentage']].head(int(__import__('os').environ.get('ROWS', '12'))).to_string())
PY
rm -rf gpurun_out/rp
