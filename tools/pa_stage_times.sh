#!/bin/bash
# where the trigger-ADC chain of the phased array spends its time: pa_czt_stage_kernel's launches grouped by LDS size (= convolution
# length of the stage) and grid (GPU box)    usage: bash tools/pa_stage_times.sh [bench args]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rp -o x -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 --config 3 --trigger pa_adc_noise --events 200000 "$@" > /dev/null 2> gpurun_out/rp.log
python3 - <<'PY'
import pandas as pd, glob
d = pd.read_csv(glob.glob('gpurun_out/rp/**/x_kernel_trace.csv', recursive=True)[0])
d = d[d.Kernel_Name.str.contains('pa_czt_stage_kernel')]
d['ms'] = (d.End_Timestamp - d.Start_Timestamp) * 1e-6
d['blocks'] = (d.Grid_Size_X // d.Workgroup_Size_X) * d.Grid_Size_Y
d['stage'] = d.Kernel_Name.str.extract(r'pa_czt_stage_kernel<(\d)>')[0]
g = d.groupby(['stage', 'Grid_Size_Y']).agg(calls=('ms', 'size'), total_ms=('ms', 'sum'), blocks=('blocks', 'sum'))
g['us_per_block_per_cu'] = g.total_ms * 1e3 / g.blocks * 256
print(g.to_string())
print('total ms', g.total_ms.sum())
PY
rm -rf gpurun_out/rp
