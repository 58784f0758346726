#!/bin/bash
# usage (GPU box): bash tools/variants.sh lib1.so lib2.so ...   -- bench.py (config 2, 8 steps) per library variant: stage times and triggers
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for L in "$@"; do
  NRHIP_LIB_NAME=$L python3 bench.py --no-cpu-baseline --steps ${STEPS:-8} $BENCH_ARGS > gpurun_out/v_$L.json 2> gpurun_out/v_$L.err
  python3 - "$L" <<'PY'
import json, sys
L = sys.argv[1]
try:
    d = json.loads(open('gpurun_out/v_%s.json' % L).read().strip().splitlines()[-1])
    print(L, 'ms/step %.2f' % d['ms_per_step'], {k: round(v, 2) for k, v in d['config']['stage_ms_avg_per_step'].items()}, 'triggers', d['config']['n_triggered_all'])
except Exception as e:
    print(L, 'FAILED', e, open('gpurun_out/v_%s.err' % L).read()[-1500:])
PY
done
