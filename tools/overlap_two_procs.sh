#!/bin/bash
# Do two station loops overlap on one GPU?  The same array bench alone, then two processes of it side by side: if the pair takes less
# than twice the single run, the tails and host round trips of one station call hide under the other's kernels.   usage: bash tools/overlap_two_procs.sh 5 200000
cd "$(dirname "$0")/.."
C=${1:-5}; N=${2:-200000}
A="--config $C --events $N --no-cpu-baseline --steps 3 --warmup 3"
python3 bench.py $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('alone   ms/step %.1f' % d['ms_per_step'])"
python3 bench.py $A > /tmp/ov_a.json 2>/dev/null &
P=$!
python3 bench.py $A > /tmp/ov_b.json 2>/dev/null
wait $P
python3 -c "
import json
for f in ('/tmp/ov_a.json','/tmp/ov_b.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print('paired  ms/step %.1f' % d['ms_per_step'])
"
