#!/bin/bash
# Two bench ranks on ONE GPU (a box with a single MI355X): RCCL refuses two ranks on the same device, so this exercises what a node
# without a working communicator does -- the vote on the TCP star fails, the collectives of the bench line (barrier, mask
# all-gather, counter sums, max of the elapsed time) go over the star (--allow-tcp; without it both ranks exit non-zero), the run
# completes with "collectives": "tcp".
cd "$(dirname "$0")/.."
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29611} WORLD_SIZE=2
A="--gpus 2 --steps 3 --warmup 1 --events 200000 --scaling strong --no-cpu-baseline --device 0 --allow-tcp"
RANK=1 LOCAL_RANK=1 timeout 600 python3 bench.py $A > /tmp/two_ranks_r1.log 2>&1 &
P=$!
RANK=0 LOCAL_RANK=0 timeout 600 python3 bench.py $A
R=$?
wait $P
echo "rank 0 rc=$R rank 1 rc=$?"; tail -3 /tmp/two_ranks_r1.log
