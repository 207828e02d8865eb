#!/bin/bash
# usage: exp/ktrace.sh outname   -- per-launch kernel trace of one bench step (env switches pass through)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$1 -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-events --no-secondary $EXTRA > gpurun_out/$1.log 2>&1 || exit 1
python3 exp/ktrace_sum.py $1
