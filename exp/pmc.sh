#!/bin/bash
# usage: exp/pmc.sh outname "COUNTER1 COUNTER2 ..."   (one pass; bench with 1 step)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/$1 -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-events > gpurun_out/$1.log 2>&1
