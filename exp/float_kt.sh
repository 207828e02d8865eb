#!/bin/bash
# kernel stats of the float-volume path (exp/float_time.py) under rocprofv3: usage exp/float_kt.sh name
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/$1 -- python3 exp/float_time.py > gpurun_out/r03/$1.log 2>&1 || { tail -n 20 gpurun_out/r03/$1.log; exit 1; }
tail -n 1 gpurun_out/r03/$1.log
python3 - <<PY
import csv, glob
f = glob.glob('gpurun_out/r03/$1/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'sgmx' in r['Name']:
        print("  %-60s calls %4s  avg %10.1f us" % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs']) / 1e3))
PY
