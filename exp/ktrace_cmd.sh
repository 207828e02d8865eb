#!/bin/bash
# usage: exp/ktrace_cmd.sh outname N script.py [args]  -- kernel trace of a python script, last N sgmx launches
name=$1; n=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$name -- python3 "$@" > gpurun_out/$name.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = glob.glob('gpurun_out/$name/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'sgmx' in r['Kernel_Name']]
for r in rows[-$n:]:
    print("  %-50s %8.1f" % (r['Kernel_Name'].split('(')[0][-50:], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
