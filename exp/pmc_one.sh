#!/bin/bash
# usage: [EXTRA="--batch 1"] exp/pmc_one.sh TAG "COUNTERS" -- one PMC pass over one bench step, per-launch values of the last launch group
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc1_$1
timeout -s KILL 90 rocprofv3 --pmc $2 --output-format csv -d $out -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-events --no-secondary $EXTRA > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob('$out/*/*counter_collection.csv')[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'sgmx' not in r['Kernel_Name']:
        continue
    k = r['Kernel_Name'].split('(')[0].split('::')[-1][:34] + '#' + r['Dispatch_Id']
    d = agg.setdefault(k, collections.OrderedDict())
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    d['us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, v in list(agg.items())[-11:]:
    print(k, ' '.join('%s=%.4g' % (a, b) for a, b in v.items()))
PY
