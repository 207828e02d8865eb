#!/bin/bash
# usage: [EXTRA="--width 3840 ..."] exp/pmc_cache.sh TAG -- cache-side counters of one bench step, one pass per block
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); out=gpurun_out/pmcc_$1_$i
  timeout -s KILL 90 rocprofv3 --pmc $set --output-format csv -d $out -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-events --no-secondary $EXTRA > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
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
done
