#!/bin/bash
# usage: [EXTRA="--width 3840 ..."] exp/pmc_valu.sh TAG  -- one PMC pass of SQ instruction / cycle counters over one bench step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-events --no-secondary $EXTRA > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
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
