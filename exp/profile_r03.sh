#!/bin/bash
# usage (on the GPU box): exp/profile_r03.sh   -- the round-3 profile set beside exp/profile_all.sh's:
#   kernel stats of the float-volume path (6 steps of 8 KITTI-size volumes) and the per-launch trace of one
#   3840x2160 D=256 step (8 pairs); raw output under gpurun_out/prof_r03x/, summaries printed to its *.txt files.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r03x
mkdir -p $out
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/float -- python3 exp/float_time.py > $out/float.log 2>&1 || { tail -n 20 $out/float.log; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $out/k4 -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-events --no-secondary --width 3840 --height 2160 --disp 256 --batch 8 > $out/k4.log 2>&1 || { tail -n 20 $out/k4.log; exit 1; }
python3 - <<PY
import csv, glob
out = "$out"
f = glob.glob(out + "/float/*/*kernel_stats.csv")[0]
with open(out + "/float_kernel_stats.txt", "w") as g:
    g.write([l for l in open(out + "/float.log").read().splitlines() if l.startswith("float volume")][-1] + "   (under the profiler)\n")
    for r in csv.DictReader(open(f)):
        if "sgmx" in r["Name"]:
            g.write("%-70s calls %4s  total %10.3f ms  avg %9.1f us\n" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
f = glob.glob(out + "/k4/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "sgmx" in r["Kernel_Name"]]
per = len(rows) // 3  # warm-up + 2 steps
last = rows[-per:]
with open(out + "/k4_kernel_trace.txt", "w") as g:
    tot = {}
    for r in last:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sgmx::", "")
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot[name.split("<")[0]] = tot.get(name.split("<")[0], 0) + us
        g.write("%-52s %9.1f\n" % (name, us))
    g.write("# sums of this step (us): " + ", ".join("%s %.0f" % kv for kv in tot.items()) + "; first launch to last end: %.1f ms\n" %
            ((int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6))
print(open(out + "/float_kernel_stats.txt").read())
print(open(out + "/k4_kernel_trace.txt").read())
PY
