import csv, glob, sys
f = sorted(glob.glob("gpurun_out/ovl/**/*_kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "sgmx" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-44:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-28s q%-3s %9.1f -> %9.1f  (%7.1f us)" % (r["Kernel_Name"][11:39], r.get("Queue_Id", "?"), s / 1e3, e / 1e3, (e - s) / 1e3))
