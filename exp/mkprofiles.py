"""Condense the rocprofv3 outputs of exp/profile_all.sh TAG into the small files kept under profiles/:
<TAG>_kernel_stats.csv, <TAG>_pmc_{FETCH,WRITE}_SIZE.csv (sgmx kernels only), <TAG>_hbm_traffic.json,
<TAG>_bench_under_rocprof.json."""
import csv, glob, json, os, sys, collections, shutil
tag = sys.argv[1]
XOR_SHARE = float(os.environ.get("SGMX_XOR_SHARE", "0.17"))  # v_xor_b32 share of the sweep's VALU instructions (exp/step_mix.py)
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "gpurun_out", "prof_" + tag, "summary")
os.makedirs(dst, exist_ok=True)
st = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))[0]
shutil.copy(st, os.path.join(dst, tag + "_kernel_stats.csv"))
lines = [l for l in open(os.path.join(src, "bench_under_rocprof.json")) if l.startswith("{")]
open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w").write(lines[-1])
per = collections.OrderedDict()
for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if "sgmx" in r["Kernel_Name"] and r["Counter_Name"] == name]
    with open(os.path.join(dst, "%s_pmc_%s.csv" % (tag, name)), "w", newline="") as g:
        wr = csv.writer(g)
        wr.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "VGPR_Count", "Counter_Name", "Counter_Value", "Duration_ns"])
        for r in rows:
            wr.writerow([r["Dispatch_Id"], r["Kernel_Name"].split("(")[0], r["Grid_Size"], r["VGPR_Count"], name,
                         r["Counter_Value"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        d = per.setdefault(k, {"FETCH_SIZE": [], "WRITE_SIZE": []})
        d[name].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 1 --warmup 1 "
               "--cpu-frames 0 --no-events` (default workload: B=32 KITTI 1241x376 pairs, D=128, 8 paths). hbm_bytes_per_launch = "
               "(2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE counts half the bytes of coalesced reads "
               "(MI355X_MICROARCH.md, HBM). Cross-check on known byte counts: the WTA kernel reads 8 E volumes = 15.08 GB "
               "(+ census); a sweep writes one E volume = 1.885 GB.",
       "config": {"batch": 32, "width": 1241, "height": 376, "disp": 128, "paths": 8, "wsz": 5}, "kernels": {}}
for k, d in per.items():
    n = min(len(d["FETCH_SIZE"]), len(d["WRITE_SIZE"]))
    if not n:
        continue
    f = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); w = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    out["kernels"][k] = {"launches_averaged": n, "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                         "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
# SQ / GRBM passes (optional): per-launch averages for the sweep kernel
sq = {}
for sub in ("sq", "grbm"):
    fs = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))
    if not fs:
        continue
    acc = collections.defaultdict(list)
    dur = []
    for r in csv.DictReader(open(fs[0])):
        if "aggregate3_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if sub == "grbm":
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in acc.items():
        sq[k] = sum(v) / len(v)
    if sub == "grbm" and dur and "GRBM_GUI_ACTIVE" in sq:
        # the counter sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
        sq["clock_GHz"] = round(sq["GRBM_GUI_ACTIVE"] / 8 / (sum(dur) / len(dur)), 3)
        sq["duration_ns_under_pmc"] = sum(dur) / len(dur)
if sq:
    sq["xor_share"] = XOR_SHARE
    out["sq"] = {"aggregate3_kernel": sq}
sys.path.insert(0, root)
import bench
out["kernel_src_sha256"] = bench.kernel_src_sha256()
out["git_head"] = os.environ.get("GIT_HEAD", "unknown")
json.dump(out, open(os.path.join(dst, tag + "_hbm_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, tag + "_kernel_stats.csv")).read())
print(json.dumps(out["kernels"], indent=1))
