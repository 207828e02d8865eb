import csv, glob, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for name in sys.argv[1:]:
    f = max(glob.glob('%s/gpurun_out/%s/*/*kernel_trace.csv' % (root, name)), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    rows = [r for r in rows if 'sgmx' in r['Kernel_Name']]
    print(name, "last 11 sgmx launches (us):")
    for r in rows[-11:]:
        print("  %-40s %8.1f" % (r['Kernel_Name'].split('(')[0][-40:], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
