"""Print (average us, calls, name) for the kernels of a rocprofv3 *_kernel_stats.csv whose name matches any argument."""
import csv, glob, re, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True) if not path.endswith(".csv") else [path]
pats = sys.argv[2:]
for f in files:
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])
        if not pats or any(p in name for p in pats):
            print(f"{float(r['AverageNs']) / 1e3:9.2f} us x {int(r['Calls']):5d}  {re.sub(r'[(].*', '', name)[:60]}")
