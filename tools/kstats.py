"""Print the msda / gemm rows of rocprofv3 kernel_stats.csv files: python tools/kstats.py DIR [substring ...]"""
import csv
import glob
import sys

for d in sys.argv[1:2]:
    pats = sys.argv[2:] or ["msda"]
    for f in sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if any(p in r["Name"] for p in pats):
                name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                print(f"{f.split('/')[-3] if f.count('/') > 2 else f}: {name[:60]:60s} calls {r['Calls']:>5s} "
                      f"avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}")
