"""Per-dispatch counter table from a rocprofv3 --pmc csv run, in dispatch order:
python tools/pmc_table.py DIR"""
import collections
import csv
import glob
import sys

for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        key = int(r["Dispatch_Id"])
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:46]
        rows.setdefault(key, {"name": name})[r["Counter_Name"]] = float(r["Counter_Value"])
    ctrs = sorted({c for v in rows.values() for c in v if c != "name"})
    print(f"{'kernel':46s} " + " ".join(f"{c[-18:]:>18s}" for c in ctrs))
    for k, v in rows.items():
        if any(s in v["name"] for s in ("gemm", "Cijk", "igemm", "Conv", "conv")):
            print(f"{v['name']:46s} " + " ".join(f"{v.get(c, 0):18.0f}" for c in ctrs))
