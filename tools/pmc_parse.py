"""Mean counter value per kernel from rocprofv3 --pmc ... --output-format csv runs:
python tools/pmc_parse.py DIR [DIR ...]"""
import collections
import csv
import glob
import os
import sys

for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (name, ctr), vals in sorted(acc.items()):
            if any(k in name for k in ("msda", "bias_act")):
                print(f"{os.path.basename(d.rstrip('/'))}: {name[:40]:40s} {ctr:14s} dispatches {len(vals):3d} mean {sum(vals) / len(vals):14.1f}")
