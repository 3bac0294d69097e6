"""Markdown table from the three PMC passes of tools/pmc_kernels.py:  python tools/pmc_report.py DIR1 DIR2 DIR3
(pass 1: SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_MFMA; pass 2: SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU; pass 3:
GRBM_GUI_ACTIVE).  The last of each group of >= 3 identical consecutive kernels is reported."""
import collections
import csv
import glob
import itertools
import sys


def load(d):
    rows = collections.OrderedDict()
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        key = int(r["Dispatch_Id"])
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        rows.setdefault(key, {"name": name, "id": key, "grid": r.get("Grid_Size", "")})[r["Counter_Name"]] = float(r["Counter_Value"])
    t = {}
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        t[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    vals = [v for v in rows.values() if any(s in v["name"] for s in ("gemm_f32", "Cijk", "conv_wino", "conv_igemm", "conv_tile"))]
    groups = []
    for _, grp in itertools.groupby(vals, key=lambda v: (v["name"], v["grid"])):
        grp = list(grp)
        if len(grp) >= 3:
            g = grp[-1]
            g["us"] = t.get(g["id"], 0.0)
            groups.append(g)
    # the Winograd convolution alternates its main and tail launches: last dispatch of every distinct (kernel, grid)
    last = collections.OrderedDict()
    for v in vals:
        if "conv_wino" in v["name"]:
            last[(v["name"], v["grid"])] = v
    for g in last.values():
        g["us"] = t.get(g["id"], 0.0)
        groups.append(g)
    return groups


r1, r2, r3 = (load(d) for d in sys.argv[1:4])
print("| kernel | us (under PMC) | MFMA busy / CU busy | wave cycles waiting (s_waitcnt, barrier) | VALU / MFMA instr | LDS / MFMA instr | "
      "SALU / MFMA instr | LDS conflict cycles / LDS active | clock GHz |")
print("|---|---|---|---|---|---|---|---|---|")
for v, v2, v3 in zip(r1, r2, r3):
    assert v["name"] == v2["name"] == v3["name"], (v["name"], v2["name"], v3["name"])
    mf = max(v["SQ_INSTS_MFMA"], 1.0)
    clk = v3.get("GRBM_GUI_ACTIVE", 0) / 8 / (v3["us"] * 1e-6) / 1e9
    print(f"| `{v['name'][:56]}` | {v3['us']:.0f} | {v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(v['SQ_BUSY_CU_CYCLES'], 1) / 4:.3f} | "
          f"{v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.2f} | {v['SQ_INSTS_VALU'] / mf:.2f} | {v2['SQ_INSTS_LDS'] / mf:.2f} | "
          f"{v2['SQ_INSTS_SALU'] / mf:.2f} | {v2['SQ_LDS_BANK_CONFLICT'] / max(v2['SQ_LDS_IDX_ACTIVE'], 1):.2f} | {clk:.2f} |")
