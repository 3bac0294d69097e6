"""HBM-side traffic per kernel family of one bench step from two PMC passes (FETCH_SIZE, WRITE_SIZE):
python tools/pmc_traffic.py FETCH_DIR WRITE_DIR STEPS [OUT.json]
(OUT.json: the same rows as data - profiles/pmc_traffic.json is what bench.py's `traffic` fields read.)
FETCH_SIZE is doubled (gfx950 reports half of the bytes of wide coalesced reads: MI355X_MICROARCH.md, calibrated in
profiles/r01_pmc_msda_level_N8.md); both counters are in KiB."""
import collections
import csv
import glob
import json
import sys

FAM = (("gemm_f32_kernel", "gemm_f32_kernel"), ("gemm_f32_kernel", "linear_rows_kernel"),   # (one family, two kernels)
       ("conv_wino_kernel", "conv_wino_kernel"), ("conv_igemm_kernel", "conv_igemm_kernel"), ("conv_igemm_kernel", "conv_tile_kernel"),
       ("msda_fused_level", "msda_fused_level"))


def load(d, ctr):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr:
            continue
        for key, pat in FAM:
            if pat in r["Kernel_Name"]:
                tot[key] += float(r["Counter_Value"])
                cnt[key] += 1
    return tot, cnt


steps = int(sys.argv[3])
ft, fc = load(sys.argv[1], "FETCH_SIZE")
wt, wc = load(sys.argv[2], "WRITE_SIZE")
print("| kernel family | launches / step | 2 x FETCH_SIZE (MB / step) | WRITE_SIZE (MB / step) | HBM traffic (MB / step) | per launch (MB) |")
print("|---|---|---|---|---|---|")
doc = {}
for key in dict.fromkeys(k for k, _ in FAM):
    n = fc[key] / steps
    rd, wr = 2 * ft[key] * 1024 / steps / 1e6, wt[key] * 1024 / steps / 1e6
    print(f"| `{key}` | {n:.1f} | {rd:.1f} | {wr:.1f} | {rd + wr:.1f} | {(rd + wr) / max(n, 1):.2f} |")
    doc[key] = {"launches_per_step": n, "fetch_bytes_per_step": rd * 1e6, "write_bytes_per_step": wr * 1e6}
if len(sys.argv) > 4:
    with open(sys.argv[4], "w") as fh:
        json.dump({"steps_profiled": steps, "frames_per_step": 32, "workload": "bench.py --no-cpu-baseline --pipeline 0 --steps 2 --warmup 1",
                   "note": "fabric side of L2 (HBM + Infinity Cache): 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, separate PMC passes",
                   "families": doc}, fh, indent=1, sort_keys=True)
