"""Timeline of the last repetition in a rocprofv3 --kernel-trace CSV: kernel, duration, idle gap before it.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 tools/tail_profile.py temporal
    python tools/trace_timeline.py DIR/run_kernel_trace.csv REPS [top]

The trace holds REPS identical repetitions (tail_profile.py: 6); the last one is printed in time order and summed by kernel."""
import collections
import csv
import glob
import sys

path = sys.argv[1]
if not path.endswith(".csv"):
    path = sorted(glob.glob(path + "/**/*kernel_trace.csv", recursive=True))[0]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))))
n = len(rows) // reps
last = rows[-n:]
short = lambda k: k.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")[:72]  # noqa: E731
busy = sum(e - s for s, e, _ in last)
span = last[-1][1] - last[0][0]
print(f"{n} launches per repetition; span {span / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us")
by = collections.OrderedDict()
prev = None
for s, e, k in last:
    gap = 0 if prev is None else max(0, s - prev)
    prev = max(e, prev or 0)
    b = by.setdefault(short(k), [0, 0, 0])
    b[0] += 1; b[1] += e - s; b[2] += gap
    if len(sys.argv) > 3 and sys.argv[3] == "all":
        print(f"{(s - last[0][0]) / 1e3:9.1f} us  +{gap / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {short(k)}")
print("calls   kernel us   gap-before us   kernel")
for k, (c, d, g) in sorted(by.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{c:5d} {d / 1e3:10.1f} {g / 1e3:10.1f}   {k}")
