"""Per-kernel table of some steps of a rocprofv3 --kernel-trace, steps delimited by the launches of an optimizer kernel:
    python tools/step_kernels.py <rocprof_out_dir> <delimiter kernel substring> <its launches per step> <first step> <n steps>
(phase 1: sgd_kernel, 5 per step; phase 2: adam_dev_kernel, 2 per step). Prints wall time per step (first kernel start to last kernel end),
kernel time per step and every kernel's calls / ms per step."""
import collections, csv, glob, re, sys

d, delim, per, first, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if delim in r["Kernel_Name"]][per - 1::per]
print(f"{len(marks)} steps in the trace; selecting steps {first}..{first + n - 1}")
lo = marks[first - 1] + 1 if first > 0 else 0
hi = marks[first + n - 1] + 1
sel = rows[lo:hi]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
    agg[name][0] += 1
    agg[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
wall = (max(int(r["End_Timestamp"]) for r in sel) - int(sel[0]["Start_Timestamp"])) / 1e6 / n
tot = sum(v[1] for v in agg.values()) / n
print(f"wall/step {wall:.2f} ms, kernel time/step {tot:.2f} ms, launches/step {len(sel) / n:.0f}")
for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:100]:100s} calls/step {c / n:7.1f}  ms/step {ms / n:8.3f}  avg us {1e3 * ms / c:8.1f}  {100 * ms / n / tot:5.1f}%")
