"""Per-step steady-state kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py (last N steps, delimited by the
Adam launches: 2 per step)."""
import collections, csv, glob, re, sys
d = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name']]
start, end = adam[-2 * nsteps - 1] + 1, adam[-1] + 1
sel = rows[start:end]
wall = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0]
    agg[n][0] += 1
    agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
tot = sum(v[1] for v in agg.values())
print(f"wall/step {wall / nsteps:.2f} ms, kernel time/step {tot / nsteps:.2f} ms")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{n[:72]:72s} calls/step {c / nsteps:7.1f}  ms/step {t / nsteps:8.3f}  {100 * t / tot:5.1f}%")
