"""Concurrency timeline of the graph-replayed steps from a rocprofv3 --kernel-trace CSV of bench.py: how much of a step's wall time
has 0 / 1 / 2+ kernels in flight, which kernels run ALONE (nothing else on the device) and for how long, and the longest gaps.
usage: step_timeline.py <profile dir> [first step] [n steps]   (steps are delimited by the Adam launch; graph steps come first)"""
import collections, csv, glob, re, sys
d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]][1::2]      # two Adam launches per step: the second closes it
sel = rows[adam[first - 1] + 1:adam[first - 1 + nsteps] + 1]
short = lambda r: re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0][:70]
t0 = int(sel[0]['Start_Timestamp'])
ev = []
for i, r in enumerate(sel):
    ev.append((int(r['Start_Timestamp']), 1, i))
    ev.append((int(r['End_Timestamp']), -1, i))
ev.sort()
live, last = set(), t0
hist = collections.Counter()
alone = collections.defaultdict(float)
gaps = []
for t, kind, i in ev:
    dt = t - last
    if dt > 0:
        hist[min(len(live), 3)] += dt
        if len(live) == 1:
            alone[short(sel[next(iter(live))])] += dt
        if len(live) == 0:
            gaps.append(dt)
    last = t
    if kind == 1: live.add(i)
    else: live.discard(i)
wall = (last - t0) / 1e6
print(f"{nsteps} steps from step {first}: wall/step {wall / nsteps:.2f} ms; kernels/step {len(sel) / nsteps:.0f}")
for k in range(4):
    print(f"  {('0 kernels (gap)', '1 kernel', '2 kernels', '3+ kernels')[k]:16s} {hist[k] / 1e6 / nsteps:7.2f} ms/step  {100 * hist[k] / (last - t0):5.1f}%")
print(f"  gaps: {len(gaps) / nsteps:.0f}/step, mean {sum(gaps) / max(len(gaps), 1) / 1e3:.2f} us, max {max(gaps) / 1e3:.1f} us")
streams = collections.Counter((r.get('Queue_Id'), r.get('Stream_Id')) for r in sel)
print("  queues/streams:", dict(streams))
print("time with exactly ONE kernel in flight, by kernel (ms/step):")
for n, t in sorted(alone.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {n:70s} {t / 1e6 / nsteps:7.3f}")
