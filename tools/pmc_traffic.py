"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into HBM-side traffic per launch of the dominant
kernel (conv_igemm_bf16x3), corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for gfx950 (FETCH_SIZE
counts 64 B per 128-B request: doubled; WRITE_SIZE exact; both in KiB). usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Only the dispatches of the LAST step (after the last-but-one Adam launch pair) are counted."""
import collections, csv, glob, json, re, sys


def last_step(d, counter):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    sel = rows[adam[-3] + 1: adam[-1] + 1] if len(adam) >= 3 else rows
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in sel:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
        agg[n][0] += 1
        agg[n][1] += float(r["Counter_Value"])
    return agg


fetch, write = last_step(sys.argv[1], "FETCH_SIZE"), last_step(sys.argv[2], "WRITE_SIZE")
out = {"note": "last eager step of bench.py (CL4_GRAPH=0 CL4_OVERLAP=0); KiB; FETCH_SIZE doubled (gfx950 correction)", "kernels": {}}
tot_l, tot_b = 0, 0.0
for k in sorted(fetch, key=lambda k: -fetch[k][1]):
    l, f = fetch[k]
    w = write.get(k, [0, 0.0])[1]
    out["kernels"][k] = {"launches": l, "fetch_kib_x2": 2 * f, "write_kib": w}
    if "conv_igemm_bf16x3" in k:
        tot_l += l
        tot_b += (2 * f + w) * 1024
out["conv_igemm_bf16x3"] = {"launches_per_step": tot_l, "hbm_bytes_per_step": tot_b, "hbm_bytes_per_launch": tot_b / max(tot_l, 1)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["conv_igemm_bf16x3"]))
