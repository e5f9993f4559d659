#!/bin/bash
# Experiments: HBM-side traffic and L2 hit rate of res4's planes convolutions (tools/probes/probe_res4_pmc.py), one counter set per pass.
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
O=$R/gpurun_out/${1:-r04}_pmc_res4
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for n in 32 16; do
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=n${n}_$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/$tag -- python3 $R/tools/probes/probe_res4_pmc.py $n > $O/$tag.log 2>&1
  echo "pass n=$n $c done"
done
tail -2 $O/n${n}_FETCH_SIZE.log
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/*/")):
    fs = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    # the last 13 launches of every (kernel, grid) pair are the probe's own (3 warm + 10 timed); earlier ones belong to the tuner
    by = collections.defaultdict(list)
    for r in rows:
        by[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:58], r.get("Grid_Size", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    print(d.rstrip("/").split("/")[-1])
    for (k, grid, c), v in sorted(by.items()):
        if "conv_b" in k or "conv_s" in k or "conv_p" in k:
            tail = v[-10:]
            print(f"  {k:60s} grid {grid:>8s} {c:14s} launches {len(v):4d}  last-10 mean {sum(tail) / len(tail) / 1e6:10.2f} M")
PY
