#!/bin/bash
# Experiments: HBM-side traffic and L2 hit rate of the fused bottleneck kernel (csrc/conv_f.hip) on tools/probes/probe_f.py, one counter set per pass.
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
O=$R/gpurun_out/${1:-r04}_pmc_f
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/$tag -- python3 $R/tools/probes/probe_f.py ${PROBE_N:-32} > $O/$tag.log 2>&1
  echo "pass $c done"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/*/")):
    fs = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = (r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    for (k, c), (n, v) in sorted(agg.items()):
        if "fused" in k or "conv_s" in k or "conv_b" in k:
            print(f"{k:62s} {c:20s} launches {n:4d} per-launch {v / n:14.1f}")
PY
