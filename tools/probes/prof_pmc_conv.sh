set -e
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is the root of the repository copy)}
mkdir -p $R/gpurun_out/pmc_conv
export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_conv/$tag -- python3 $R/tools/probes/probe_pmc_conv.py > $R/gpurun_out/pmc_conv/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(R + "/gpurun_out/pmc_conv/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_igemm_bf16x3" not in k: continue
        key = k.split("<")[1].split(">")[0][:40] + " grid" + r.get("Grid_Size", "")
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(key, r["Counter_Name"])] += 1
for key, d in agg.items():
    print(key)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v / cnt[(key, c)]:.4g} per launch")
PY
