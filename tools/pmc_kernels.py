"""Per-kernel hardware evidence for the top kernels of the eager roofline leg: MFMA-busy fraction and HBM GB/s.

    python tools/pmc_kernels.py <pmc_dir> <kernel_trace_dir> <first eager step> <out.json>

Inputs: the three separate `rocprofv3 --pmc` passes of tools/prof_pmc.sh over the last eager step of bench.py (bench_FETCH_SIZE,
bench_WRITE_SIZE, bench_MFMA: SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES + GRBM_GUI_ACTIVE) and an un-instrumented --kernel-trace of
the same command for the kernel DURATIONS (a PMC pass serialises and slows the launches, so its own times are not used).
Per kernel name (top N by time):
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   — share of all SIMD-cycles of the launches in which the
              matrix pipe executes (SQ_VALU_MFMA_BUSY_CYCLES counts pipe cycles summed over the SIMDs: 16 per v_mfma_f32_16x16x32_bf16;
              GRBM_GUI_ACTIVE is summed over the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back');
  hbm_gbs   = (FETCH_SIZE x measured factor + WRITE_SIZE x measured factor) / kernel time of the un-instrumented trace.
bench.py reports the bf16x3 forward family's row as roofline.mfma_busy."""
import collections, csv, glob, json, re, sys

pmc, trace, first, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0]


def is_x3(name):
    if "conv_s_bf16x3" in name:                          # conv_s<KS, SPLIT, MF, BNT, PL>: PL = 3 is a bf16x6 launch
        return re.search(r"conv_s_bf16x3<([^>]*)>", name).group(1).split(",")[4].strip() == "2"
    if "conv_b_bf16x3" in name:                          # conv_b<KS, BMB, NW, PB, GEMM, NSTG>: PB = 3 is bf16x6, GEMM = true the weight-gradient GEMM
        t = [v.strip() for v in re.search(r"conv_b_bf16x3<([^>]*)>", name).group(1).split(",")]
        return t[3] == "2" and (len(t) < 5 or t[4] == "false")
    if "conv_u_bf16x3" in name or "conv_p_bf16x3" in name or "stem_fused" in name or "bottleneck_fused" in name:
        return True
    m = re.search(r"conv_igemm_bf16x3<([^>]*)>", name)
    return bool(m) and m.group(1).split(",")[5].strip() == "2"


def last_step(d, counters):
    f = glob.glob(f"{pmc}/{d}/**/*_counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    disp = collections.OrderedDict()
    for x in rows:
        k = int(x["Dispatch_Id"])
        disp.setdefault(k, {"name": x["Kernel_Name"]})[x["Counter_Name"]] = float(x["Counter_Value"])
    seq = [disp[k] for k in sorted(disp)]
    adam = [i for i, x in enumerate(seq) if "adam" in x["name"]]
    sel = seq[adam[-3] + 1: adam[-1] + 1] if len(adam) >= 3 else seq
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for x in sel:
        a = agg[short(x["name"])]
        a["launches"] += 1
        for c in counters:
            a[c] += x.get(c, 0.0)
    return agg


fetch = last_step("bench_FETCH_SIZE", ["FETCH_SIZE"])
write = last_step("bench_WRITE_SIZE", ["WRITE_SIZE"])
mfma = last_step("bench_MFMA", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
cal = json.load(open(f"{pmc}/traffic.json"))["calibration"]
f_fac = cal["vec_gather_16B_per_lane"]["fetch_factor"]
w_fac = cal["vec_gather_16B_per_lane"]["write_factor"]

f = glob.glob(trace + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_dev_kernel" in r["Kernel_Name"]][1::2]
n_steps = 2
sel = rows[adam[first - 1] + 1: adam[first + n_steps - 1] + 1]
dur = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    k = short(r["Kernel_Name"])
    dur[k][0] += 1
    dur[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6


def row(names):
    ms = sum(dur[k][1] for k in names) / n_steps
    launches = sum(dur[k][0] for k in names) / n_steps
    fb = sum(fetch[k]["FETCH_SIZE"] for k in names if k in fetch) * 1024 * f_fac
    wb = sum(write[k]["WRITE_SIZE"] for k in names if k in write) * 1024 * w_fac
    mb = sum(mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"] for k in names if k in mfma)
    gui = sum(mfma[k]["GRBM_GUI_ACTIVE"] for k in names if k in mfma)
    sqb = sum(mfma[k]["SQ_BUSY_CYCLES"] for k in names if k in mfma)
    return {"launches_per_step": launches, "ms_per_step": round(ms, 4), "hbm_read_bytes_per_step": fb, "hbm_write_bytes_per_step": wb,
            "hbm_gbs": round((fb + wb) / (ms * 1e-3) / 1e9, 1) if ms else None,
            "hbm_frac_of_achievable": round((fb + wb) / (ms * 1e-3) / 1e9 / 6300.0, 3) if ms else None,     # 6.3 TB/s: MI355X_MICROARCH.md, HBM [CDNA4]
            "mfma_busy": round(mb / (1024.0 * gui / 8.0), 4) if gui else None,
            "mfma_busy_cycles": mb, "grbm_gui_active": gui, "sq_busy_cycles": sqb,
            "clock_ghz_under_pmc": None}


# r05: EVERY kernel of the step that takes >= 0.15 ms per step (BatchNorm, depthwise, loss, resize, Adam, weight
# gradients next to the convolutions) — SURVEY §8(d): "HBM bounds the individual loss / label / normalisation / optimizer kernels, reported per kernel"
min_ms = float(sys.argv[5]) if len(sys.argv) > 5 else 0.15
top = [k for k in sorted(dur, key=lambda k: -dur[k][1]) if dur[k][1] / n_steps >= min_ms]
out = {"note": "eager, overlap-off step of bench.py; durations from the un-instrumented kernel trace, counters from separate --pmc passes "
               "(FETCH_SIZE x %.3f, WRITE_SIZE x %.3f: factors measured by tools/pmc_calib.py); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
               "(1024 SIMDs x GRBM_GUI_ACTIVE / 8)" % (f_fac, w_fac),
       "kernels": {k: row([k]) for k in top},
       "bf16x3_forward_family": row([k for k in dur if is_x3(k)])}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16      # the kernel sources these counters were measured on: bench.py refuses the figures for any other build
out["csrc_sha16"] = csrc_sha16()
json.dump(out, open(out_path, "w"), indent=1)
print(f"{'kernel':72s} {'ms/step':>8s} {'launches':>8s} {'HBM GB/s':>9s} {'of 6.3TB/s':>10s} {'MFMA busy':>9s}")
for k, v in list(out["kernels"].items()) + [("bf16x3 forward family", out["bf16x3_forward_family"])]:
    print(f"{k[:72]:72s} {v['ms_per_step']:8.3f} {v['launches_per_step']:8.1f} {str(v['hbm_gbs']):>9s} {str(v['hbm_frac_of_achievable']):>10s} {str(v['mfma_busy']):>9s}")
