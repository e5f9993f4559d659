"""Aggregate the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_pmc.sh into HBM-side traffic per launch of the
dominant kernel family (the bf16x3 forward kernels: conv_igemm_bf16x3 with two planes, conv_u_bf16x3, conv_s_bf16x3 — the bf16x6
instantiations are a different entry of bench.py's record). usage: pmc_traffic.py <pmc_dir> <out.json>

Units / corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; WRITE_SIZE is exact for streaming stores; FETCH_SIZE
under-reports wide coalesced reads by 2x on gfx950 and is uncalibrated for other widths — so the factor is MEASURED here on
calibration launches of this very kernel with known minimal traffic (tools/pmc_calib.py; both loader variants) and that factor,
not an assumed one, scales the bench launches. Only the dispatches of the LAST step (between the last two Adam launch pairs) count."""
import collections, csv, glob, json, re, sys

root, out_path = sys.argv[1], sys.argv[2]


def rows(d, counter):
    f = glob.glob(f"{root}/{d}/**/*_counter_collection.csv", recursive=True)[0]
    r = [x for x in csv.DictReader(open(f)) if x["Counter_Name"] == counter]
    r.sort(key=lambda x: int(x["Dispatch_Id"]))
    return r


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


def calib(tag):
    """FETCH_SIZE / WRITE_SIZE (KiB) of the last calibration conv launch vs its known bytes."""
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        r = [x for x in rows(f"{tag}_{c}", c) if is_x3(x["Kernel_Name"])]
        res[c] = float(r[-1]["Counter_Value"]) * 1024
    known_in = 80 * 1024 * 33 * 33 * 4 + 1024 * 128 * 4
    known_out = 80 * 128 * 33 * 33 * 4
    return {"fetch_counter_bytes": res["FETCH_SIZE"], "known_read_bytes": known_in, "fetch_factor": known_in / res["FETCH_SIZE"],
            "write_counter_bytes": res["WRITE_SIZE"], "known_write_bytes": known_out, "write_factor": known_out / res["WRITE_SIZE"]}


def last_step(counter):
    r = rows(f"bench_{counter}", counter)
    adam = [i for i, x in enumerate(r) if "adam" in x["Kernel_Name"]]
    sel = r[adam[-3] + 1: adam[-1] + 1] if len(adam) >= 3 else r
    agg = collections.defaultdict(lambda: [0, 0.0])
    for x in sel:
        n = re.sub(r"\(anonymous namespace\)::", "", x["Kernel_Name"]).split("(")[0]
        agg[n][0] += 1
        agg[n][1] += float(x["Counter_Value"]) * 1024
    return agg


cal = {"vec_gather_16B_per_lane": calib("calib1"), "dword_gather_4B_per_lane": calib("calib0")}
fetch, write = last_step("FETCH_SIZE"), last_step("WRITE_SIZE")
f_fac = cal["vec_gather_16B_per_lane"]["fetch_factor"]      # 1x1 launches use it; the 3x3 / strided ones the dword gather
f_fac0 = cal["dword_gather_4B_per_lane"]["fetch_factor"]
out = {"note": "last eager step of bench.py (CL4_GRAPH=0 CL4_OVERLAP=0); bytes; FETCH_SIZE scaled by the factor measured on the "
               "calibration launch of the same loader variant (1x1 unsplit 128-tile launches: vector gather; all others: dword gather)",
       "calibration": cal, "kernels": {}}
tot_l, tot_b = 0, 0.0
for k in sorted(fetch, key=lambda k: -fetch[k][1]):
    l, f = fetch[k]
    w = write.get(k, [0, 0.0])[1]
    out["kernels"][k] = {"launches": l, "fetch_counter_bytes": f, "write_counter_bytes": w}
    if is_x3(k):
        m = re.search(r"<([^>]*)>", k)
        args = m.group(1).split(",") if m else []
        vec = "conv_u_bf16x3" in k or "conv_s_bf16x3" in k or "conv_b_bf16x3" in k or "conv_p_bf16x3" in k or "stem_fused" in k or "bottleneck_fused" in k or (len(args) > 9 and args[9].strip() == "true")   # 16-byte-per-lane activation loads
        fac = f_fac if vec else f_fac0
        out["kernels"][k]["fetch_factor_used"] = fac
        tot_l += l
        tot_b += fac * f + cal["vec_gather_16B_per_lane"]["write_factor"] * w
out["conv_igemm_bf16x3"] = {"launches_per_step": tot_l, "hbm_bytes_per_step": tot_b, "hbm_bytes_per_launch": tot_b / max(tot_l, 1)}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16      # the kernel sources these counters were measured on: bench.py refuses the figures for any other build
out["csrc_sha16"] = csrc_sha16()
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({"calibration": {k: {kk: round(vv, 3) if isinstance(vv, float) and vv < 100 else vv for kk, vv in v.items()} for k, v in cal.items()},
                  "conv_igemm_bf16x3": out["conv_igemm_bf16x3"]}))
