"""Recompute bench.py's `roofline` numbers from a rocprofv3 --kernel-trace of the SAME command.

    python tools/roofline_leg_stats.py <rocprof_out_dir> [n_eager_steps=2]

bench.py's roofline leg runs, after its timed graph replays, two eager steps with the two-stream overlap off and brackets
every cl4_conv2d_fwd launch with HIP events. In the kernel trace those two steps are what lies between the last graph
replay and the structured-labels leg; Adam launches delimit steps (two `adam_dev_kernel` launches per step: decoder and head groups). This script locates the
eager pair as the first two steps whose kernels were NOT launched from a graph... rocprofv3 does not tag graph launches, so
the steps are identified by position instead: warm-up (W eager/captured) + K replays come first, then exactly the two eager
steps, then the structured-labels leg (2 + K steps). Pass --index to choose the step numbers explicitly.
Prints per-kernel totals of the selected steps and the average launch duration of the bf16x3 forward family (is_x3 below).
"""
import collections
import csv
import glob
import re
import sys


def is_x3(name):
    """The bf16x3 forward kernels behind cl4_conv2d_fwd (precision 1) / cl4_conv2d_fwd_planes: the role-split kernel's two-plane
    instantiations, the unified-wave kernel and the planes-input kernel."""
    if "conv_s_bf16x3" in name:                          # conv_s<KS, SPLIT, MF, BNT, PL>: PL = 3 is a bf16x6 launch
        return re.search(r"conv_s_bf16x3<([^>]*)>", name).group(1).split(",")[4].strip() == "2"
    if "conv_b_bf16x3" in name:                          # conv_b<KS, BMB, NW, PB, GEMM, NSTG>: PB = 3 is bf16x6, GEMM = true the weight-gradient GEMM
        t = [v.strip() for v in re.search(r"conv_b_bf16x3<([^>]*)>", name).group(1).split(",")]
        return t[3] == "2" and (len(t) < 5 or t[4] == "false")
    if "conv_u_bf16x3" in name or "conv_p_bf16x3" in name or "stem_fused" in name or "bottleneck_fused" in name:
        return True
    m = re.search(r"conv_igemm_bf16x3<([^>]*)>", name)
    return bool(m) and m.group(1).split(",")[5].strip() == "2"


def main():
    d = sys.argv[1]
    first = int(sys.argv[2])            # index (0-based) of the first eager roofline step among all steps of the process
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_dev_kernel" in r["Kernel_Name"]][1::2]   # two parameter groups = two launches per step
    print(f"{len(adam)} steps in the trace; selecting steps {first}..{first + n - 1}")
    lo = adam[first - 1] + 1 if first > 0 else 0
    hi = adam[first + n - 1] + 1
    sel = rows[lo:hi]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in sel:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        key = name.split("(")[0]
        agg[key][0] += 1
        agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
    tot = sum(v[1] for v in agg.values())
    print(f"wall/step {wall / n:.2f} ms, kernel time/step {tot / n:.2f} ms")
    x3_n = x3_ms = 0
    for k, (c, t) in agg.items():
        if is_x3(k):
            x3_n += c
            x3_ms += t
    print(f"bf16x3 forward family (conv_igemm_bf16x3 P=2, conv_u / conv_s / conv_b / conv_p _bf16x3, stem_fused, bottleneck_fused): {x3_n / n:.1f} launches/step, {x3_ms / n:.3f} ms/step, "
          f"avg launch {x3_ms / max(x3_n, 1):.5f} ms")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{k[:80]:80s} calls/step {c / n:7.1f}  ms/step {t / n:8.3f}  {100 * t / tot:5.1f}%")


if __name__ == "__main__":
    main()
