"""Numbers of a measurement pass (tools/run_final.sh <tag>) in one place, for DESIGN.md section 6 / README: python tools/summarize_final.py <tag>"""
import json, os, sys
tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = lambda *p: os.path.join(R, "gpurun_out", *p)
d = json.load(open(g(tag, "bench.json")))
print("bench default:", round(d["value"], 1), "images/s", round(d["ms_per_step"], 2), "ms  vs_stock_rocm", round(d.get("vs_stock_rocm") or 0, 1))
r = d["roofline"]
for k in ("achieved", "frac", "kernel_ms_per_step", "avg_launch_ms", "launches_per_step", "traffic", "algorithmic_bytes_per_launch",
          "algorithmic_bytes_per_launch_incl_epilogue_operands", "traffic_over_algorithmic", "mfma_busy", "hbm_gbs"):
    print("  roofline.%s = %s" % (k, r.get(k)))
print("  cpu_baseline", d["cpu_baseline"]["value"], "stock_rocm", d["stock_rocm"]["value"])
s = d.get("structured_labels", {})
print("  structured", s.get("value"), s.get("ms_per_step"))
for k in ("conv_wgrad", "conv_fwd_bf16x6"):
    print(" ", k, {q: round(v, 3) if isinstance(v, float) else v for q, v in d[k].items() if q in ("achieved", "frac", "ms_per_step", "launches_per_step")})
print("  whole_step_frac_of_peak", d.get("whole_step_frac_of_peak"), "algorithmic_tflops_whole_step", d.get("algorithmic_tflops_whole_step"))
p = json.load(open(g(tag, "prof_bench.json")))
print("profiled run:", round(p["value"], 1), round(p["ms_per_step"], 2), "family", round(p["roofline"]["achieved"], 1), round(p["roofline"]["kernel_ms_per_step"], 2))
for f in ("roofline_leg.txt", "step_breakdown_graph.txt", "timeline.txt"):
    print("--", f)
    print("".join(open(g(tag, f)).readlines()[:8]), end="")
t = json.load(open(g(tag + "_pmc", "traffic.json")))
print("pmc traffic:", t["conv_igemm_bf16x3"])
k = json.load(open(g(tag, "pmc_kernels.json")))
print("pmc family:", {q: k["bf16x3_forward_family"][q] for q in ("launches_per_step", "ms_per_step", "hbm_gbs", "mfma_busy")})
for n, v in k["kernels"].items():
    print("  %-62s %5.0f %7.3f ms %7.0f GB/s  mfma busy %.2f" % (n[:62], v["launches_per_step"], v["ms_per_step"], v["hbm_gbs"], v["mfma_busy"]))
