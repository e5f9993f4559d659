"""Is a conv launch power-bound? Runs one shape back to back for ~3 s and samples board power and shader clock from sysfs /
rocm-smi meanwhile. usage: probe_power.py [name ...]   (CL4_CONV_U / CL4_LIB select the kernel as usual)"""
import glob, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from cl4wsis_amd import _hip, ops

dev = torch.device("cuda:0")
SHAPES = {"conv2": (30, 256, 33, 256, 3, 1, 1, False), "conv1": (30, 1024, 33, 256, 1, 0, 1, False),
          "conv3": (30, 256, 33, 1024, 1, 0, 1, True), "aspp": (30, 2048, 33, 256, 3, 12, 12, False)}


def sysfs_sample():
    out = {}
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        try:
            out["power_W"] = int(open(f).read()) / 1e6
        except Exception:
            pass
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(f):
                if "*" in line:
                    out["sclk"] = line.strip()
        except Exception:
            pass
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"):
        try:
            out["freq1_MHz"] = int(open(f).read()) / 1e6
        except Exception:
            pass
    return out


def smi_sample():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=10)
        return r.stdout.strip().replace("\n", " | ")[:600]
    except Exception as e:
        return f"rocm-smi failed: {e}"


def run(name):
    n, cin, h, cout, k, pad, dil, post = SHAPES[name]
    x = torch.randn(n, cin, h, h, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.02
    wp = ops.pack_conv_weight(w, 1)
    res = torch.randn(n, cout, h, h, device=dev) if post else None
    y = torch.empty(n, cout, h, h, device=dev)
    d = ops._conv_desc(x, cin, 0, cout, k, 1, pad, dil, cout, 0, 0, 0.0, 0, 0.0, wp.precision)
    _hip.call("cl4_conv2d_fwd_set_plan", C.byref(d), 2, 128, 1)
    for _ in range(10):
        ops.conv2d(x, wp, k, pad=pad, dil=dil, post=res, out=y)
    torch.cuda.synchronize()
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(sysfs_sample())
            time.sleep(0.25)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 0
    e0.record()
    while time.time() - t0 < 3.0:
        for _ in range(200):
            ops.conv2d(x, wp, k, pad=pad, dil=dil, post=res, out=y)
        iters += 200
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    smi = smi_sample() if name == "conv2" else ""
    stop.set(); th.join()
    us = e0.elapsed_time(e1) * 1e3 / iters
    tf = 2.0 * y.numel() * cin * k * k / us / 1e6
    pw = [s.get("power_W") for s in samples if "power_W" in s]
    fr = [s.get("freq1_MHz") for s in samples if "freq1_MHz" in s]
    print(f"{name}: {us:.1f} us/launch {tf:.0f} TF/s  power W min/avg/max {min(pw, default=0):.0f}/{sum(pw) / max(len(pw), 1):.0f}/{max(pw, default=0):.0f}"
          f"  freq1 MHz avg {sum(fr) / max(len(fr), 1):.0f}  sclk {samples[-1].get('sclk') if samples else None}  ({len(samples)} samples)", flush=True)
    if smi:
        print("   rocm-smi:", smi)


print("CL4_CONV_U =", os.environ.get("CL4_CONV_U", "1"), " CL4_LIB =", os.environ.get("CL4_LIB", "default"))
print("idle:", sysfs_sample())
for nm in (sys.argv[1:] or ["conv2", "conv1", "aspp"]):
    run(nm)
for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap*"):
    try:
        print(f.split("/")[-1], open(f).read().strip())
    except Exception as e:
        print(f, e)
