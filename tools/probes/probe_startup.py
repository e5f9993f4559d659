"""Where does a child process's start-up time go on a fresh GPU box? (tests that spawn ranks / loader workers take 40-200 s there)
usage: python tools/probes/probe_startup.py"""
import subprocess
import sys
import time

CONV = ("import cl4wsis_amd.ops as o; x = torch.randn(2, 64, 33, 33, device='cuda'); w = torch.randn(64, 64, 3, 3, device='cuda'); "
        "o.conv2d(x, o.pack_conv_weight(w, 1), 3, pad=1); torch.cuda.synchronize()")
PG = ("import torch.distributed as d, os; os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='%d'); d.init_process_group('%s', rank=0, "
      "world_size=1); x = torch.ones(4, device='%s'); d.all_reduce(x); d.destroy_process_group()")
CASES = [
    ("python -c pass", "pass"),
    ("import torch", "import torch"),
    ("import torch (again)", "import torch"),
    ("+ first device tensor", "import torch; torch.zeros(1, device='cuda'); torch.cuda.synchronize()"),
    ("+ first device tensor (again)", "import torch; torch.zeros(1, device='cuda'); torch.cuda.synchronize()"),
    ("+ libcl4wsis_hip.so loaded", "import torch, cl4wsis_amd._hip as h; h.load()"),
    ("+ one convolution", "import torch, cl4wsis_amd._hip as h; h.load(); " + CONV),
    ("+ one convolution (again)", "import torch, cl4wsis_amd._hip as h; h.load(); " + CONV),
    ("gloo group + all_reduce", "import torch; " + PG % (29533, "gloo", "cpu")),
    ("nccl group + all_reduce", "import torch; " + PG % (29534, "nccl", "cuda")),
    ("nccl group + all_reduce (again)", "import torch; " + PG % (29535, "nccl", "cuda")),
]
for name, code in CASES:
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print(f"{time.time() - t0:8.2f} s  rc={r.returncode}  {name}", flush=True)
    if r.returncode:
        print(r.stderr[-500:])
