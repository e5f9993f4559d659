"""Kernel-experiment probe: one 128x128 tile per CU (n=15) and two per CU (n=30), 1x1 1024->256, and a 3x3 layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.sweep_conv import run
tag = (os.environ.get("CL4_LIB") or "product").split("/")[-1]
a = run(15, 1024, 33, 256, 1, 0, 1, (2, 128, 1))
b = run(30, 1024, 33, 256, 1, 0, 1, (2, 128, 1))
c = run(32, 256, 33, 256, 3, 1, 1, (2, 128, 1))
d = run(30, 4096, 33, 256, 1, 0, 1, (2, 128, 1))
print(f"{tag:16s} solo {a[0]:7.1f} us  pair {b[0]:7.1f} us  3x3 {c[0]:7.1f} us  K4096x512tiles {d[0]:7.1f} us", flush=True)
