"""How much of a short-K 1x1 layer is epilogue: res4 conv3 shape (n=32, 256 -> 1024 @33x33) with K swept down to 32 and
with / without the fused residual (+ReLU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.sweep_conv import run
for cin in (32, 64, 128, 256, 512, 1024):
    us, tf = run(32, cin, 33, 1024, 1, 0, 1, (2, 128, 1))
    us2, tf2 = run(32, cin, 33, 1024, 1, 0, 1, (2, 128, 1), post=True)
    us3, tf3 = run(32, cin, 33, 1024, 1, 0, 1, (2, 64, 1), post=True)
    print(f"K={cin:5d}  t128 {us:7.1f} us {tf:6.1f} TF | +residual {us2:7.1f} us {tf2:6.1f} TF | t64 +residual {us3:7.1f} us {tf3:6.1f} TF", flush=True)
