"""Short-K layers whose tile count is not a whole number of rounds: plain 128x128, main + small-tile tail (-1), main + split tail (-4)
and plain 64x64 on the res4 conv3 shapes (256 -> 1024, 33x33, residual + ReLU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_conv import run
for n in (16, 32):
    for cin, cout in ((256, 1024), (1024, 256), (64, 256)):
        h = 33 if cin != 64 else 129
        res = []
        for plan in ((2, 128, 1), (2, 128, -1), (2, 128, -4), (2, 64, 1)):
            us, tf = run(n, cin, h, cout, 1, 0, 1, plan, post=True)
            res.append(f"{plan[1]}/{plan[2]:>2d}: {us:7.1f} us {tf:6.1f} TF")
        print(f"n={n} {cin}->{cout}@{h}  " + " | ".join(res), flush=True)
