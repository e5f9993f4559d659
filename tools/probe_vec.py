"""1x1 layers of the bench step with the 4-pixel vector gather on / off (CL4_CONV_VEC, read once per process: run twice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_conv import run
print("CL4_CONV_VEC =", os.environ.get("CL4_CONV_VEC", "1"))
for n in (32, 16):
    for name, args, post in (("conv3 256->1024 +res", (256, 33, 1024, 1, 0, 1), True), ("conv1 1024->256", (1024, 33, 256, 1, 0, 1), False),
                             ("res5 512->2048 +res", (512, 33, 2048, 1, 0, 1), True), ("res5 2048->512", (2048, 33, 512, 1, 0, 1), False),
                             ("res2 64->256 @129 +res", (64, 129, 256, 1, 0, 1), True), ("res2 256->64 @129", (256, 129, 64, 1, 0, 1), False),
                             ("res3 128->512 @65 +res", (128, 65, 512, 1, 0, 1), True), ("res3 512->128 @65", (512, 65, 128, 1, 0, 1), False)):
        us, tf = run(n, *args, (2, 128, 1), post=post)
        print(f"n={n} {name:24s} {us:7.1f} us {tf:6.1f} TF/s", flush=True)
