"""Unified-wave kernel (conv_u.hip) against the role-split kernel on the bench step's dense shapes: run once with CL4_CONV_U=1 and
once with CL4_CONV_U=0 (the switch is read once per process). Unsplit 128x128 plans; n = 30 / 15 give exactly one / half a round
of resident workgroups on the 33x33 maps, 32 / 16 are the bench batches (1.07 rounds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_conv import run
print("CL4_CONV_U =", os.environ.get("CL4_CONV_U", "1"))
SHAPES = (("conv1 1024->256", (1024, 33, 256, 1, 0, 1), False), ("conv2 3x3 256->256", (256, 33, 256, 3, 1, 1), False),
          ("conv3 256->1024 +res", (256, 33, 1024, 1, 0, 1), True), ("aspp 3x3 d12 2048->256", (2048, 33, 256, 3, 12, 12), False),
          ("res5 3x3 d2 512->512", (512, 33, 512, 3, 2, 2), False), ("res5 512->2048 +res", (512, 33, 2048, 1, 0, 1), True),
          ("res5 2048->512", (2048, 33, 512, 1, 0, 1), False))
for n in (30, 15, 32, 16):
    for name, args, post in SHAPES:
        us, tf = run(n, *args, (2, 128, 1), post=post)
        print(f"n={n} {name:24s} {us:7.1f} us {tf:6.1f} TF/s", flush=True)
for n in (32, 16):
    for name, args, post in (("res2 64->256 @129 +res", (64, 129, 256, 1, 0, 1), True), ("res2 256->64 @129", (256, 129, 64, 1, 0, 1), False),
                             ("res2 3x3 64->64 @129", (64, 129, 64, 3, 1, 1), False), ("res3 128->512 @65 +res", (128, 65, 512, 1, 0, 1), True),
                             ("res3 512->128 @65", (512, 65, 128, 1, 0, 1), False), ("res3 3x3 128->128 @65", (128, 65, 128, 3, 1, 1), False)):
        us, tf = run(n, *args, (2, 128, 1), post=post)
        print(f"n={n} {name:24s} {us:7.1f} us {tf:6.1f} TF/s", flush=True)
