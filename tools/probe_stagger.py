"""Does offsetting the second resident workgroup of each CU (CL4_CONV_STAGGER) let K loops and epilogues overlap?
res4 shapes at the bench batch: conv3 (256->1024 + residual), conv1 (1024->256), conv2 (3x3 256->256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.sweep_conv import run
for n in (32, 16):
    for name, args, post in (("conv3 256->1024 +res", (256, 33, 1024, 1, 0, 1), True), ("conv1 1024->256", (1024, 33, 256, 1, 0, 1), False),
                             ("conv2 3x3 256->256", (256, 33, 256, 3, 1, 1), False), ("res2 64->256 @129 +res", (64, 129, 256, 1, 0, 1), True),
                             ("res3 128->512 @65 +res", (128, 65, 512, 1, 0, 1), True)):
        row = []
        for st in (0, 1, 2, 3, 4, 6, 8):
            os.environ["CL4_CONV_STAGGER"] = str(st)
            us, tf = run(n, *args, (2, 128, 1), post=post)
            row.append(f"s{st}: {us:6.1f}us")
        print(f"n={n} {name:24s} " + "  ".join(row), flush=True)
