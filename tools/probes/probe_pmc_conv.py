"""Three res4 conv shapes, 10 launches each, for a rocprofv3 --pmc pass (tools/probes/prof_pmc_conv.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.sweep_conv import run
for name, args, post in (("conv3", (256, 33, 1024, 1, 0, 1), True), ("conv1", (1024, 33, 256, 1, 0, 1), False), ("conv2", (256, 33, 256, 3, 1, 1), False)):
    us, tf = run(32, *args, (2, 128, 1), post=post, iters=10)
    print(name, us, tf, flush=True)
