"""Where does the stock-PyTorch-ROCm comparator (oracle.p2_step on cuda) spend its time? (experiment; bench.py's stock_rocm leg)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cl4wsis_amd.synthetic import synthetic_batch
from oracle import model as M, step as ostep

dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
opts = bench.p2_opts(16)
new, old, pl, pg = bench._oracle_state(opts, dev)
images, l1h = synthetic_batch(b, 42, dev, 513)


def T(name, fn, n=2):
    for i in range(n):
        torch.cuda.synchronize(); t0 = time.time(); r = fn(); torch.cuda.synchronize()
        print(f"{name} run {i}: {time.time() - t0:.2f} s", flush=True)
    return r


with torch.no_grad():
    T("old forward_full", lambda: M.forward_full(images, old, pooling=opts.pooling, training=False))
    out2, aux = T("forward_seg 2b", lambda: M.forward_seg(torch.cat([images, images.flip(-1)], 0), new, pooling=opts.pooling, training=False))
    with torch.autocast("cuda", dtype=torch.float16):
        T("forward_seg 2b autocast", lambda: M.forward_seg(torch.cat([images, images.flip(-1)], 0), new, pooling=opts.pooling, training=False))
T("whole p2_step", lambda: ostep.p2_step(new, old, pl, pg, images, l1h, 16, pooling=opts.pooling, lr=5e-4), n=2)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
ostep.p2_step(new, old, pl, pg, images, l1h, 16, pooling=opts.pooling, lr=5e-4); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
