"""Per-chain Gantt of the pipelined phase-2 step from time stamps INSIDE the captured graphs: `cl4_stamp` kernels
(10 ns device clock) sit at the head and tail of every chain — the trainable stage on the main stream, the next batch's old-model
forward and (deep mode) its frozen new-model stage on the side streams — so the table shows what really ran side by side, which a
profiler that serialises streams cannot. Also prints the un-stamped step time of the same configuration.

usage: [CL4_FORCE_COLLECTIVES=1] python tools/chain_gantt.py [per-GPU batch ...]      (default 2 4 8)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from cl4wsis_amd.parallel import init_distributed
from cl4wsis_amd.synthetic import synthetic_batch
from cl4wsis_amd.train import Trainer


class _Log:
    def info(self, *a, **k): pass
    debug = info


def run(b, device, stamps, steps=20):
    opts = bench.p2_opts(bench.GLOBAL_BATCH)
    opts.batch_size = b
    opts.max_iters, opts.start_decay = 30 * 661, 5 * 661
    torch.manual_seed(42)
    tr = Trainer(_Log(), device, opts)
    if stamps:
        tr.enable_chain_stamps()
    images, l1h = synthetic_batch(b, 42, device, bench.CROP)
    batch = (images, l1h)
    for _ in range(6):
        tr.step(batch, next_batch=batch)
    torch.cuda.synchronize()
    rows = []
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(batch, next_batch=batch)
        if stamps:
            rows.append(tr.chain_stamps())            # (synchronises: the stamped run is not the timed one)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    deep = bool(tr._pipe and tr._pipe.get("deep"))
    del tr
    from cl4wsis_amd.nn import CACHE
    CACHE.clear()
    torch.cuda.empty_cache()
    return ms, rows, deep


def main():
    rank, world, local = init_distributed()
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    path = "N > 1 code path at world size 1 (CL4_FORCE_COLLECTIVES=1: graph up to the ASPP backward, eager tail + bucketed all-reduce)" \
        if os.environ.get("CL4_FORCE_COLLECTIVES") == "1" else "one-process path"
    print(f"# chain Gantt from in-graph time stamps — {path}")
    for b in [int(v) for v in sys.argv[1:]] or [2, 4, 8]:
        ms_plain, _, deep = run(b, device, stamps=False)
        _, rows, _ = run(b, device, stamps=True)
        print(f"\n== {b} images per GPU ({'deep' if deep else 'shallow'} pipelining): {ms_plain:.2f} ms per step without stamps")
        for q in ("g0.", "g1."):
            mine = [r for r in rows if (q + "main.head") in r]
            # a replay of parity q rewrote q's slots; the other parity's slots hold the previous replay — keep the rows where q is the newest
            mine = [r for r in mine if r[q + "main.head"] == max(r.get("g0.main.head", 0), r.get("g1.main.head", 0))]
            if not mine:
                continue
            names = [n for n in mine[0] if n.startswith(q)]
            avg = {n: sum(r[n] - r[q + "main.head"] for r in mine) / len(mine) for n in names}
            print(f"  graph of parity {q[1]} ({len(mine)} replays): microseconds after main.head")
            for n in sorted(names, key=lambda n: avg[n]):
                print(f"    {n[3:]:28s} {avg[n]:9.1f}")
            end = avg.get(q + "main.end", avg.get(q + "main.joined"))
            chains = []
            for nm, h, t in (("trainable stage (main)", "main.head", "main.trainable_tail"), ("old model (side 1)", "old_model.head", "old_model.tail"),
                             ("frozen new model (side 2)", "frozen_new.head", "frozen_new.tail")):
                if q + h in avg and q + t in avg:
                    chains.append((nm, avg[q + h], avg[q + t]))
            tot = sum(t - h for _, h, t in chains)
            for nm, h, t in chains:
                print(f"    chain {nm:28s} [{h:8.1f} .. {t:8.1f}] us = {t - h:8.1f} us")
            if end:
                print(f"    graph wall {end:8.1f} us; chains sum to {tot:8.1f} us -> {tot / end:.2f} chains in flight on average")
    if world > 1 or os.environ.get("CL4_FORCE_COLLECTIVES") == "1":
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
