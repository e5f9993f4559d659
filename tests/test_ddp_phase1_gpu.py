"""Phase 1 at N > 1: InPlaceABNSync(group=WORLD) exchanges one packed float64 vector per
layer and direction (ops.STAT_EXCHANGE), and from its third call on the step is a chain of HIP-graph segments cut at those exchanges
(train._SegmentedGraph) with the bucketed gradient average and the SGD kernels at the end. Two processes share the one GPU of the box
and talk over gloo (RCCL refuses two ranks on one device; the code path is the same, only the transport differs); rank r trains on
image r of 2-image batches.

* the sharded emulation: with SYNCHRONISED batch statistics and batch-mean losses, two ranks with one image each compute exactly the
  iteration of one process with both images — the single-process trainer on the full batch is the emulation (the eager first step is
  compared: losses and the SGD update);
* the graph step against the eager step FROM THE SAME STATE at world size 2 (like tests/test_phase1_gpu.py at world size 1:
  trajectories of this ill-conditioned tiny configuration cannot be compared, single steps can);
* the collectives of one iteration are counted and printed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from util import QuietLogger, p1_opts

pytestmark = pytest.mark.gpu
CROP = 65


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batches():
    g = torch.Generator().manual_seed(29)
    return [(torch.randn(2, 3, CROP, CROP, generator=g), torch.eye(20)[[15 + i % 5, 19 - i % 3]].clone()) for i in range(5)]


def _make(dev, batch):
    from cl4wsis_amd.train import Trainer
    from oracle import weights
    opts = p1_opts(CROP, extra=["--batch_size", str(batch)])
    opts.max_iters, opts.start_decay = 1000, 1000
    torch.manual_seed(0)
    tr = Trainer(QuietLogger(), dev, opts)
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=41 + i))
    tr._realias()
    return tr


def _flat(tr):
    return torch.cat([p.detach().flatten() for m in (tr.model, tr.pseudolabeler, tr.peakgenerator) for p in m.parameters()]).clone()


def _rank(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      CL4_DIST_BACKEND="gloo", CL4_AUTOTUNE="0")
    torch.set_num_threads(8)
    import torch.distributed as dist
    from cl4wsis_amd import ops
    from cl4wsis_amd.parallel import init_distributed
    init_distributed()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    tr = _make(dev, 1)
    assert tr.buckets.distributed and tr.buckets.world == world
    mods = (tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)
    shard = lambda b: (b[0][rank:rank + 1], b[1][rank:rank + 1])
    batches = _batches()
    np.random.seed(11)                                  # the FLAC angle is drawn on the host (train.py:316): the same on every rank
    w_init = _flat(tr)
    n0 = ops.STAT_EXCHANGE.count
    out0 = tr.step(shard(batches[0]))                   # eager (iteration 0)
    eager_exchanges = ops.STAT_EXCHANGE.count - n0
    torch.cuda.synchronize()
    l0 = {k: float(v) for k, v in out0["losses"].items()}
    upd0 = (_flat(tr) - w_init).cpu()
    for b in batches[1:4]:                              # iteration 1 eager, iterations 2 and 3 replay the segmented graph
        out = tr.step(shard(b))
        assert all(np.isfinite(float(v)) for v in out["losses"].values())
    assert tr._p1_static is not None and 1 <= len(tr._p1_static["graphs"]) <= 2
    coll = tr.collectives_per_iteration()
    assert coll is not None and coll["abn_statistics"] == eager_exchanges and coll["graph_segments"] == coll["abn_statistics"] + 2
    torch.cuda.synchronize()
    # one step from one state: segmented-graph replay, then (state restored) the eager step with the same FLAC angle
    snap = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in mods]
    opt_snap = [v.clone() for v in tr.optimizer.buf_fb.views]
    seed = tr._p1_seed_dev.clone()

    def restore():
        for m, sd in zip(mods, snap):
            m.load_state_dict(sd)
        tr._realias()
        for v, s in zip(tr.optimizer.buf_fb.views, opt_snap):
            v.copy_(s)
        tr._p1_seed_dev.copy_(seed)

    x, l = shard(batches[4])
    x, l = x.to(dev), l.to(dev).float()
    angle = next(iter(tr._p1_static["graphs"]))[0]      # an angle whose graph exists
    w_before = _flat(tr)
    lg = {k: float(v) for k, v in tr._replay_phase1(x, l, angle, False).items()}
    torch.cuda.synchronize()
    wg = _flat(tr)
    restore()
    le = {k: float(v) for k, v in tr._p1.run(x, l, 0, flac_angle=angle).items()}
    for i in range(len(tr.buckets.fb.sizes)):
        tr.buckets.ready(i)
    tr.buckets.finish()
    tr.optimizer.step()
    torch.cuda.synchronize()
    we = _flat(tr)
    rel = ((we - wg).norm() / (we - w_before).norm()).item()
    ret[rank] = {"l0": l0, "upd0": upd0, "coll": coll, "lg": lg, "le": le, "rel": rel, "w_end": we.cpu()}
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_phase1_segmented_graph_step(gpu, monkeypatch):
    monkeypatch.setenv("CL4_AUTOTUNE", "0")
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    ranks = mp.spawn(_rank, args=(2, _free_port(), ret), nprocs=2, join=False)
    # ---- the emulation: one process, the full 2-image batch, one eager step from the same weights --------------------------------
    tr = _make(gpu, 2)
    assert not tr.buckets.distributed
    np.random.seed(11)
    w_init = _flat(tr)
    out = tr.step(_batches()[0])
    torch.cuda.synchronize()
    emu_l = {k: float(v) for k, v in out["losses"].items()}
    emu_upd = (_flat(tr) - w_init).cpu()
    import time
    deadline = time.time() + 600
    while not ranks.join(timeout=5):
        if time.time() > deadline:
            for p in ranks.processes:
                if p.is_alive():
                    p.terminate()
            pytest.fail("the two ranks did not finish within 600 s: a collective of the phase-1 N > 1 path is unpaired")
    r0, r1 = ret[0], ret[1]
    assert torch.equal(r0["w_end"], r1["w_end"]), "the two ranks diverged"
    print("P1DDP collectives per iteration and rank:", r0["coll"])
    assert r0["coll"] == r1["coll"] and r0["coll"]["abn_statistics"] > 150 and r0["coll"]["gradient_buckets"] >= 1
    # (a) sharded emulation, eager first step: every loss term is a batch mean over synchronised statistics -> mean over ranks == full batch
    for k, v in emu_l.items():
        two = 0.5 * (r0["l0"][k] + r1["l0"][k])
        assert abs(two - v) <= 2e-3 * max(abs(v), 1e-3), (k, two, v)
    assert torch.equal(r0["upd0"], r1["upd0"])
    rel0 = ((r0["upd0"] - emu_upd).norm() / emu_upd.norm()).item()
    print(f"P1DDP first step: |update(two ranks) - update(one process, both images)| / |update| = {rel0:.3e}")
    assert rel0 < 5e-2, rel0
    # (b) segmented-graph step == eager step from the same state, on both ranks
    for r in (r0, r1):
        assert set(r["lg"]) == set(r["le"])
        for k in r["le"]:
            assert abs(r["le"][k] - r["lg"][k]) <= 1e-4 * max(abs(r["le"][k]), 1e-3), (k, r["le"][k], r["lg"][k])
        print("P1DDP segmented graph vs eager: update difference relative to the update", r["rel"])
        assert r["rel"] < 2e-2, r["rel"]
