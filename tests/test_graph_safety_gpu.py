"""Lifetime of what a captured step addresses by raw pointer.

* scratch buffers (split-K slabs, BN / loss / label workspaces) are retired on growth, never freed (ops._Scratch);
* the single-stream captured step (CL4_OVERLAP=0 + graph) — the configuration that faulted in r02 because hipMemsetAsync nodes were
  replayed through the runtime's captured-packet path — runs and reproduces the two-stream results;
* a captured step keeps reproducing the undisturbed run after every scratch buffer has been regrown under it, a short batch has gone
  through the eager path (another label-workspace size), an eval pass has cleared and rebuilt the derived-tensor cache, and the
  allocator has handed the freed sizes to someone else.

These comparisons run with a learning rate of ZERO: the weights then never move, every step's losses are a function of its batch and
its (deterministic) Dropout seed alone, and a run with and without the disturbance must agree to the last bits (fp32 atomics aside) —
with a real learning rate the random-init networks' discrete label decisions make any two runs drift apart by tens of per cent within
a few iterations (r02's trajectory comparisons passed or failed by luck). What a stale or overwritten operand does to a captured kernel
does not need a moving weight to show."""
import os

import pytest
import torch

from util import QuietLogger, p2_opts


def test_scratch_is_retired_on_growth_cpu():
    from cl4wsis_amd import ops
    sc = ops._Scratch(1 << 10)
    dev = torch.device("cpu")
    a = sc.get(dev, 100)
    assert a.numel() == 1 << 10 and sc.get(dev, 1 << 10) is a
    b = sc.get(dev, 5000)
    assert b is not a and b.numel() >= 5000 and any(t is a for t in sc.retired) and any(t is a for t in sc.live())
    c = sc.get(dev, 5001 if b.numel() < 5001 else b.numel() + 1)
    assert c.numel() >= b.numel() + b.numel() // 2          # geometric growth: retired buffers sum to less than the live one
    assert sum(t.numel() for t in sc.retired) < 2 * c.numel()
    sc.release_retired()
    assert not sc.retired and sc.live() == [c]
    assert [e[1] for e in sc.events] == [1 << 10, b.numel(), c.numel()]


def _trainer(gpu, opts, seed=31):
    from cl4wsis_amd.train import Trainer
    from oracle import weights
    torch.manual_seed(0)
    tr = Trainer(QuietLogger(), gpu, opts)
    for i, m in enumerate((tr.model, tr.model_old, tr.pseudolabeler, tr.peakgenerator)):
        m.load_state_dict(weights.fill_({k: v.cpu().clone() for k, v in m.state_dict().items()}, seed=seed + i))
    tr._realias()
    return tr


def _batches(crop, n, b=4, seed=9):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(b, 3, crop, crop, generator=g), torch.eye(20)[[16 + i % 4, 15, 17, 18][:b]].clone()) for i in range(n)]


def _rel(a, b):
    return ((a - b).abs() / a.abs().clamp_min(1e-6)).max().item()


def _frozen_lr(opts):
    opts.lr = 0.0                      # see the module docstring
    opts.max_iters, opts.start_decay = 1000, 1000
    return opts


@pytest.mark.gpu
def test_single_stream_graph_matches_two_stream_graph(gpu, monkeypatch):
    """CL4_OVERLAP=0 used to mean eager steps (r02g open issue). The captured single-stream step must now run — 10 replays — and give
    the losses of the default two-stream captured step on the same batches."""
    crop = 97
    opts = _frozen_lr(p2_opts(crop))
    batches = _batches(crop, 12)

    def run(overlap):
        monkeypatch.setenv("CL4_OVERLAP", overlap)
        tr = _trainer(gpu, opts)
        assert (tr._side_stream is None) == (overlap == "0")
        losses = [tr.step(b)["losses"].clone() for b in batches]
        torch.cuda.synchronize()
        assert tr._graph is not None, "the step was not captured"
        tr.check_label_overflow()
        return torch.stack(losses).cpu()

    two, one = run("1"), run("0")
    assert torch.isfinite(one).all()
    assert _rel(two, one) <= 1e-4, (two, one)


@pytest.mark.gpu
@pytest.mark.parametrize("lookahead", [False, True])
def test_graph_steps_survive_regrow_short_batch_and_eval_pass(gpu, lookahead):
    """Twin trainers, same batches. One of them, after step 3: takes a SHORT batch through the eager path (another label-workspace
    size — in BOTH runs, so that the Dropout seeds stay aligned), and then is disturbed: an eval pass (predict_instances), the
    derived-tensor cache cleared and rebuilt eagerly for the old model, every scratch buffer regrown on every stream key that has one
    (as a larger eager workload on that stream would), the allocator's free lists filled with canaries of the retired sizes. The
    replays that follow must leave the canaries alone and give the undisturbed run's losses. lookahead=True drives the pipelined
    two-graph path. CL4_TEST_NO_RETIRE=1 drops the retire rule, CL4_TEST_NO_PIN=1 the trainer's pin of derived tensors and scratch (r02
    behaviour) to show that the test has teeth."""
    from cl4wsis_amd import ops
    from cl4wsis_amd.nn import CACHE
    from cl4wsis_amd.train import Trainer
    crop = 97
    opts = _frozen_lr(p2_opts(crop))
    opts.val_flip = False
    batches = _batches(crop, 8)
    short = _batches(crop, 1, b=2, seed=5)[0]
    x_eval = torch.randn(1, 3, crop, crop, generator=torch.Generator().manual_seed(3))

    def run(disturb):
        tr = _trainer(gpu, opts)
        losses, canaries = [], []
        for i, b in enumerate(batches):
            nb = batches[i + 1] if lookahead and i + 1 < len(batches) else None
            losses.append(tr.step(b, next_batch=nb)["losses"].clone())
            if i == 3:
                tr.step(short)                                     # eager path (other shape) in BOTH runs
                if disturb:
                    tr.model.eval()
                    tr.predict_instances(x_eval, (crop, crop), opts)
                    tr.model.train()
                    CACHE.clear()
                    with torch.no_grad():
                        tr.model_old(x_eval.to(gpu), interpolate=False)      # rebuilds the old model's derived tensors eagerly
                    torch.cuda.synchronize()
                    sizes = []
                    for sc in (ops._conv_scratch, ops._misc_scratch):
                        for key, ws in list(sc.cur.items()):
                            if key[0] != "cuda":
                                continue
                            st = torch.cuda.ExternalStream(key[2]) if key[2] else torch.cuda.default_stream(gpu)
                            with torch.cuda.stream(st):
                                assert sc.get(gpu, 2 * ws.numel()) is not ws
                            sizes.append(ws.numel())
                            del ws
                    canaries = [torch.full((n,), 7, dtype=torch.uint8, device=gpu) for n in sizes for _ in range(2)]
                    canaries += [torch.full((n,), 7.0, device=gpu) for n in (1 << 12, 1 << 16, 1 << 20, 1 << 22) for _ in range(8)]
        torch.cuda.synchronize()
        assert (tr._pipe is not None) if lookahead else (tr._graph is not None)
        assert all(bool((c == 7).all()) for c in canaries), "a captured kernel wrote into memory the allocator had handed out again"
        tr.check_label_overflow()
        return torch.stack(losses).cpu()

    if os.environ.get("CL4_TEST_NO_RETIRE"):
        ops._Scratch.keep_retired = False
    if os.environ.get("CL4_TEST_NO_PIN"):
        Trainer._derived_snapshot = lambda self: []
    try:
        a, b = run(False), run(True)
    finally:
        ops._Scratch.keep_retired = True
    assert torch.isfinite(b).all()
    assert _rel(a, b) <= 1e-4, (a, b)
