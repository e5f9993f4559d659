"""GPU: checkpoints in the reference's own layout (tests/golden/state_dicts.json.gz: names / shapes / dtypes dumped from the
reference's modules, `module.` prefixes of run.py:17-33) go through Trainer.load_step_ckpt / load_seg_ckpt / load_ckpt
(train.py:747-812) and land in the right tensors; the resume file this build writes has the reference's layout."""
import gzip
import json
import os

import pytest
import torch

from util import QuietLogger, p2_opts

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_dicts.json.gz")


def _random_like(desc, seed):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, (shape, dtype) in desc.items():
        dt = getattr(torch, dtype)
        out[k] = (torch.rand(shape, generator=g) + 0.5).to(dt) if dt.is_floating_point else torch.full(shape, 3, dtype=dt)
    return out


def test_reference_layout_checkpoints_load_and_resume(gpu, tmp_path):
    from cl4wsis_amd import run
    from cl4wsis_amd.train import Trainer
    with gzip.open(GOLD, "rt") as f:
        gold = json.load(f)
    opts = p2_opts(65)
    opts.max_iters, opts.start_decay = 100, 100
    tr = Trainer(QuietLogger(), gpu, opts)
    pre = lambda d: {"module." + k: v for k, v in d.items()}
    # step-0 checkpoint as the reference writes it: model with classes [16] incl. the instance branch
    s0 = _random_like(gold["voc15-5_s0"]["model"], 1)
    torch.save({"model_state": pre(s0), "epoch": 29, "best_score": 0.0}, tmp_path / "step0.pth")
    opts.step_ckpt = None
    tr.load_step_ckpt(str(tmp_path / "step0.pth"))
    new, old = tr.model.state_dict(), tr.model_old.state_dict()
    assert set(old) == set(s0)
    for k, v in s0.items():
        assert torch.equal(old[k].cpu(), v), k                      # strict load of the old model
        assert torch.equal(new[k].cpu(), v), k                      # non-strict into the new one: everything but the new classifiers
    assert "cls.1.weight" in new and "cls.1.weight" not in s0
    with pytest.raises(FileNotFoundError):
        tr.load_step_ckpt(str(tmp_path / "missing.pth"))
    # phase-1 checkpoint: branch 'none' model + PseudoLabeler + PeakGenerator
    p1 = _random_like(gold["voc15-5_s1_p1"]["model"], 2)
    pl = _random_like(gold["pseudolabeler_2048_21"], 3)
    pg = _random_like(gold["peakgenerator_21_20_15"], 4)
    torch.save({"model_state": pre(p1), "pseudolabeler": pre(pl), "peakgenerator": pre(pg)}, tmp_path / "seg.pth")
    opts.seg_ckpt = str(tmp_path / "seg.pth")
    tr.load_seg_ckpt(opts.seg_ckpt)
    new = tr.model.state_dict()
    for k, v in p1.items():
        assert torch.equal(new[k].cpu(), v), k
    dec = "decoder.instance_decoder.aspp.convs.0.0.weight"
    assert torch.equal(new[dec].cpu(), s0[dec])                    # the instance branch keeps the step-0 weights (Appendix B)
    for k, v in pl.items():
        assert torch.equal(tr.pseudolabeler.state_dict()[k].cpu(), v)
    for k, v in pg.items():
        assert torch.equal(tr.peakgenerator.state_dict()[k].cpu(), v)
    # the trainable parameters still alias the flat buffer after all that loading
    p = next(iter(tr._branch.trainable_parameters()))
    assert p.data_ptr() == tr.optimizer.params_fb.views[tr.optimizer.index[id(p)]].data_ptr()
    # resume file: the layout this build writes == the reference's (run.py:17-33), and it loads back
    path = tmp_path / "resume.pth"
    run.save_ckpt(str(path), tr, 4, 0.25)
    ck = torch.load(path, map_location="cpu")
    g = gold["save_ckpt_voc15-5_s1_p2"]
    assert set(g["keys"]) <= set(ck)
    for part in ("model_state", "pseudolabeler", "peakgenerator"):
        got = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in ck[part].items()}
        assert got == g[part], part
    opts.ckpt = str(path)
    tr2 = Trainer(QuietLogger(), gpu, opts)
    epoch, best = tr2.load_ckpt(str(path))
    assert (epoch, best) == (5, 0.25)
    for k, v in tr.model.state_dict().items():
        assert torch.equal(tr2.model.state_dict()[k], v), k
