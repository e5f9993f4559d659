"""GPU: the reference's continual-learning flow end to end through the CLI entry (cl4wsis_amd.run.main with the synthetic
loader): step 0 (supervised) -> checkpoint -> step 1 phase 1 (--step_ckpt) -> checkpoint -> step 1 phase 2 (--step_ckpt,
--seg_ckpt). Checks the orchestration, the checkpoint dictionary contract of run.py:17-33 / train.py:747-812 (module.
prefixes, optimizer / scheduler / pseudolabeler / peakgenerator entries) and that every phase trains (finite losses,
weights move)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(args, tmp, monkeypatch, autotune="1"):
    from cl4wsis_amd import argparser, run
    monkeypatch.chdir(tmp)
    monkeypatch.setenv("CL4_SYNTH_LEN", "4")
    monkeypatch.setenv("CL4_AUTOTUNE", autotune)          # the CLI flow runs with the warm-up autotuner, like production
    from cl4wsis_amd import ops
    monkeypatch.setattr(ops, "_SPLITS", [1, 2, -4])       # ... over a short list of K splits (13 by default): the tuner's time is not the point here
    opts = argparser.modify_command_options(argparser.get_argparser().parse_args(args))
    run.main(opts)


def test_step0_phase1_phase2_chain(gpu, tmp_path, monkeypatch):
    common = ["--dataset", "voc", "--task", "15-5", "--crop_size", "65", "--batch_size", "2", "--epochs", "1", "--num_workers", "0",
              "--no_pretrained", "--name", "T", "--print_interval", "1"]
    _run(common + ["--step", "0", "--bce", "--lr", "5e-5", "--optim", "adam", "--weight_decay", "0"], tmp_path, monkeypatch)
    ck0 = tmp_path / "checkpoints/step/voc-15-5/T_0.pth"
    assert ck0.is_file()
    c0 = torch.load(ck0, map_location="cpu")
    assert {"epoch", "model_state", "optimizer_state", "scheduler_state", "scaler", "best_score"} <= set(c0)
    assert all(k.startswith("module.") for k in c0["model_state"])
    assert "module.instance_head.classifier.center.cls.0.weight" in c0["model_state"]

    weak = ["--step", "1", "--weakly", "--alpha", "0.5", "--loss_de", "1", "--lr_policy", "warmup", "--affinity", "--step_ckpt", str(ck0)]
    _run(common + weak + ["--phase", "1", "--optim", "sgd", "--lr", "0.001", "--pseudo_ep", "0"], tmp_path, monkeypatch)
    ck1 = tmp_path / "checkpoints/step/voc-15-5/T_1.pth"
    c1 = torch.load(ck1, map_location="cpu")
    assert "pseudolabeler" in c1 and "peakgenerator" in c1 and "module.cls.1.weight" in c1["model_state"]
    # the body trained in phase 1: it differs from the step-0 weights it was initialised with
    k = "module.body.mod5.block3.convs.conv3.weight"
    assert not torch.equal(c1["model_state"][k], c0["model_state"][k])
    seg_ck = tmp_path / "seg.pth"
    os.replace(ck1, seg_ck)

    _run(common + weak + ["--phase", "2", "--optim", "adam", "--lr", "5e-5", "--weight_decay", "0", "--seg_ckpt", str(seg_ck)],
         tmp_path, monkeypatch)
    c2 = torch.load(ck1, map_location="cpu")
    # phase 2 freezes body / head / cls (they equal the phase-1 weights) and trains the instance branch
    assert torch.equal(c2["model_state"][k], c1["model_state"][k])
    ki = "module.decoder.instance_decoder.aspp.convs.0.0.weight"
    assert not torch.equal(c2["model_state"][ki], c0["model_state"][ki])
    for ck in (c0, c1, c2):
        assert all(torch.isfinite(v).all() for v in ck["model_state"].values() if v.is_floating_point())


def test_cli_trains_from_a_voc_tree_through_the_device_loader(gpu, tmp_path, monkeypatch):
    """run.main finds <data_root>/voc/pascal_sbd_train.json and switches from the synthetic loader to cl4wsis_amd.dataset
    (host decode, device-side crop / resample / normalise / label_generation): one supervised step-0 epoch and one weakly
    supervised phase-2 epoch on the tiny VOC-shaped tree, with worker processes."""
    from util import make_mini_voc
    make_mini_voc(tmp_path / "data", n_train=8, n_val=2)
    # worker processes (each one pages the whole torch image in: ~10 s per loader on a fresh box) and the warm-up autotuner only in the
    # phase-2 run, where the device loader's weakly-supervised path is the point; the step-0 run decodes in the main process
    common = ["--dataset", "voc", "--task", "15-5", "--crop_size", "65", "--batch_size", "2", "--epochs", "1",
              "--no_pretrained", "--name", "D", "--print_interval", "1", "--data_root", str(tmp_path / "data")]
    _run(common + ["--num_workers", "0", "--step", "0", "--bce", "--lr", "5e-5", "--optim", "adam", "--weight_decay", "0"], tmp_path, monkeypatch,
         autotune="0")
    ck0 = tmp_path / "checkpoints/step/voc-15-5/D_0.pth"
    c0 = torch.load(ck0, map_location="cpu")
    assert all(torch.isfinite(v).all() for v in c0["model_state"].values() if v.is_floating_point())
    _run(common + ["--num_workers", "1", "--step", "1", "--weakly", "--alpha", "0.5", "--loss_de", "1", "--lr_policy", "warmup", "--affinity", "--phase", "2",
                   "--optim", "adam", "--lr", "5e-5", "--weight_decay", "0", "--step_ckpt", str(ck0)], tmp_path, monkeypatch)
    c1 = torch.load(tmp_path / "checkpoints/step/voc-15-5/D_1.pth", map_location="cpu")
    ki = "module.decoder.instance_decoder.aspp.convs.0.0.weight"
    assert not torch.equal(c1["model_state"][ki], c0["model_state"][ki])
    assert all(torch.isfinite(v).all() for v in c1["model_state"].values() if v.is_floating_point())
