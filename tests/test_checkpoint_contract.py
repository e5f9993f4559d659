"""CPU: the checkpoint contract against the reference's own modules — tests/golden/state_dicts.json.gz
holds name / shape / dtype of every state-dict entry the reference instantiates (make_golden.state_dict_fixture). A checkpoint
written by the reference must load here and vice versa, including the `module.` prefixes of run.py:17-33 and the pretrained
backbone file layout of segmentation_module.py:37-57."""
import gzip
import json
import os

import pytest
import torch

from cl4wsis_amd import argparser

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_dicts.json.gz")
BASE = ["--lr_policy", "warmup", "--no_pretrained", "--crop_size", "65"]
CFGS = {
    "voc15-5_s0": ["--dataset", "voc", "--task", "15-5", "--step", "0", "--bce", "--optim", "adam"],
    "voc15-5_s1_p1": ["--dataset", "voc", "--task", "15-5", "--step", "1", "--weakly", "--phase", "1", "--affinity", "--optim", "sgd"],
    "voc15-5_s1_p2": ["--dataset", "voc", "--task", "15-5", "--step", "1", "--weakly", "--phase", "2", "--affinity", "--optim", "adam"],
    "voc10-10_s1_p2": ["--dataset", "voc", "--task", "10-10", "--step", "1", "--weakly", "--phase", "2", "--affinity", "--optim", "adam"],
    "cocovoc_s1_p2": ["--dataset", "coco-voc", "--task", "voc", "--step", "1", "--weakly", "--phase", "2", "--affinity", "--optim", "adam"],
}


@pytest.fixture(scope="module")
def gold():
    with gzip.open(GOLD, "rt") as f:
        return json.load(f)


def _opts(args, base=BASE):
    return argparser.modify_command_options(argparser.get_argparser().parse_args(args + base))


def _desc(sd, prefix=""):
    return {prefix + k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()}


@pytest.mark.parametrize("name", sorted(CFGS))
def test_model_state_dicts_match_the_reference(gold, name):
    from cl4wsis_amd import segmentation_module as sm
    g = gold[name]
    o = _opts(CFGS[name])
    assert o.branch == g["branch"] and o.backbone == g["backbone"]
    assert _desc(sm.make_model(o, classes=g["classes"]).state_dict()) == g["model"]
    if "model_old" in g:
        assert _desc(sm.make_model(o, classes=g["classes"][:-1]).state_dict()) == g["model_old"]


def test_wss_module_state_dicts_match_the_reference(gold):
    from cl4wsis_amd import segmentation_module as sm
    from cl4wsis_amd.wss import PeakGenerator, PseudoLabeler
    o = _opts(CFGS["voc15-5_s1_p2"])
    assert _desc(PseudoLabeler(2048, 21, sm.get_norm(o)).state_dict()) == gold["pseudolabeler_2048_21"]
    assert _desc(PseudoLabeler(4096, 81, sm.get_norm(o)).state_dict()) == gold["pseudolabeler_4096_81"]
    assert _desc(PeakGenerator(in_channel=21, num_classes=20, old_classes=15).state_dict()) == gold["peakgenerator_21_20_15"]


def _random_like(desc, seed):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, (shape, dtype) in desc.items():
        dt = getattr(torch, dtype)
        out[k] = torch.randn(shape, generator=g).to(dt) if dt.is_floating_point else torch.zeros(shape, dtype=dt)
    return out


def test_pretrained_backbone_file_layout_loads(gold, tmp_path, monkeypatch):
    """get_body (segmentation_module.py:37-57): `pretrained/resnet101_iabn_sync.pth.tar` = {'state_dict': {...}} with `module.`
    prefixes and the ImageNet classifier, which is dropped."""
    from cl4wsis_amd import segmentation_module as sm
    pre = _random_like(gold["pretrained_resnet101_iabn_sync"]["state_dict"], 3)
    os.makedirs(tmp_path / "pretrained")
    torch.save({"state_dict": pre}, tmp_path / "pretrained" / "resnet101_iabn_sync.pth.tar")
    monkeypatch.chdir(tmp_path)
    o = _opts(CFGS["voc15-5_s1_p2"], base=["--lr_policy", "warmup", "--crop_size", "65"])      # no --no_pretrained
    model = sm.make_model(o, classes=[16, 5])
    sd = model.body.state_dict()
    assert "classifier.fc.weight" not in sd
    for k, v in sd.items():
        assert torch.equal(v, pre["module." + k]), k
    # a negative InPlaceABN scale survives loading untouched (it is the ABN arithmetic that treats it, see test_abn_gpu.py)
    assert (sd["mod1.bn1.weight"] < 0).any()


def test_save_ckpt_layout_matches_the_reference(gold):
    g = gold["save_ckpt_voc15-5_s1_p2"]
    assert all(k.startswith("module.") for k in g["model_state"])
    assert {k[7:] for k in g["model_state"]} == set(gold["voc15-5_s1_p2"]["model"])
    assert {k[7:] for k in g["pseudolabeler"]} == set(gold["pseudolabeler_2048_21"])
    assert {k[7:] for k in g["peakgenerator"]} == set(gold["peakgenerator_21_20_15"])
