"""CPU: host-side logic that needs no GPU — CLI surface, task splits, schedules, flat buffers, state-dict contract."""
import pytest
import torch

from util import p2_opts


def test_p2_derived_options():
    o = p2_opts(513)
    assert (o.branch, o.freeze, o.freeze_seg, o.pooling, o.use_DeeplabV3_as_seg_branch) == ("ins", True, True, 32, True)
    assert o.run_refine is True and o.kernel == 41 and o.pseudo_thresh == 0.7 and o.lr_head == 10.0


def test_phase1_and_step0_options():
    from cl4wsis_amd import argparser as ap
    o = ap.modify_command_options(ap.get_argparser().parse_args(["--step", "1", "--phase", "1", "--task", "15-5"]))
    assert o.branch == "none" and o.flac and o.randrop
    o = ap.modify_command_options(ap.get_argparser().parse_args(["--step", "0", "--task", "10-10"]))
    assert o.lr_head == 1.0 and o.branch == "ins"
    o = ap.modify_command_options(ap.get_argparser().parse_args(["--dataset", "coco-voc", "--task", "voc", "--crop_size", "513"]))
    assert (o.backbone, o.output_stride, o.crop_size, o.pooling) == ("wider_resnet38_a2", 8, 448, 56)


def test_task_class_counts():
    from cl4wsis_amd import tasks
    assert tasks.get_per_task_classes("voc", "15-5", 1) == [16, 5]
    assert tasks.get_per_task_classes("voc", "10-10", 1) == [11, 10]
    assert tasks.get_per_task_classes("coco-voc", "voc", 1) == [61, 20]
    labels, old, name = tasks.get_task_labels("voc", "15-5", 1)
    assert labels == [16, 17, 18, 19, 20] and old == list(range(16)) and name == "voc/15-5"
    with pytest.raises(AssertionError):
        tasks.get_per_task_classes("voc", "15-5", 7)


def test_state_dict_contract():
    """Key names of SURVEY.md Appendix B (checkpoint contract)."""
    from cl4wsis_amd import segmentation_module as sm
    m = sm.make_model(p2_opts(65), classes=[16, 5])
    sd = m.state_dict()
    for k, shape in {"body.mod1.conv1.weight": (64, 3, 7, 7), "body.mod1.bn1.running_mean": (64,),
                     "body.mod4.block23.convs.conv2.weight": (256, 256, 3, 3), "body.mod2.block1.proj_conv.weight": (256, 64, 1, 1),
                     "head.map_convs.3.weight": (256, 2048, 3, 3), "head.red_bn.running_var": (256,),
                     "cls.0.weight": (16, 256, 1, 1), "cls.1.bias": (5,),
                     "decoder.instance_decoder.aspp.convs.0.0.weight": (256, 2048, 1, 1),
                     "decoder.instance_decoder.aspp.convs.4.aspp_pooling.1.weight": (256, 2048, 1, 1),
                     "decoder.instance_decoder.aspp.project.1.num_batches_tracked": (),
                     "decoder.instance_decoder.fuse.0.0.0.0.weight": (320, 1, 5, 5),
                     "instance_head.classifier.center.fuse.0.0.0.weight": (128, 1, 5, 5),
                     "instance_head.classifier.center.cls.1.bias": (5,),
                     "instance_head.classifier.offset.cls.0.weight": (2, 32, 1, 1)}.items():
        assert tuple(sd[k].shape) == shape, k
    assert len(sd) == 658
    assert sum(p.numel() for p in m.parameters()) == 73789339
    assert not any(k.endswith("num_batches_tracked") for k in sd if k.startswith("body."))


def test_warmup_poly_schedule():
    from cl4wsis_amd.scheduler import WarmUpPolyLR
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([{"params": [p], "lr": 0.5}], lr=0.1)
    s = WarmUpPolyLR(opt, max_iters=10, power=0.9, start_decay=3)
    lrs = []
    for _ in range(6):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        s.step()
    assert lrs[:3] == [0.5, 0.5, 0.5]
    assert abs(lrs[3] - 0.5 * (1 - 3 / 10) ** 0.9) < 1e-12 and abs(lrs[5] - 0.5 * (1 - 5 / 10) ** 0.9) < 1e-12


def test_flat_buffer_and_buckets_single_process():
    from cl4wsis_amd.parallel import FlatBuffer, GradBuckets
    fb = FlatBuffer([(3, 4), (5,), (2, 2, 2)], torch.device("cpu"))
    assert fb.numel == 25 and fb.offsets == [0, 12, 17, 25]
    fb.views[1].fill_(2.0)
    assert fb.flat[12:17].eq(2).all() and fb.flat[:12].eq(0).all()
    gb = GradBuckets(fb, bucket_bytes=48)
    assert gb.bounds == [(0, 1), (1, 3)]
    gb.ready(2); gb.ready(0); gb.ready(1)
    gb.finish()          # world size 1: nothing to reduce, values untouched
    assert fb.flat[12:17].eq(2).all()


def test_trainer_needs_gpu_library_and_phase2():
    from cl4wsis_amd import ops
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 4, 8, 8), torch.zeros(4, 8), 1)   # CPU tensor: no fallback path


def test_gaussian_table_matches_definition():
    import numpy as np
    from cl4wsis_amd.train import gaussian
    g = gaussian(6)
    assert g.shape == (39, 39) and g[19, 19] == 1.0
    assert abs(g[19, 25] - np.exp(-36 / 72)) < 1e-15


def test_derived_tensor_cache_invalidation_and_snapshot():
    """nn.CACHE: entries are rebuilt when the source tensor changes (version bump, external-write touch, clear) and `snapshot` holds
    the tensors a captured graph reads, so that a rebuild cannot free them (Trainer._derived_snapshot)."""
    import gc
    import weakref
    from cl4wsis_amd.nn import _Cache
    cache = _Cache()
    m = torch.nn.Conv2d(2, 2, 1)
    builds = []

    def build():
        builds.append(1)
        return m.weight.detach().clone()

    a = cache.get(m, "k", (m.weight,), build)
    assert cache.get(m, "k", (m.weight,), build) is a and len(builds) == 1          # hit
    cache.touch(m.weight)                                                          # a kernel wrote the weight behind torch's back
    b = cache.get(m, "k", (m.weight,), build)
    assert b is not a and len(builds) == 2
    with torch.no_grad():
        m.weight.add_(1.0)                                                         # in-place torch op: _version
    c = cache.get(m, "k", (m.weight,), build)
    assert c is not b and len(builds) == 3
    keep = cache.snapshot([torch.nn.Sequential(m), None])
    assert len(keep) == 1 and keep[0] is c
    ref = weakref.ref(c)
    del a, b, c
    cache.clear()
    d = cache.get(m, "k", (m.weight,), build)                                      # rebuilt: the store now holds d
    assert len(builds) == 4 and d is not keep[0]
    gc.collect()
    assert ref() is keep[0]                                                        # still alive: the snapshot pins what the graph captured
    del keep
    gc.collect()
    assert ref() is None
