"""Command-line surface of the reference (argparser.py:4-238), table-driven.

Every flag, type, default and help group of the reference parser is kept so that existing launch scripts
(scripts/run.sh, scripts/coco.sh) work unchanged; ``modify_command_options`` applies the same derived settings
(argparser.py:4-34).
"""
import argparse

from . import tasks


def str2bool(v):
    return v.lower() in ("yes", "y", "true", "t", "1")


_T = "store_true"
_F = "store_false"

# (flag, kwargs) in the reference's order; action flags carry their reference default explicitly.
_FLAGS = [
    # performance
    ("--local_rank", dict(type=int, default=0)),
    ("--random_seed", dict(type=int, default=42)),
    ("--num_workers", dict(type=int, default=1)),
    ("--device", dict(type=int, default=None)),
    # dataset
    ("--data_root", dict(type=str, default="data")),
    ("--dataset", dict(type=str, default="voc")),
    ("--weakly", dict(action=_T, default=False)),
    ("--num_classes", dict(type=int, default=None)),
    # train
    ("--epochs", dict(type=int, default=30)),
    ("--batch_size", dict(type=int, default=24)),
    ("--crop_size", dict(type=int, default=512)),
    ("--crop_size_val", dict(type=int, default=512)),
    ("--optim", dict(type=str, default="adabelief", choices=["sgd", "adam", "adabelief", "adamw"])),
    ("--lr", dict(type=float, default=0.01)),
    ("--momentum", dict(type=float, default=0.9)),
    ("--weight_decay", dict(type=float, default=1e-4)),
    ("--lr_policy", dict(type=str, default="poly", choices=["poly", "step", "none", "warmup", "one_cycle"])),
    ("--lr_decay_step", dict(type=int, default=5000)),
    ("--lr_decay_factor", dict(type=float, default=0.1)),
    ("--lr_power", dict(type=float, default=0.9)),
    ("--bce", dict(action=_T, default=False)),
    ("--dce", dict(action=_T, default=False)),
    # validation
    ("--val_on_trainset", dict(action=_T, default=False)),
    ("--crop_val", dict(action=_F, default=True)),
    # logging
    ("--logdir", dict(type=str, default="./logs")),
    ("--name", dict(type=str, default="Experiment")),
    ("--sample_num", dict(type=int, default=8)),
    ("--debug", dict(action=_T, default=False)),
    ("--visualize", dict(action=_F, default=True)),
    ("--print_interval", dict(type=int, default=10)),
    ("--val_interval", dict(type=int, default=5)),
    # model
    ("--model", dict(type=str, default="PanopticDeepLab", choices=["PanopticDeepLab"])),
    ("--backbone", dict(type=str, default="resnet101", choices=["resnet50", "resnet101", "wider_resnet38_a2"])),
    ("--output_stride", dict(type=int, default=16, choices=[8, 16])),
    ("--no_pretrained", dict(action=_T, default=False)),
    ("--norm_act", dict(type=str, default="iabn_sync")),
    ("--pooling", dict(type=int, default=32)),
    # test / checkpoints
    ("--test", dict(action=_T, default=False)),
    ("--ckpt", dict(type=str, default=None)),
    ("--continue_ckpt", dict(action=_T, default=False)),
    ("--ckpt_interval", dict(type=int, default=1)),
    ("--seg_ckpt", dict(type=str, default=None)),
    # distillation (ILTSS)
    ("--freeze", dict(action=_T, default=False)),
    ("--freeze_seg", dict(action=_T, default=False)),
    ("--loss_de", dict(type=float, default=0.)),
    ("--loss_kd", dict(type=float, default=0.)),
    # ICaRL
    ("--icarl", dict(action=_T, default=False)),
    ("--icarl_importance", dict(type=float, default=1.)),
    ("--icarl_disjoint", dict(action=_T, default=False)),
    ("--icarl_bkg", dict(type=float, default=-1)),
    # methods
    ("--init_balanced", dict(action=_T, default=False)),
    ("--unkd", dict(action=_T, default=False)),
    ("--unce", dict(action=_T, default=False)),
    # incremental
    ("--task", dict(type=str, default="19-1", choices=tasks.get_task_list())),
    ("--step", dict(type=int, default=0)),
    ("--no_mask", dict(action=_T, default=False)),
    ("--overlap", dict(action=_T, default=False)),
    ("--step_ckpt", dict(type=str, default=None)),
    ("--phase", dict(type=int, default=None)),
    # weak supervision
    ("--pseudo", dict(type=str, default=None)),
    ("--pl_ckpt", dict(type=str, default=None)),
    ("--alpha", dict(type=float, default=0.5)),
    ("--pos_w", dict(type=float, default=1.)),
    ("--affinity", dict(action=_T, default=False)),
    ("--affinity_method", dict(type=str, default="pamr", choices=["pamr"])),
    ("--pseudo_ep", dict(type=int, default=5)),
    ("--lr_pseudo", dict(type=float, default=0.01)),
    ("--lr_head", dict(type=float, default=10.)),
    ("--cam", dict(type=str, default="ngwp")),
    ("--ss_dist", dict(action=_T, default=False)),
    ("--l_seg", dict(type=float, default=1)),
    # panoptic-deeplab label generation
    ("--val_thresh", dict(type=float, default=0.1)),
    ("--val_kernel", dict(type=int, default=41)),
    ("--val_flip", dict(type=str2bool, default=False)),
    ("--val_clean", dict(type=str2bool, default=False)),
    ("--val_ignore", dict(type=str2bool, default=False)),
    ("--pseudo_thresh", dict(type=float, default=0.7)),
    ("--refine_thresh", dict(type=float, default=0.3)),
    ("--kernel", dict(type=int, default=41)),
    ("--sigma", dict(type=int, default=6)),
    ("--beta", dict(type=float, default=3.0)),
    ("--detach_instance", dict(action=_T, default=False)),
    ("--run_refine", dict(type=str2bool, default=True)),
    # peak generator
    ("--pam_alpha", dict(type=int, default=0.7)),
    ("--peak_from", dict(type=str, default="peakgenerator")),
    # branches
    ("--branch", dict(type=str, default="all", choices=["all", "seg", "ins", "none"])),
    ("--use_DeeplabV3_as_seg_branch", dict(action=_T, default=False)),
    # continual WSSS
    ("--flac", dict(action=_T, default=False)),
    ("--randrop", dict(action=_T, default=False)),
]


def get_argparser():
    parser = argparse.ArgumentParser()
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    return parser


def modify_command_options(opts):
    """Derived options — reference argparser.py:4-34 (same order of overrides)."""
    opts.num_classes = {"voc": 21, "coco": 80}.get(opts.dataset, opts.num_classes)
    if not opts.visualize:
        opts.sample_num = 0
    if opts.dataset == "coco-voc":
        opts.backbone, opts.output_stride = "wider_resnet38_a2", 8
        opts.crop_size, opts.crop_size_val = 448, 512
    opts.use_DeeplabV3_as_seg_branch = True
    opts.branch = "none" if opts.phase == 1 else "ins"
    if opts.phase == 1:
        opts.flac = opts.randrop = True
    if opts.phase == 2:
        opts.freeze = opts.freeze_seg = True
    opts.no_overlap = not opts.overlap
    opts.pooling = opts.crop_size // opts.output_stride
    if opts.step == 0:
        opts.lr_head = 1.
    return opts
