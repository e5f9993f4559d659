import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def p2_opts(crop=513, extra=()):
    """Options of BASELINE config 3 (VOC 15-5, step 1, phase 2), scripts/run.sh:39-40 + --crop_size."""
    from cl4wsis_amd import argparser
    args = ["--dataset", "voc", "--task", "15-5", "--step", "1", "--weakly", "--phase", "2", "--alpha", "0.5",
            "--loss_de", "1", "--lr_policy", "warmup", "--affinity", "--optim", "adam", "--weight_decay", "0",
            "--lr", "5e-5", "--crop_size", str(crop), "--batch_size", "16", "--no_pretrained"] + list(extra)
    return argparser.modify_command_options(argparser.get_argparser().parse_args(args))


def rel_err(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def p1_opts(crop=513, extra=()):
    """Options of BASELINE config 2 (VOC 15-5, step 1, phase 1), scripts/run.sh:31-33 + --crop_size."""
    from cl4wsis_amd import argparser
    args = ["--dataset", "voc", "--task", "15-5", "--step", "1", "--weakly", "--phase", "1", "--alpha", "0.5",
            "--loss_de", "1", "--lr_policy", "warmup", "--affinity", "--optim", "sgd", "--lr", "0.001", "--epochs", "40",
            "--crop_size", str(crop), "--batch_size", "16", "--no_pretrained"] + list(extra)
    return argparser.modify_command_options(argparser.get_argparser().parse_args(args))


def s0_opts(crop=513, extra=()):
    """Options of BASELINE config 1 (VOC 15-5, step 0, supervised), scripts/run.sh:24 + --crop_size."""
    from cl4wsis_amd import argparser
    args = ["--dataset", "voc", "--task", "15-5", "--step", "0", "--bce", "--lr", "5e-5", "--epochs", "100", "--optim", "adam",
            "--weight_decay", "0", "--crop_size", str(crop), "--batch_size", "16", "--no_pretrained"] + list(extra)
    return argparser.modify_command_options(argparser.get_argparser().parse_args(args))
