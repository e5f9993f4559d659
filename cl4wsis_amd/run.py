"""Process entry with the reference's CLI (run.py:37-183): one process per GPU, RCCL bootstrap from the launcher's
environment, seeding, loaders, `Trainer`, epoch loop, checkpoints.

    python -m torch.distributed.run --nproc-per-node 8 -m cl4wsis_amd.run --dataset voc --task 15-5 --step 1 \\
        --weakly --phase 2 --optim adam --lr 5e-5 --weight_decay 0 --lr_policy warmup --batch_size 16 ...

The reference's data pipeline (`dataset/`) is outside the hot-path scope: when a module named
`dataset` with the reference's `get_dataset(opts)` is importable it is used unchanged, otherwise `--data_root synthetic`
drives the step with VOC-shaped synthetic batches (the same generator bench.py uses).
"""
import os
import random

import numpy as np
import torch
from torch.utils import data
from torch.utils.data.distributed import DistributedSampler

from . import argparser
from .parallel import init_distributed
from .train import Trainer


class SyntheticVOC(data.Dataset):
    """Loader-contract stand-in (dataset/dataset.py:245): (images, seg, center, offset, weight, l1h)."""

    def __init__(self, n, crop, seed):
        self.n, self.crop, self.seed = n, crop, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        c = self.crop
        l1h = torch.zeros(20)
        k = 1 if torch.rand(1, generator=g).item() < 0.6 else 2
        for cls in torch.randperm(5, generator=g)[:k].tolist():
            l1h[15 + cls] = 1
        z = torch.zeros(1)
        return torch.randn(3, c, c, generator=g), z, z, z, z, l1h


class _Logger:
    def __init__(self, rank):
        self.rank = rank

    def info(self, msg):
        if self.rank == 0:
            print(msg, flush=True)

    debug = info
    print = info

    def add_scalar(self, *a, **k):
        pass

    def commit(self, *a, **k):
        pass

    def close(self):
        pass


def save_ckpt(path, trainer, epoch, best_score):
    """run.py:17-33 — same dictionary keys; model keys carry the `module.` prefix the reference's loaders strip."""
    pref = lambda sd: {"module." + k: v.detach().cpu().clone() for k, v in sd.items()}
    state = {"epoch": epoch, "model_state": pref(trainer.model.state_dict()),
             "optimizer_state": trainer.optimizer.state_dict(), "scheduler_state": trainer.scheduler.state_dict(),
             "scaler": trainer.scaler.state_dict(), "best_score": best_score}
    if trainer.pseudolabeler is not None:
        state["pseudolabeler"] = pref(trainer.pseudolabeler.state_dict())
    if trainer.peakgenerator is not None:
        state["peakgenerator"] = pref(trainer.peakgenerator.state_dict())
    torch.save(state, path)


def main(opts):
    rank, world, local = init_distributed()
    device = torch.device("cuda", local) if torch.cuda.is_available() else None
    if device is None:
        raise SystemExit("cl4wsis_amd.run needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    opts.device_id = local
    task_name = f"{opts.dataset}-{opts.task}" + ("-ov" if opts.overlap else "")
    os.makedirs(f"checkpoints/step/{task_name}", exist_ok=True)
    path = f"checkpoints/step/{task_name}/{opts.name}_{opts.step}.pth"
    torch.manual_seed(opts.random_seed)
    np.random.seed(opts.random_seed)
    random.seed(opts.random_seed)
    opts.batch_size = opts.batch_size // world                       # run.py:66
    try:
        from dataset import get_dataset                              # the reference's pipeline, if present
        train_dst, val_dst, test_dst, labels, n_classes = get_dataset(opts)
    except ImportError:
        train_dst = SyntheticVOC(10582, opts.crop_size, opts.random_seed)
    sampler = DistributedSampler(train_dst, num_replicas=world, rank=rank) if world > 1 else None
    loader = data.DataLoader(train_dst, batch_size=opts.batch_size, sampler=sampler, shuffle=sampler is None,
                             num_workers=opts.num_workers, drop_last=True)
    opts.max_iters = opts.epochs * len(loader)
    opts.start_decay = opts.pseudo_ep * len(loader)
    logger = _Logger(rank)
    trainer = Trainer(logger, device, opts)
    if opts.step > 0 and opts.step_ckpt is not None:
        trainer.load_step_ckpt(opts.step_ckpt)
    cur_epoch, best = 0, 0.0
    if opts.continue_ckpt and os.path.isfile(path):
        cur_epoch, best = trainer.load_ckpt(path)
    elif opts.phase == 2 and opts.seg_ckpt is not None:
        trainer.load_seg_ckpt(opts.seg_ckpt)
    while cur_epoch < opts.epochs and not opts.test:
        epoch_loss = trainer.train(cur_epoch=cur_epoch, train_loader=loader, print_int=opts.print_interval)
        logger.info(f"End of Epoch {cur_epoch}/{opts.epochs}, Average Loss={float(epoch_loss[0]) + float(epoch_loss[1])}")
        if rank == 0 and (cur_epoch + 1) % opts.ckpt_interval == 0:   # the reference writes from every rank (race)
            save_ckpt(path, trainer, cur_epoch, best)
        cur_epoch += 1
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    logger.close()


if __name__ == "__main__":
    parser = argparser.get_argparser()
    main(argparser.modify_command_options(parser.parse_args()))
