"""Process entry with the reference's CLI (run.py:37-183): one process per GPU, RCCL bootstrap from the launcher's
environment, seeding, loaders, `Trainer`, epoch loop, checkpoints.

    python -m torch.distributed.run --nproc-per-node 8 -m cl4wsis_amd.run --dataset voc --task 15-5 --step 1 \\
        --weakly --phase 2 --optim adam --lr 5e-5 --weight_decay 0 --lr_policy warmup --batch_size 16 ...

Data: when `<data_root>/voc/pascal_sbd_train.json` (or the COCO tree of the coco-voc task) exists, the loaders come from
`cl4wsis_amd.dataset` — the mirror of the reference's `dataset/` package (`get_dataset`, the VOC / COCO classes, the incremental
wrapper) whose pixel work (crop, Pillow-exact resampling, normalisation, label_generation) runs on the device behind
`DeviceLoader`; without such a tree (`--data_root synthetic`, the default of the tests and of bench.py) the step is driven
by VOC-shaped synthetic batches from `cl4wsis_amd.synthetic`.
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
    """Loader-contract stand-in (dataset/dataset.py:245): (images, seg, center, offset, weight, l1h). The weakly-supervised
    phases only read images and l1h (train.py:243-248); `supervised=True` (step 0) also synthesises the dense targets:
    a semantic map of rectangles over the 15 base classes (+255 borders), a gaussian centre bump per rectangle in its class
    channel, offsets to the rectangle centre and unit weights inside the rectangles."""

    def __init__(self, n, crop, seed, supervised=False, n_classes=16):
        self.n, self.crop, self.seed, self.supervised, self.n_classes = n, crop, seed, supervised, n_classes

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        c = self.crop
        img = torch.randn(3, c, c, generator=g)
        if not self.supervised:
            l1h = torch.zeros(20)
            k = 1 if torch.rand(1, generator=g).item() < 0.6 else 2
            for cls in torch.randperm(5, generator=g)[:k].tolist():
                l1h[15 + cls] = 1
            z = torch.zeros(1)
            return img, z, z, z, z, l1h
        nc = self.n_classes - 1
        seg = torch.zeros(c, c, dtype=torch.long)
        center = torch.zeros(nc, c, c)
        offset = torch.zeros(2, c, c)
        weight = torch.zeros(1, c, c)
        l1h = torch.zeros(nc)
        ys, xs = torch.meshgrid(torch.arange(c, dtype=torch.float32), torch.arange(c, dtype=torch.float32), indexing="ij")
        for _ in range(int(torch.randint(1, 4, (1,), generator=g))):
            cls = int(torch.randint(1, self.n_classes, (1,), generator=g))
            h, w = (int(v) for v in torch.randint(max(c // 8, 2), max(c // 3, 3), (2,), generator=g))
            y0, x0 = int(torch.randint(0, c - h, (1,), generator=g)), int(torch.randint(0, c - w, (1,), generator=g))
            seg[y0:y0 + h, x0:x0 + w] = cls
            cy, cx = y0 + h // 2, x0 + w // 2
            center[cls - 1] = torch.maximum(center[cls - 1], torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / 72.0))
            offset[0, y0:y0 + h, x0:x0 + w] = (cy - ys)[y0:y0 + h, x0:x0 + w]
            offset[1, y0:y0 + h, x0:x0 + w] = (cx - xs)[y0:y0 + h, x0:x0 + w]
            weight[0, y0:y0 + h, x0:x0 + w] = 1.0
            l1h[cls - 1] = 1
        seg[:2] = 255
        return img, seg, center, offset, weight, l1h


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
    """run.py:37-183 — same flow: loaders, Trainer, previous-step weights, resume / phase-1 weights, epoch loop with a checkpoint
    per epoch and validation every `val_interval`, final checkpoint, test pass."""
    rank, world, local = init_distributed()
    device = torch.device("cuda", local) if torch.cuda.is_available() else None
    if device is None:
        raise SystemExit("cl4wsis_amd.run needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    opts.device_id = local
    task_name = f"{opts.dataset}-{opts.task}" + ("-ov" if (opts.overlap and opts.dataset == "voc") else "")   # run.py:45-47
    os.makedirs(f"checkpoints/step/{task_name}", exist_ok=True)
    ckpt_path = f"checkpoints/step/{task_name}/{opts.name}_{opts.step}.pth"
    torch.manual_seed(opts.random_seed)
    np.random.seed(opts.random_seed)
    random.seed(opts.random_seed)
    opts.batch_size = opts.batch_size // world                       # run.py:66
    supervised = opts.step == 0 and not opts.weakly
    logger = _Logger(rank)
    val_loader = test_loader = None
    real_data = os.path.isfile(os.path.join(opts.data_root, "voc", "pascal_sbd_train.json")) or opts.dataset == "coco-voc"
    if real_data:
        # the real photos: host decode + device-side crop / resample / normalise / target generation (cl4wsis_amd/dataset)
        from .dataset import DeviceLoader, get_dataset
        train_dst, val_dst, test_dst, labels, n_classes = get_dataset(opts)
        random.seed(opts.random_seed)                                # run.py:68-69
        sampler = DistributedSampler(train_dst, num_replicas=world, rank=rank) if world > 1 else None
        loader = DeviceLoader(train_dst, opts.batch_size, device, sampler=sampler, shuffle=True, num_workers=opts.num_workers,
                              drop_last=True, supervised=supervised)
        val_loader = DeviceLoader(val_dst, 1, device, shuffle=False, num_workers=opts.num_workers)     # run.py:75-77 (every rank)
        test_loader = DeviceLoader(test_dst, 1, device, shuffle=False, num_workers=opts.num_workers)   # run.py:162-164
        logger.info(f"Dataset: {opts.dataset}, Train set: {len(train_dst)}, Val set: {len(val_dst)}, Test set: {len(test_dst)}, "
                    f"n_classes {n_classes}")
    else:
        n_img = int(os.environ.get("CL4_SYNTH_LEN", "10582"))        # VOC 2012 aug train size
        train_dst = SyntheticVOC(n_img, opts.crop_size, opts.random_seed, supervised=supervised)
        sampler = DistributedSampler(train_dst, num_replicas=world, rank=rank) if world > 1 else None
        from .dataset.loader import worker_options
        loader = data.DataLoader(train_dst, batch_size=opts.batch_size, sampler=sampler, shuffle=sampler is None,
                                 num_workers=opts.num_workers, drop_last=True, **worker_options(opts.num_workers))
        n_classes = 21
        logger.info(f"no dataset under {opts.data_root!r}: synthetic VOC-shaped batches ({n_img} images); validation / test are skipped")
    logger.info(f"Total batch size is {opts.batch_size * world}")
    opts.max_iters = opts.epochs * len(loader)
    if opts.lr_policy == "warmup":                                   # run.py:82-83
        opts.start_decay = opts.pseudo_ep * len(loader)
    trainer = Trainer(logger, device, opts)
    if opts.step > 0:                                                # run.py:91-98: the previous step's weights are mandatory
        path = opts.step_ckpt if opts.step_ckpt is not None else f"checkpoints/step/{task_name}/{opts.name}_{opts.step - 1}.pth"
        if os.environ.get("CL4_ALLOW_RANDOM_INIT") == "1" and not os.path.exists(path):
            logger.info(f"[!] {path} not found; CL4_ALLOW_RANDOM_INIT=1: old and new model start from the random initialisation")
        else:
            trainer.load_step_ckpt(path)                             # FileNotFoundError when missing (train.py:771)
    if opts.continue_ckpt:                                           # run.py:101-110
        opts.ckpt = ckpt_path
    if opts.ckpt is not None:
        cur_epoch, best_score = trainer.load_ckpt(opts.ckpt)
    else:
        if opts.seg_ckpt is not None:
            trainer.load_seg_ckpt(opts.seg_ckpt)
        logger.info("[!] Start from epoch 0")
        cur_epoch, best_score = 0, 0.0
    TRAIN = not opts.test
    val_metrics = None
    if real_data:
        from .metrics import StreamSegMetrics
        val_metrics = StreamSegMetrics(n_classes)
    while cur_epoch < opts.epochs and TRAIN:
        epoch_loss = trainer.train(cur_epoch=cur_epoch, train_loader=loader, print_int=opts.print_interval)
        logger.info(f"End of Epoch {cur_epoch}/{opts.epochs}, Average Loss={float(epoch_loss[0]) + float(epoch_loss[1])}, "
                    f"Class Loss={float(epoch_loss[0])}, Reg Loss={float(epoch_loss[1])}")
        if rank == 0:                                                # run.py:131 writes from every rank (a race on one file)
            save_ckpt(ckpt_path, trainer, cur_epoch, 0)
        if (cur_epoch + 1) % opts.val_interval == 0 and val_loader is not None:      # run.py:133-155
            if (opts.phase is None or opts.phase == 2) and opts.model == "PanopticDeepLab":
                logger.info("validate on val set...")
                val_score = trainer.validate(val_loader, opts)
                logger.info(val_score)
            if opts.phase == 1 and opts.weakly:
                score = trainer.validate_semseg(loader=val_loader, metrics=val_metrics, opts=opts)
                logger.add_scalar("Val_CAM/MeanIoU", score["Mean IoU"], cur_epoch)
                logger.info(val_metrics.to_str(score))
            logger.info(f"End of Validation {cur_epoch}/{opts.epochs}")
        cur_epoch += 1
    if rank == 0 and TRAIN:                                          # run.py:158-160
        save_ckpt(ckpt_path, trainer, cur_epoch, best_score)
        logger.info("[!] Checkpoint saved.")
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    results = None
    if test_loader is not None:                                      # run.py:166-181
        logger.info("*** Test the model on all seen classes...")
        if opts.phase == 1:
            results = trainer.validate_semseg(loader=test_loader, metrics=val_metrics, opts=opts)
            logger.info(val_metrics.to_str(results))
        else:
            results = trainer.validate(test_loader, opts)
            logger.info("*** End of Test")
            logger.info(results)
    logger.close()
    return results


if __name__ == "__main__":
    parser = argparser.get_argparser()
    main(argparser.modify_command_options(parser.parse_args()))
