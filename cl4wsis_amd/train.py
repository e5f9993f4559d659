"""Trainer with the reference's surface (train.py:38-812) whose per-batch body is `Trainer.step(batch)`.

The reference has no `step` method: the body of the `for` loop of `Trainer.train` (train.py:241-566) is factored out
here unchanged in meaning, and `train()` stays its caller. Phase 2 (the metric path) runs entirely on the HIP kernels
of libcl4wsis_hip.so: frozen old-model forward, flip-TTA semantic forward, trainable instance branch forward /
hand-written backward, label stage on device, fused losses, RCCL-averaged gradients, fused Adam.
"""
import os
from functools import reduce

import numpy as np
import torch
import torch.distributed as dist

from . import ops, tasks
from .optim import FusedAdam, FusedSGD
from .parallel import GradBuckets, world_size
from .scheduler import get_scheduler
from .segmentation_module import get_norm, make_model
from .wss import PeakGenerator, PseudoLabeler


def gaussian(sigma=6):
    """2-D gaussian splat kernel, float64 like the reference (modules/utils.py:49-59); cast to f32 on upload, which is
    the rounding np.maximum(f32 map, f64 g) -> f32 map performs."""
    size = 6 * sigma + 3
    x = np.arange(0, size, 1, float)
    y = x[:, np.newaxis]
    c = 3 * sigma + 1
    return np.exp(-((x - c) ** 2 + (y - c) ** 2) / (2 * sigma ** 2))


class _NoScaler:
    """The reference wraps the step in amp.autocast + GradScaler (train.py:43,252,537-541). This build computes in
    fp32 end to end (north star: 1e-3 relative to fp32), where loss scaling is an exact no-op; the object only keeps
    the checkpoint contract (`scaler` entry of run.py:26)."""

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass


def _clone_tree(v):
    """Static twins of a nest of device tensors (dicts / tuples / lists): contiguous clones, planes twins dropped."""
    if isinstance(v, dict):
        return {k: _clone_tree(x) for k, x in v.items()}
    if isinstance(v, (tuple, list)):
        return type(v)(_clone_tree(x) for x in v)
    return v.detach().clone().contiguous()


def _copy_tree(dst, src):
    if isinstance(dst, dict):
        for k in dst:
            _copy_tree(dst[k], src[k])
    elif isinstance(dst, (tuple, list)):
        for d, x in zip(dst, src):
            _copy_tree(d, x)
    else:
        dst.copy_(src)


class _SegmentedGraph:
    """A captured iteration cut at every point where the ranks exchange data: [graph, exchange, graph, exchange, ..., graph]. Collectives
    are never captured (RCCL calls stay out of every capture, like the gradient buckets of phase 2; a gloo rehearsal could not be captured
    at all): `cut(fn)` ends the current segment, remembers `fn` and begins the next segment in the same memory pool; `replay()` launches
    the segments in order and runs each `fn` eagerly between them, on the replaying stream. Used by phase 1 at N > 1, where every
    InPlaceABNSync layer exchanges its batch statistics once per direction (ops.STAT_EXCHANGE) — r04 ran that configuration eagerly."""

    def __init__(self, pool):
        self.pool = pool
        self.items = []
        self._g = None

    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
        self._g.capture_begin(pool=self.pool)

    def _end(self):
        self._g.capture_end()
        self.items.append(self._g)
        self._g = None

    def cut(self, fn):
        self._end()
        self.items.append(fn)
        self._begin()

    def capture(self, body):
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._begin()
            try:
                out = body(self)
            finally:
                self._end()
        cur.wait_stream(side)
        return out

    @property
    def n_graphs(self):
        return sum(isinstance(it, torch.cuda.CUDAGraph) for it in self.items)

    def replay(self):
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()


class Trainer:
    def __init__(self, logger, device, opts):
        self.logger = logger
        self.device = device
        self.opts = opts
        self.scaler = _NoScaler()
        self.classes = classes = tasks.get_per_task_classes(opts.dataset, opts.task, opts.step)
        new_classes = classes[-1]
        self.tot_classes = reduce(lambda a, b: a + b, classes)
        self.old_classes = self.tot_classes - new_classes

        self.model = make_model(opts, classes=classes)
        if opts.step == 0:
            self.model_old = None
        else:
            self.model_old = make_model(opts, classes=tasks.get_per_task_classes(opts.dataset, opts.task, opts.step - 1))
            self.model_old.to(self.device)
            for par in self.model_old.parameters():
                par.requires_grad = False
            self.model_old.eval()

        self.weakly = opts.weakly and opts.step > 0
        self.pseudolabeler = None
        self.peakgenerator = None
        self.peak_conf_thresh = opts.pseudo_thresh
        self.sigma = opts.sigma
        self.g = gaussian(self.sigma)
        if self.weakly and opts.pseudo is None:
            norm = get_norm(opts)
            channels = 4096 if "wide" in opts.backbone else 2048
            self.pseudolabeler = PseudoLabeler(channels, self.tot_classes, norm)
            if opts.peak_from == "peakgenerator":
                self.peakgenerator = PeakGenerator(in_channel=self.tot_classes, num_classes=self.tot_classes - 1,
                                                   old_classes=self.old_classes - 1)
        self.use_aff = opts.affinity                      # train.py:68
        self.pseudo_epoch = opts.pseudo_ep                # train.py:70
        self.lde = opts.loss_de                           # train.py:120-121
        self.center_loss_weight = 200
        self.offset_loss_weight = 0.01
        self.model.to(self.device)          # parameters must be resident before the optimizer aliases them
        if self.pseudolabeler is not None:
            self.pseudolabeler.to(self.device)
        if self.peakgenerator is not None:
            self.peakgenerator.to(self.device)
        self._g_dev = torch.from_numpy(self.g.astype(np.float32)).to(self.device)
        self.optimizer, self.scheduler = self.get_optimizer(opts)
        self.distribute(opts)
        self._iter = 0
        self._side_stream = (torch.cuda.Stream(device=self.device, priority=int(os.environ.get("CL4_SIDE_PRIORITY", "0")))
                             if os.environ.get("CL4_OVERLAP", "1") != "0" else None)
        self._use_graph = os.environ.get("CL4_GRAPH", "1") != "0"
        self._s0_static = None
        # N > 1: the ASPP tail as a chain of per-bucket graphs (_distributed_tail). Built and tested (world-size-1 RCCL rehearsal, two gloo ranks),
        # measured at world size 1: 34.57 / 19.53 / 11.95 / 8.27 ms per step at 16 / 8 / 4 / 2 images against 34.57 / 19.44 / 11.96 / 8.21 with the
        # eager tail (profiles/r04_solo_collectives.txt) — the host runs ahead of the device either way, so it is OFF by default: the only real
        # N > 1 RCCL run is the driver's, and a capture of ~15 kernels buys nothing there.
        self._tail_graph = os.environ.get("CL4_TAIL_GRAPH", "0") != "0"
        self._tail_segs, self._tail_keep = {}, []
        self._p1_static = None
        # phase 1 at N > 1: graph segments cut at the cross-rank exchanges (_SegmentedGraph); CL4_P1_SEG_GRAPH=0 keeps that configuration eager
        self._p1_seg_graph = os.environ.get("CL4_P1_SEG_GRAPH", "1") != "0"
        self._p1_exchanges = None
        self._graph, self._static, self._graph_active = None, None, False
        # Software pipelining across iterations (single-GPU graph mode): the frozen old model's forward of batch t+1 is replayed on
        # its own stream while batch t's step runs (it depends on nothing the optimizer touches). CL4_PIPELINE=0 switches it off.
        self._pipeline = os.environ.get("CL4_PIPELINE", "1") != "0" and self._side_stream is not None   # the fork needs a second stream
        self._pipe = None
        # per-GPU batches below this stay on the one-graph path: measured +4.8 % at b = 16, +1-2 % at 8 / 4; at 2 it was -1.8 % in r02
        # and is +2.2 % with the r03 kernels (8.50 -> 8.31 ms)
        self._pipe_min_b = int(os.environ.get("CL4_PIPELINE_MIN_BATCH", "2"))
        # where the next batch's old-model forward forks off: after the new model's body ("body") or head ("seg"). Measured (images/s):
        # body 362-364, head 361-362, after mod4 361-363, after mod3 349 = no pipelining: what pays is the overlap with the trainable
        # branch / backward, not with the new model's own convolutions.
        self._pipe_fork = os.environ.get("CL4_PIPE_FORK", "body")
        # r04 — deep pipelining: ALL frozen work of batch t+1 (old model on one side stream; new model's semantic forward, PseudoLabeler, CAM,
        # peaks and the semantic pseudo ground truth on a second one) runs beside batch t's trainable branch / label stage / backward / Adam, and
        # hands its results over in per-parity static buffers. At 16 images per GPU the chip is full either way (the shallow form above is what
        # pays there); at the 8 / 4 / 2 images per GPU of N = 2 / 4 / 8 the step is a chain of launches that fill a tenth of the chip, and
        # three chains side by side shorten it to the longest one. CL4_DEEP_PIPE=0: shallow pipelining at every batch size.
        self._deep_pipe = os.environ.get("CL4_DEEP_PIPE", "1") != "0" and self._pipeline
        self._deep_max_b = int(os.environ.get("CL4_DEEP_PIPE_MAX_BATCH", "8"))
        self._side2 = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("CL4_SIDE2_PRIORITY", "0"))) if self._deep_pipe else None
        self._label_override = None
        self._label_acc = torch.zeros(5, dtype=torch.int32, device=self.device)   # sticky label-stage counters, see check_label_overflow
        # the launch advances the seed by 1 per step: the rank lives in bits 40+ so that no rank ever replays another rank's stream
        # of masks (with "+ 7919 * rank", rank r at step t drew rank 0's mask of step t + 7919 r: ADVICE r02)
        self._seed_dev = torch.tensor([(opts.random_seed * 1000003) % (1 << 40) + ((dist.get_rank() if dist.is_initialized() else 0) << 40)],
                                      dtype=torch.int64, device=self.device)
        # RandDrop's random old-class labels (phase 1, utils/wss_loss.py:237): their own device-resident counter, same rank layout
        self._p1_seed_dev = torch.tensor([(opts.random_seed * 7919) % (1 << 40) + ((dist.get_rank() if dist.is_initialized() else 0) << 40)],
                                         dtype=torch.int64, device=self.device)

    # ------------------------------------------------------------------------------------------ optimizer
    def get_optimizer(self, opts):
        """train.py:144-185 (same groups, same order, same learning rates)."""
        tr = lambda m: [p for p in m.parameters() if p.requires_grad]
        params = []
        if not opts.freeze:
            params.append({"params": tr(self.model.body), "weight_decay": opts.weight_decay})
        if opts.phase is None or opts.phase == 2:
            if self.model.decoder is not None and hasattr(self.model.decoder, "instance_decoder"):
                params.append({"params": tr(self.model.decoder.instance_decoder), "weight_decay": opts.weight_decay,
                               "lr": opts.lr * opts.lr_head})
            if self.model.instance_head is not None:
                params.append({"params": tr(self.model.instance_head), "weight_decay": opts.weight_decay,
                               "lr": opts.lr * opts.lr_head})
        if not opts.freeze_seg:
            params.append({"params": tr(self.model.head), "weight_decay": opts.weight_decay, "lr": opts.lr * opts.lr_head})
            params.append({"params": tr(self.model.cls), "weight_decay": opts.weight_decay, "lr": opts.lr * opts.lr_head})
            if self.weakly and opts.pseudo is None:
                params.append({"params": tr(self.pseudolabeler), "weight_decay": opts.weight_decay, "lr": opts.lr_pseudo})
                if self.peakgenerator is not None:
                    params.append({"params": tr(self.peakgenerator), "weight_decay": opts.weight_decay, "lr": opts.lr_pseudo})
        params = [g for g in params if g["params"]]
        if opts.optim == "sgd":       # train.py:179-180 (phase 1, scripts/run.sh:33)
            optimizer = FusedSGD(params, lr=opts.lr, momentum=0.9, nesterov=True)
        elif opts.optim == "adam":
            optimizer = FusedAdam(params, lr=opts.lr, weight_decay=opts.weight_decay)
        else:
            raise NotImplementedError(f"--optim {opts.optim}")
        if not hasattr(opts, "max_iters"):
            opts.max_iters = getattr(opts, "epochs", 1) * 1000
        if not hasattr(opts, "start_decay"):
            opts.start_decay = opts.max_iters
        return optimizer, get_scheduler(opts, optimizer)

    def distribute(self, opts):
        """Replaces the DDP wrappers of train.py:187-196: gradients of the flat buffer are averaged over RCCL."""
        # 8 MB buckets: the four ASPP convolutions (8.4 / 18.9 / 18.9 / 18.9 MB of the 63 MB) each close their own bucket, in
        # the order the backward finishes them (InstanceBranch._decoder_backward_aspp)
        self.buckets = GradBuckets(self.optimizer.grads_fb, bucket_bytes=8 << 20)
        self._branch = self.model.instance_branch()
        self._p1 = None
        self._cur_epoch = 0
        self._grad_map = {id(p): self.optimizer.grad_view(p) for g in self.optimizer.param_groups for p in g["params"]}
        self._bn_counters = [b for n, b in self.model.named_buffers()
                             if n.endswith("num_batches_tracked") and n.startswith(("decoder.", "instance_head."))]

    # ------------------------------------------------------------------------------------------ the hot path
    def _set_phase2_modes(self):
        self.model.train()
        if self.opts.freeze:
            self.model.body.eval()
        if self.opts.freeze_seg:
            self.model.head.eval()
            self.model.cls.eval()

    def step(self, batch, dropout_keep=None, next_batch=None):
        """One training iteration = the body of train.py:241-566 for phase 2. `batch` is the loader tuple
        (images, seg_maps, center_maps, offset_maps, weights, l1h) or just (images, l1h): the weakly-supervised branch
        only consumes images and l1h (the other four are uploaded but unused by the reference, train.py:243-248).
        `next_batch` (optional, phase 2): the batch the NEXT call will be given — its frozen old-model forward (train.py:253-255)
        is then enqueued now, on a separate stream, and overlaps this iteration; the results are those of the unpipelined step.
        Returns a dict of device scalars (no host synchronisation)."""
        opts = self.opts
        if self.weakly and opts.phase == 1:
            return self._step_phase1(batch)
        if not self.weakly and opts.step == 0 and opts.phase is None:
            return self._step0_supervised(batch, dropout_keep)
        if not (self.weakly and opts.phase == 2):
            raise NotImplementedError("Trainer.step: built for step 0 (supervised, --bce) and for the weakly-supervised phase-1 / "
                                      "phase-2 steps (BASELINE configs 1-3); other option combinations are not on the CL4WSIS path")
        images, l1h = (batch[0], batch[-1])
        P = self._pipe
        if P is not None and P["have_old"][P["cur"]] and P.get("announced") is batch:
            # the batch the previous call announced: it went up then, into this parity's static inputs (no second upload)
            images, l1h = P["x"][P["cur"]], P["l"][P["cur"]]
        else:
            images = images.to(self.device, dtype=torch.float32).contiguous()
            l1h = l1h.to(self.device, dtype=torch.float32).contiguous()
        self._set_phase2_modes()
        use_graph = (self._use_graph and dropout_keep is None and self._iter >= 1 and ops.TIMER is None
                     and (self._static is None or tuple(images.shape) == tuple(self._static["images"].shape))
                     and (self._pipe is None or tuple(images.shape) == tuple(self._pipe["x"][0].shape)))
        if not use_graph:
            if self._pipe is not None:
                self._pipe["have_old"] = [False, False]      # an eager step consumes its batch itself: drop any announced prefetch
            if self._graph_active:
                from .nn import CACHE
                CACHE.clear()            # packed weights cached by the captured run are one step stale for eager code
                self._graph_active = False
            # first iteration: time the launch plans of every convolution shape of this step once (ops._autotune_conv)
            ops.AUTOTUNE = self._iter == 0 and os.environ.get("CL4_AUTOTUNE", "1") != "0"
            try:
                out = self._forward_backward(images, l1h, dropout_keep)
            finally:
                ops.AUTOTUNE = False
            self.buckets.finish()
            self.optimizer.step()
        elif (self._pipeline and images.shape[0] >= self._pipe_min_b and
              (next_batch is not None or (self._pipe is not None and self._pipe["have_old"][self._pipe["cur"]]))):
            out = self._replay_pipelined(images, l1h, next_batch, batch)
        else:
            out = self._replay(images, l1h)
        if self.scheduler is not None:
            self.scheduler.step()
        self._iter += 1
        return out

    def _step0_supervised(self, batch, dropout_keep=None):
        """One supervised step-0 iteration (train.py:241-300, BASELINE config 1) — see cl4wsis_amd/step0.py."""
        from .step0 import Step0
        if not self.opts.bce:
            raise NotImplementedError("step 0 is built for --bce (scripts/run.sh:24); plain BCE / DeepLabCE are not on the path")
        if getattr(self, "_s0", None) is None:
            self._s0 = Step0(self)
        images, seg_maps, center_maps, offset_maps, weights = batch[:5]
        f32 = lambda t: t.to(self.device, dtype=torch.float32).contiguous()
        self.model.train()
        tensors = (f32(images), seg_maps.to(self.device, dtype=torch.long).contiguous(), f32(center_maps), f32(offset_maps), f32(weights))
        # r05: on one GPU the step is HIP-graph replayed from its third call on, like phase 1 (the first two run eagerly: launch plans are
        # timed in iteration 0, the one-launch weight packs get their job table in iteration 1). The Dropout mask then comes from the
        # trainer's device-resident seed, which every launch advances. Injected masks (the parity tests), N > 1, the WiderResNet body
        # (Dropout2d seeds drawn on the host) and CL4_GRAPH=0 take the eager path.
        use_graph = (self._use_graph and dropout_keep is None and self._iter >= 2 and ops.TIMER is None and not self.buckets.distributed
                     and not hasattr(self.model.body, "bn_out") and os.environ.get("CL4_S0_GRAPH", "1") != "0"
                     and (self._s0_static is None or all(tuple(a.shape) == tuple(b.shape) for a, b in zip(tensors, self._s0_static["in"]))))
        if use_graph:
            losses = self._replay_step0(tensors)
        else:
            ops.AUTOTUNE = self._iter == 0 and os.environ.get("CL4_AUTOTUNE", "1") != "0"
            try:
                losses = self._s0.run(*tensors, dropout_keep)
            finally:
                ops.AUTOTUNE = False
            for i in range(len(self.buckets.fb.sizes)):
                self.buckets.ready(i)
            self.buckets.finish()
            self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        self._iter += 1
        return {"losses": losses}

    def _replay_step0(self, tensors):
        from .nn import CACHE
        opt = self.optimizer
        if self._s0_static is None:
            def stale_trainable():       # the graph re-derives (packs / folds) what it reads from the trainable weights itself (as _replay_phase1)
                for grp in opt.param_groups:
                    for prm in grp["params"]:
                        CACHE.touch(prm)
                for buf in self.model.buffers():
                    CACHE.touch(buf)
            S = {"in": [t.clone() for t in tensors], "pool": torch.cuda.graph_pool_handle()}
            opt.upload_lr()
            torch.cuda.synchronize()
            stale_trainable()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=S["pool"]):
                losses = self._s0.run(*S["in"], None)
                opt.step(upload=False)
            if hasattr(opt, "_step"):
                opt._step -= 1                            # capture only recorded the kernels
            stale_trainable()
            S["g"], S["losses"], S["keep"] = g, losses, self._derived_snapshot()
            self._s0_static = S
        S = self._s0_static
        for dst, src in zip(S["in"], tensors):
            dst.copy_(src)
        opt.upload_lr()
        S["g"].replay()
        self._graph_active = True
        opt.note_replayed_step()
        return S["losses"]

    def _step_phase1(self, batch, cur_epoch=None, flac_angle=None, randrop_neg=None):
        """One phase-1 iteration (train.py:241-566 with --phase 1) — see cl4wsis_amd/phase1.py.

        r04: on one GPU the step is HIP-graph replayed from its third call on. What varies between iterations is
        device-resident (learning rates, the RandDrop seed) or selects the graph: the FLAC angle (train.py:316 draws 90 / 180 / 270 on
        the host — three graphs, captured on first use, sharing one memory pool) and the side of `pseudo_ep` the epoch is on (two loss
        stacks). Injected angles / draws (the parity tests), N > 1 (cross-rank ABN statistics and the bucketed all-reduce sit between
        the kernels) and CL4_GRAPH=0 take the eager path."""
        from .phase1 import Phase1Step
        if self._p1 is None:
            self._p1 = Phase1Step(self)
        images, l1h = (batch[0], batch[-1])
        images = images.to(self.device, dtype=torch.float32).contiguous()
        l1h = l1h.to(self.device, dtype=torch.float32).contiguous()
        self.model.train()
        epoch = self._cur_epoch if cur_epoch is None else cur_epoch
        late = epoch >= self.pseudo_epoch
        use_graph = (self._use_graph and flac_angle is None and randrop_neg is None and self._iter >= 2 and ops.TIMER is None
                     and (not self.buckets.distributed or self._p1_seg_graph)
                     and (self._p1_static is None or tuple(images.shape) == tuple(self._p1_static["images"].shape)))
        if use_graph:
            angle = [90, 180, 270][np.random.randint(3)]          # train.py:316, drawn on the host like the reference does
            losses = self._replay_phase1(images, l1h, angle, late)
        else:
            ops.AUTOTUNE = self._iter == 0 and os.environ.get("CL4_AUTOTUNE", "1") != "0"
            try:
                losses = self._p1.run(images, l1h, epoch, flac_angle, randrop_neg)
            finally:
                ops.AUTOTUNE = False
            for i in range(len(self.buckets.fb.sizes)):
                self.buckets.ready(i)
            self.buckets.finish()
            self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        self._iter += 1
        return {"losses": losses}

    def _replay_phase1(self, images, l1h, angle, late):
        from .nn import CACHE
        opt = self.optimizer
        if self._p1_static is None:
            self._p1_static = {"images": images.clone(), "l1h": l1h.clone(), "graphs": {}, "pool": torch.cuda.graph_pool_handle(), "keep": []}
        S = self._p1_static
        key = (angle, late)
        if key not in S["graphs"]:
            def stale_trainable():
                # every graph re-derives (packs / folds) what it reads from the TRAINABLE weights itself; eager code and the other
                # graphs must not read a derived tensor that only this graph refreshes (same rule as _replay_pipelined). In phase 1
                # that includes the running statistics: the PseudoLabeler's eval pass (train.py:304-311) folds the statistics its own
                # training pass updated in the previous iteration.
                for grp in opt.param_groups:
                    for prm in grp["params"]:
                        CACHE.touch(prm)
                for mod in (self.model, self.pseudolabeler, self.peakgenerator):
                    if mod is not None:
                        for buf in mod.buffers():
                            CACHE.touch(buf)
            opt.upload_lr()
            torch.cuda.synchronize()
            stale_trainable()
            if not self.buckets.distributed:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=S["pool"]):
                    losses = self._p1.run(S["images"], S["l1h"], self.pseudo_epoch if late else 0, angle, None)
                    opt.step(upload=False)
            else:
                # N > 1 (r05): the same iteration as a chain of graph segments cut at every cross-rank exchange — the
                # packed float64 statistics of each InPlaceABNSync layer and direction (ops.STAT_EXCHANGE -> _SegmentedGraph.cut), then the
                # bucketed gradient average — with the update kernels in the last segment
                g = _SegmentedGraph(S["pool"])

                def body(seg):
                    ops.STAT_EXCHANGE.hook = seg.cut
                    try:
                        out = self._p1.run(S["images"], S["l1h"], self.pseudo_epoch if late else 0, angle, None)
                    finally:
                        ops.STAT_EXCHANGE.hook = None

                    def grad_average():
                        for i in range(len(self.buckets.fb.sizes)):
                            self.buckets.ready(i)
                        self.buckets.finish()
                    seg.cut(grad_average)
                    opt.step(upload=False)
                    return out

                n0 = ops.STAT_EXCHANGE.count
                losses = g.capture(body)
                self._p1_exchanges = {"abn_statistics": ops.STAT_EXCHANGE.count - n0, "gradient_buckets": len(self.buckets.bounds),
                                      "graph_segments": g.n_graphs}
            if hasattr(opt, "_step"):
                opt._step -= 1                            # capture only recorded the kernels (as _replay does)
            stale_trainable()
            S["graphs"][key] = (g, losses)
            S["keep"].append(self._derived_snapshot())
        g, losses = S["graphs"][key]
        S["images"].copy_(images)
        S["l1h"].copy_(l1h)
        opt.upload_lr()
        g.replay()
        self._graph_active = True
        opt.note_replayed_step()                          # host step count == the device's step_dev (state_dict / resume)
        return losses

    # HIP-graph path: iteration 0 runs eagerly (autotuning), iteration 1 captures the whole step — both streams, ~600
    # kernel launches — into one graph, and every later iteration is a copy of the batch into static buffers plus one
    # graph launch. Everything that changes between iterations lives in device memory (Dropout seed, Adam step count,
    # learning rates), so the captured kernel arguments stay valid.
    def _replay(self, images, l1h):
        opt = self.optimizer
        if self._static is None:
            self._static = {"images": images.clone(), "l1h": l1h.clone()}
            opt.upload_lr()
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            self._static["pool"] = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(self._graph, pool=self._static["pool"]):
                out = self._forward_backward(self._static["images"], self._static["l1h"], None,
                                             split_tail=self.buckets.distributed)
                if not self.buckets.distributed:
                    opt.step(upload=False)
            self._static["out"] = out
            self._static["keep"] = self._derived_snapshot()
            opt._step -= 1 if not self.buckets.distributed else 0   # capture only recorded the kernels
        self._static["images"].copy_(images)
        self._static["l1h"].copy_(l1h)
        opt.upload_lr()
        self._graph.replay()
        self._graph_active = True
        if not self.buckets.distributed:
            opt.note_replayed_step()
        else:
            self._distributed_tail(self._static["out"], self._static.get("pool"))
        return self._static["out"]

    def _derived_snapshot(self):
        # The captured kernels read the frozen networks' packed weights / folded BN at fixed addresses; those tensors were cached by
        # the eager first iteration and would be freed (and their memory reused) by the first eager rebuild after a CACHE.clear()
        # — validation between epochs, an eager fallback step. Holding them for the life of the graph keeps the addresses valid.
        # The scratch buffers the captured kernels address (split-K slabs, BN / loss / label workspaces) ride along: ops._Scratch
        # never frees a buffer it has handed out, this list is the graph owner's own reference to the same memory.
        from .nn import CACHE
        return (CACHE.snapshot([self.model, self.model_old, getattr(self, "pseudolabeler", None), getattr(self, "peakgenerator", None)])
                + ops.workspace_snapshot())

    def _distributed_tail(self, out, pool=None):
        # N > 1: the main graph ends where the ASPP block's backward begins. Those last kernels produce 90 % of the gradient bytes, and
        # each bucket's RCCL all-reduce has to be issued on the communication stream as soon as its last weight-gradient kernel is
        # enqueued, so that it overlaps the remaining ones; the RCCL calls stay out of every capture. r02 / r03 ran these ~15 launches
        # eagerly (0.3-0.6 ms of Python + launch time per step at world size 1). r04: the tail is captured ONCE as a
        # chain of small graphs cut exactly where a bucket completes (the ready-hook of the backward pass ends one capture and begins the
        # next); a step replays segment, marks its parameters ready (which launches the bucket's all-reduce behind it), replays the next.
        # CL4_TAIL_GRAPH=1 selects it; the default is the eager tail (no measurable difference at world size 1, see __init__).
        b = self.buckets
        aspp = {id(p) for p in self._branch.aspp_parameters()}
        if not self._tail_graph:
            tape, dy = out["_tail"]
            for p in self.optimizer.flat_params():
                if id(p) not in aspp:
                    b.ready(self.optimizer.index[id(p)])       # finalised inside the main graph
            self._branch.backward_aspp(tape, dy, self._grad_map, on_ready=self._ready)
            b.finish()
            self.optimizer.step()
            return
        key = id(out)
        if key not in self._tail_segs:
            self._tail_segs[key] = self._capture_tail(out, aspp, pool)
        pre, segs = self._tail_segs[key]
        for i in pre:
            b.ready(i)
        for g, ready in segs:
            g.replay()
            for i in ready:
                b.ready(i)
        b.finish()
        self.optimizer.step()

    def _capture_tail(self, out, aspp, pool):
        """Capture `backward_aspp` as a chain of graphs, one per gradient bucket it completes. Returns (indices of the parameters the
        main graph already finalised, [(graph, indices that become final with it), ...])."""
        tape, dy = out["_tail"]
        b, opt = self.buckets, self.optimizer
        pre = [opt.index[id(p)] for p in opt.flat_params() if id(p) not in aspp]
        pending = [hi - lo for lo, hi in b.bounds]
        for i in pre:
            pending[b.bucket_of[i]] -= 1
        segs, cur, state = [], [], {"g": None}
        kw = {} if pool is None else {"pool": pool}

        def begin():
            state["g"] = torch.cuda.CUDAGraph()
            state["g"].capture_begin(**kw)

        def cut():
            state["g"].capture_end()
            segs.append((state["g"], list(cur)))
            del cur[:]

        def hook(param):
            i = opt.index[id(param)]
            cur.append(i)
            bk = b.bucket_of[i]
            pending[bk] -= 1
            if pending[bk] == 0:              # this bucket is complete: its all-reduce starts behind the kernels captured so far
                cut()
                begin()

        opt.upload_lr()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            begin()
            self._branch.backward_aspp(tape, dy, self._grad_map, on_ready=hook)
            cut()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._tail_keep.append(self._derived_snapshot())
        return pre, segs

    # Pipelined graph path (single GPU, used when the caller announces the next batch). Two parities p = iteration % 2 with their
    # own static inputs X[p], L[p] and old-model outputs OLD[p]. Graph G[p] = the step on X[p] reading OLD[p], PLUS, forked off
    # after the new model's forward, the frozen old model's forward of X[1-p] into OLD[1-p] (joined at the end). The old model
    # of the next batch thus runs beside this step's trainable branch / label stage / backward / Adam — mostly memory-bound
    # kernels — instead of beside the new model's forward, which is the same kind of work. (Two separate graphs on two streams do
    # not overlap on this ROCm: their replays serialise — measured 49.2 vs 45.7 ms — so the fork has to live inside one graph.)
    # Also used at N > 1 (graph up to the ASPP backward). Deep mode (r04, P["deep"]: up to `_deep_max_b` images per GPU): G[p] is the
    # TRAINABLE stage of X[p] on the handed-over frozen results FZ[p] / OLD[p], beside the whole frozen stage of X[1-p] on two side
    # streams (`_deep_step`); "have_old[p]" then stands for both OLD[p] and FZ[p].
    def _replay_pipelined(self, images, l1h, next_batch, batch=None):
        opt = self.optimizer
        if self._pipe is None:
            opt.upload_lr()
            with torch.no_grad():
                o, _ = self.model_old(images, interpolate=False)         # shapes of the two maps the step reads from the old model
            P = {"x": [images.clone(), images.clone()], "l": [l1h.clone(), l1h.clone()],
                 "old": [{k: torch.empty_like(o[k]) for k in ("center", "offset")} for _ in range(2)],
                 "g": [], "out": [], "have_old": [False, False], "cur": 0,
                 "deep": self._deep_pipe and images.shape[0] <= self._deep_max_b}
            del o
            if P["deep"]:                                # per-parity static copies of everything the frozen new model hands to the step
                cur = torch.cuda.current_stream()        # (on the second side stream, so that its scratch buffers exist before the capture)
                self._side2.wait_stream(cur)
                with torch.no_grad(), torch.cuda.stream(self._side2):
                    fz = self._frozen_new(images, l1h)
                    P["fz"] = [_clone_tree(fz), _clone_tree(fz)]
                    del fz
                cur.wait_stream(self._side2)
            torch.cuda.synchronize()
            pool = torch.cuda.graph_pool_handle()
            P["pool"] = pool
            from .nn import CACHE

            def stale_trainable():
                # every graph must re-derive (pack / fold) what it reads from the TRAINABLE weights itself — a derived tensor cached by
                # the other parity's capture would only be refreshed when that other graph replays. The frozen networks' derived
                # tensors (made by the eager first iteration, outside any graph pool) stay valid and are not rebuilt per replay.
                for grp in opt.param_groups:
                    for prm in grp["params"]:
                        CACHE.touch(prm)

            # opt-in (see _deep_step): the frozen stage of the next batch as a graph of its own, replayed on the side stream (own memory
            # pool: the two graphs of a step run concurrently)
            split = P["deep"] and self.buckets.distributed and os.environ.get("CL4_SPLIT_SIDE_GRAPH", "0") != "0"
            if split:
                P["gB"], P["pool_side"] = [], torch.cuda.graph_pool_handle()
            for q in range(2):
                stale_trainable()
                if split:
                    gb = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gb, pool=P["pool_side"]):
                        self._deep_step(P, q, part="frozen")
                    P["gB"].append(gb)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    if getattr(self, "_stamps", None) is not None:
                        self._stamps["prefix"] = f"g{q}."
                    if P["deep"]:
                        out = self._deep_step(P, q, part="trainable" if split else "all")
                    else:
                        out = self._forward_backward(P["x"][q], P["l"][q], None, split_tail=self.buckets.distributed,
                                                     old_outputs=P["old"][q], prefetch_old=(P["x"][1 - q], P["old"][1 - q]))
                    if not self.buckets.distributed:
                        opt.step(upload=False)
                    self._stamp("main.end")
                if not self.buckets.distributed:
                    opt._step -= 1                        # capture only recorded the kernels
                P["g"].append(g)
                P["out"].append(out)
            stale_trainable()                             # nothing eager may read a derived tensor that only one of the graphs refreshes
            P["keep"] = self._derived_snapshot()
            self._pipe = P
        P = self._pipe
        p = P["cur"]
        if P["have_old"][p] and P.get("announced") is not batch:
            P["have_old"][p] = False                     # the caller came with another batch object than it announced: prefetch void
        if images is not P["x"][p]:                      # (the announced batch is already there: step() handed the static tensors back)
            P["x"][p].copy_(images)
            P["l"][p].copy_(l1h)
        if not P["have_old"][p]:                         # not announced by the previous call: the old model runs now, up front
            with torch.no_grad():
                o, _ = self.model_old(P["x"][p], interpolate=False)
                for k in ("center", "offset"):
                    P["old"][p][k].copy_(o[k])
                if P["deep"]:                            # ... and so does the rest of the frozen stage
                    _copy_tree(P["fz"][p], self._frozen_new(P["x"][p], P["l"][p]))
        # The graph always runs the old model on X[1-p]; its result counts only if the caller announced that batch and it has the
        # captured shape (a last, smaller batch of an epoch takes the eager path of step()). Nothing announced — the last step of an
        # epoch, a caller that stops looking ahead — leaves OLD[1-p] marked invalid, and the next call runs the old model itself.
        announced = False
        if next_batch is not None:
            nx, nl = next_batch[0], next_batch[-1]
            if tuple(nx.shape) == tuple(P["x"][1 - p].shape):
                P["x"][1 - p].copy_(nx.to(self.device, dtype=torch.float32))
                P["l"][1 - p].copy_(nl.to(self.device, dtype=torch.float32))
                announced = True
        P["announced"] = next_batch if announced else None       # identity of the promise (held, so that its id cannot be reused)
        opt.upload_lr()
        if P.get("gB"):
            main = torch.cuda.current_stream()
            self._side_stream.wait_stream(main)          # the batch copies above; the previous step's readers of FZ / OLD[1 - p]
            with torch.cuda.stream(self._side_stream):
                P["gB"][p].replay()
        P["g"][p].replay()
        P["have_old"][p] = False
        P["have_old"][1 - p] = announced
        P["cur"] = 1 - p
        self._graph_active = True
        if not self.buckets.distributed:
            opt.note_replayed_step()
        else:
            from .nn import CACHE
            for prm in self._branch.aspp_parameters():    # the eager ASPP tail derives its operands from the current weights itself
                CACHE.touch(prm)
            if "_tail" in P["out"][p]:
                self._distributed_tail(P["out"][p], P.get("pool"))
            else:                                        # the whole backward ran inside the graph: every bucket goes out now
                for i in range(len(self.buckets.fb.sizes)):
                    self.buckets.ready(i)
                self.buckets.finish()
                self.optimizer.step()
        if P.get("gB"):
            torch.cuda.current_stream().wait_stream(self._side_stream)      # the step ends when the next batch's frozen stage has
        return P["out"][p]

    def enable_chain_stamps(self, slots=64):
        """Measurement aid (tools/chain_gantt.py): from the next capture on, the chains of the pipelined step graphs carry time-stamp
        kernels (cl4_stamp, 10 ns device clock) at their heads, tails and stage boundaries. Call before the first graph step."""
        self._stamps = {"buf": torch.zeros(slots, dtype=torch.int64, device=self.device), "names": [], "prefix": ""}

    def _stamp(self, name):
        S = getattr(self, "_stamps", None)
        if S is None:
            return
        full = S["prefix"] + name
        if full not in S["names"]:
            if len(S["names"]) >= S["buf"].numel():
                return
            S["names"].append(full)
        i = S["names"].index(full)
        ops.call("cl4_stamp", S["buf"][i:i + 1].data_ptr(), ops.stream())

    def chain_stamps(self):
        """{name: device clock in microseconds} of the most recent replays (host synchronisation)."""
        S = getattr(self, "_stamps", None)
        if S is None:
            return {}
        torch.cuda.synchronize()
        v = S["buf"].cpu().tolist()
        return {n: v[i] * 0.01 for i, n in enumerate(S["names"])}

    def _deep_step(self, P, q, part="all"):
        """Body of the deep-pipelined graph of parity q: the trainable stage of batch X[q] on its handed-over frozen results FZ[q] / OLD[q]
        (main stream), beside the whole frozen stage of X[1 - q] into FZ[1 - q] / OLD[1 - q] (two side streams), joined at the end.
        `part` (r05, N > 1, opt-in: CL4_SPLIT_SIDE_GRAPH=1): "frozen" / "trainable" capture the two halves as SEPARATE graphs — the frozen
        one is replayed on the side stream, so that everything enqueued behind the trainable graph (the eager ASPP tail, the bucketed
        all-reduce, Adam) runs while the frozen chains are still busy; one graph is one stream-ordered unit, behind which the all-reduce
        of 63 MB (~2 ms over xGMI at N = 8 by the ring estimate) would be fully exposed. At world size 1, where the all-reduce costs
        nothing, two graph launches are 0.3-0.5 ms SLOWER than one (FINDINGS.md finding 60): hence off by default, for a node to try."""
        main = torch.cuda.current_stream()
        r = 1 - q
        H, W = P["x"][q].shape[2:]
        if getattr(self, "_stamps", None) is not None:
            self._stamps["prefix"] = f"g{q}."
        if part == "trainable":
            self._stamp("main.head")
            self.optimizer.zero_grad()
            out = self._trainable_stage(P["fz"][q], P["old"][q], P["l"][q], None, (H, W), split_tail=self.buckets.distributed)
            self._stamp("main.trainable_tail")
            return out
        if part == "all":
            self._stamp("main.head")
        self._side_stream.wait_stream(main)
        self._side2.wait_stream(main)

        def old_chain():
            with torch.cuda.stream(self._side_stream):
                self._stamp("old_model.head")
                o_next, _ = self.model_old(P["x"][r], interpolate=False)
                for k in ("center", "offset"):
                    P["old"][r][k].copy_(o_next[k])
                del o_next
                self._stamp("old_model.tail")

        def frozen_chain():
            with torch.cuda.stream(self._side2):
                self._stamp("frozen_new.head")
                _copy_tree(P["fz"][r], self._frozen_new(P["x"][r], P["l"][r]))
                self._stamp("frozen_new.tail")

        # r05: the in-graph time stamps (tools/chain_gantt.py) show that the chain captured SECOND starts ~0.5 ms after the first — the
        # replay enqueues the graph's ~570 nodes in capture order — so the longest chain (the new model's frozen stage: it ends last) is
        # captured first; CL4_DEEP_ORDER=old restores the r04 order
        order = (old_chain, frozen_chain) if os.environ.get("CL4_DEEP_ORDER", "frozen") == "old" else (frozen_chain, old_chain)
        for chain in order:
            chain()
        if part == "frozen":
            main.wait_stream(self._side_stream)
            main.wait_stream(self._side2)
            return None
        self.optimizer.zero_grad()

        def join_end():
            self._stamp("main.trainable_tail")
            main.wait_stream(self._side_stream)
            main.wait_stream(self._side2)
            self._stamp("main.joined")

        # r05, N > 1 at the smallest per-GPU batches: the ASPP backward stays INSIDE the graph (beside the frozen chains, like the one-process
        # path) and the gradient buckets go out behind the graph. The eager tail behind the graph could hide at most its own ~0.4 ms of the
        # all-reduce (63 MB: ~2 ms over xGMI at N = 8 by the ring estimate) and costs 0.45 ms of exposed kernels, because one graph is one
        # stream-ordered unit and the frozen chains end ~1 ms after the trainable stage (finding 49). Larger per-GPU batches (N = 2: a tail of
        # ~1 ms against an all-reduce of ~1.3 ms) keep the overlapped tail. CL4_TAIL_IN_GRAPH_MAX_BATCH [4]; 0 = always the eager tail.
        in_graph = self.buckets.distributed and P["x"][q].shape[0] <= int(os.environ.get("CL4_TAIL_IN_GRAPH_MAX_BATCH", "4"))
        return self._trainable_stage(P["fz"][q], P["old"][q], P["l"][q], None, (H, W), split_tail=self.buckets.distributed and not in_graph,
                                     join_end=join_end)

    def _frozen_new(self, images, l1h, fork=None):
        """The part of the step that only the FROZEN new model and the batch feed (train.py:259-263, 421-449): semantic forward on
        [images ; flipped images], PseudoLabeler / PeakGenerator in eval mode -> CAM -> peaks, semantic pseudo ground truth. No trainable
        weight is read, so a pipelined replay may run it for the NEXT batch (`_replay_pipelined`, deep mode). `fork` (shallow pipelining):
        called where the next batch's old-model forward starts on the side stream."""
        opts = self.opts
        bs, _, H, W = images.shape
        oc = self.old_classes
        x2 = torch.empty((2 * bs, 3, H, W), dtype=torch.float32, device=self.device)
        x2[:bs].copy_(images)
        ops.flip_w(images, out=x2[bs:])
        self.model.body.f32_images = bs if os.environ.get('CL4_F32_HINT', '1') != '0' else None         # low-level fp32 features are read for the un-flipped half only (feats below)
        try:
            if fork is not None and self._pipe_fork == "body":
                # fork between the new model's body and its DeepLabV3 head (forward_seg = body -> head -> cls, segmentation_module.py:400-426)
                feats_all = self.model.forward_seg(x2, as_feature_extractor=True)
                fork()
                outputs = {"seg": self.model.cls(self.model.head(feats_all["body"]))}
                features = feats_all
            else:
                outputs, features = self.model.forward_seg(x2, interpolate=False)
                if fork is not None:
                    fork()
        finally:
            self.model.body.f32_images = None         # the hint must not outlive this forward (validate / flip-TTA read every image)
        feats = {k: ops.batch_slice(v, bs) for k, v in features["features"].items()}   # first half = un-flipped images (contiguous views, planes twins kept)
        # train.py:421-436 — CAM of the new classes -> peaks
        self.pseudolabeler.eval()
        self.peakgenerator.eval()
        int_masks = self.pseudolabeler(ops.batch_slice(features["body"], bs))
        label_new = l1h[:, oc - 1:].contiguous()
        seg2b = outputs["seg"]
        if self._label_override is not None:    # benchmark-only: see set_label_override (every network above still ran)
            int_masks, seg2b = self._label_override["int_masks"], self._label_override["seg2b"]
        cam = self.peakgenerator.cam(int_masks, label_new)
        peaks = ops.peak_extract(cam, label_new, (H, W), kernel=15, K=25)
        # train.py:438-449 — semantic pseudo ground truth
        seg_gt, old_w, prob_win = ops.seg_pseudo_gt(seg2b, l1h, oc, (H, W), no_mask=opts.no_mask)
        return {"feats": feats, "peaks": peaks, "seg_gt": seg_gt, "old_w": old_w, "prob_win": prob_win}

    def _trainable_stage(self, fz, outputs_old, l1h, dropout_keep, size, split_tail=False, join_old=None, join_end=None):
        """Everything behind the frozen networks (train.py:265-272, 451-541): trainable instance branch on the detached features, label
        stage on its predictions, blended targets + losses, backward. `join_old` / `join_end`: stream joins (the old model's maps are
        consumed from the loss on; a prefetch of the next batch is joined before the step, and its graph, ends)."""
        opts = self.opts
        H, W = size
        oc = self.old_classes
        feats = fz["feats"]
        bs = l1h.shape[0]
        if dropout_keep is None:
            n5, _, h5, w5 = feats["res5"].shape
            drop = self.model.decoder.instance_decoder.aspp.project
            dropout_keep = ops.dropout_mask_dev((n5, drop[0].out_channels, h5, w5), drop[3].p, self._seed_dev,
                                                1, self.device)        # the kernel hashes the seed: +1 per step is enough
        pred, tape = self._branch.forward_train(feats, dropout_keep=dropout_keep)
        self._stamp("main.branch_forward_done")
        torch._foreach_add_(self._bn_counters, 1)
        label_new = l1h[:, oc - 1:].contiguous()
        lab_center, lab_offset = pred["center"], pred["offset"]
        if self._label_override is not None:
            lab_center, lab_offset = self._label_override["center_lo"], self._label_override["offset_lo"]
        seg_gt, old_w, prob_win, peaks = fz["seg_gt"], fz["old_w"], fz["prob_win"], fz["peaks"]
        # train.py:451-500 — pseudo labels + refined labels
        pseudo, refined = ops.label_generation(
            seg_gt, prob_win, label_new, peaks, lab_center, lab_offset, self._g_dev, oc, sigma=self.sigma,
            pseudo_thresh=opts.pseudo_thresh, refine_thresh=opts.refine_thresh, nms_kernel=opts.kernel, beta=opts.beta,
            run_refine=opts.run_refine)
        ops.label_counters_accumulate_(self._label_acc, bs, H, W)
        self._stamp("main.labels_done")
        if join_old is not None:
            join_old()                               # the old model's centre / offset maps are consumed from here on
        # train.py:503-522 — blended targets + four weighted losses (+ gradient w.r.t. the low-resolution maps)
        losses, dcen, doff = ops.p2_loss(pred["center"], pred["offset"], outputs_old["center"], outputs_old["offset"],
                                         old_w, pseudo, refined, (H, W), self.center_loss_weight, self.offset_loss_weight)
        # train.py:537 — backward of the trainable sub-net; gradient buckets are all-reduced as they complete
        out = {"losses": losses, "seg_gt": seg_gt, "peaks": peaks, "pseudo": pseudo, "refined": refined,
               "center": pred["center"], "offset": pred["offset"], "old_w": old_w}
        if split_tail:       # data-parallel graph capture: stop before the ASPP backward, see _replay
            dy = self._branch.backward(tape, {"center": dcen, "offset": doff}, grads=self._grad_map, stop_before_aspp=True)
            out["_tail"] = (tape, dy)
            if join_end is not None:
                join_end()
            return out
        self._branch.backward(tape, {"center": dcen, "offset": doff}, grads=self._grad_map,
                              on_ready=self._ready if (self.buckets.distributed and not torch.cuda.is_current_stream_capturing()) else None)
        if join_end is not None:
            join_end()                               # join the next batch's frozen work before the step (and its graph) ends
        return out

    def _forward_backward(self, images, l1h, dropout_keep, split_tail=False, old_outputs=None, prefetch_old=None):
        bs, _, H, W = images.shape

        # train.py:253-255 — frozen old model, full forward, low resolution outputs. It is independent of the new model's
        # forward below, so it runs on a side stream: the two kernel sequences interleave on the device and fill each
        # other's workgroup-quantisation tails (33x33 layers at b=16 are only ~1 workgroup per CU).
        main = torch.cuda.current_stream()
        if old_outputs is not None:
            outputs_old = old_outputs                    # pipelined replay: computed by its own graph (see _replay_pipelined)
        elif self._side_stream is not None:
            self._side_stream.wait_stream(main)
            with torch.cuda.stream(self._side_stream):
                outputs_old, _ = self.model_old(images, interpolate=False)
                for t in outputs_old.values():
                    t.record_stream(main)
        else:
            outputs_old, _ = self.model_old(images, interpolate=False)
        self.optimizer.zero_grad()

        def fork_old():                          # pipelined replay: the NEXT batch's old-model forward starts here, on the side stream
            x_next, old_next = prefetch_old
            self._stamp("main.fork")
            self._side_stream.wait_stream(main)
            with torch.cuda.stream(self._side_stream):
                self._stamp("old_model.head")
                o_next, _ = self.model_old(x_next, interpolate=False)
                for k in ("center", "offset"):
                    old_next[k].copy_(o_next[k])
                self._stamp("old_model.tail")

        if prefetch_old is not None:
            self._stamp("main.head")
        fz = self._frozen_new(images, l1h, fork=fork_old if prefetch_old is not None else None)
        if prefetch_old is not None:
            self._stamp("main.frozen_new_done")

        def join():
            self._stamp("main.trainable_tail")
            main.wait_stream(self._side_stream)
            self._stamp("main.joined")
        return self._trainable_stage(fz, outputs_old, l1h, dropout_keep, (H, W), split_tail=split_tail,
                                     join_old=join if (self._side_stream is not None and old_outputs is None) else None,
                                     join_end=join if prefetch_old is not None else None)

    def set_label_override(self, inputs):
        """Benchmark hook (SURVEY.md §8(d) *structured-labels* mode): with random-init networks the label stage only ever sees
        noise (mostly sub-20-px components), so its connected-component / peak / refine kernels do unrepresentative work.
        `inputs` = {seg2b [2b,C,h,w], int_masks [b,C,h,w], center_lo [b,C-1,hc,wc], offset_lo [b,2,hc,wc]} (device tensors)
        replaces what the label stage READS; every network still runs and the losses still act on the real predictions.
        None restores the end-to-end path. The captured graph is dropped because the kernel arguments change."""
        if inputs is not None:
            inputs = {k: inputs[k].to(self.device, dtype=torch.float32).contiguous() for k in ("seg2b", "int_masks", "center_lo", "offset_lo")}
        self._label_override = inputs
        self._graph, self._static, self._pipe = None, None, None
        self._tail_segs, self._tail_keep = {}, []
        if self._graph_active:
            from .nn import CACHE
            CACHE.clear()
            self._graph_active = False
        self._label_acc.zero_()

    def comm_stats(self):
        """Exposed gradient all-reduce time per step etc. (None unless `self.buckets.timing` was switched on; N > 1 only)."""
        return self.buckets.comm_stats() if self.buckets.distributed else None

    def collectives_per_iteration(self):
        """What one phase-1 iteration at N > 1 exchanges across ranks (None before its graph exists): packed ABN statistics (one float64
        vector per layer and direction), gradient buckets, and the graph segments between them."""
        return self._p1_exchanges

    def label_counters(self):
        """Running maxima of the label stage's device-side table counts since construction + the sticky overflow flag
        (forces a host sync)."""
        return dict(zip(ops.LABEL_COUNTER_NAMES, self._label_acc.tolist()))

    def check_label_overflow(self):
        """The label kernels keep contours / centres / splats in fixed-capacity device tables (csrc/refine.hip CT_CAP,
        LIST_CAP) and set a flag instead of writing past them. Truncated labels would be silently wrong targets, so the
        flag is read wherever the loop synchronises with the host anyway and is fatal."""
        c = self.label_counters()
        if c["overflow"]:
            raise RuntimeError(f"label stage: a device table capacity was exceeded (counters {c}); labels of at least one "
                               "iteration were truncated")
        return c

    def _ready(self, param):
        self.buckets.ready(self.optimizer.index[id(param)])

    # ------------------------------------------------------------------------------------------ epoch loop
    def train(self, cur_epoch, train_loader, print_int=10):
        """train.py:198-586 — epoch loop around `step`; returns (epoch_loss, reg_loss) like the reference."""
        logger = self.logger
        logger.info("Epoch %d, lr = %f" % (cur_epoch, self.optimizer.param_groups[0]["lr"]))
        if hasattr(train_loader, "sampler") and hasattr(train_loader.sampler, "set_epoch"):
            train_loader.sampler.set_epoch(cur_epoch)
        total = torch.zeros((), dtype=torch.float32, device=self.device)
        interval = torch.zeros_like(total)
        n = 0
        self._cur_epoch = cur_epoch
        it = iter(train_loader)
        batch = next(it, None)
        cur_step = -1
        while batch is not None:
            cur_step += 1
            nxt = next(it, None)                 # one batch of lookahead: Trainer.step overlaps its frozen old-model forward
            out = self.step(batch, next_batch=nxt) if (self.weakly and self.opts.phase == 2) else self.step(batch)
            batch = nxt
            l = out["losses"]                   # phase 2: l_center + l_offset (the zeroed terms of train.py:524-529 vanish)
            loss = torch.stack([v.reshape(()) for v in l.values()]).sum() if isinstance(l, dict) else l.sum()
            total += loss
            interval += loss
            n += 1
            if (cur_step + 1) % print_int == 0:
                v = interval.item() / print_int  # the only host sync, every print_int iterations
                assert np.isfinite(v), f"l_reg: {v}"   # train.py:535
                if self.weakly and self.opts.phase == 2:
                    self.check_label_overflow()
                logger.debug(f"Epoch {cur_epoch}, Batch {cur_step + 1}/{len(train_loader)}, Loss={v}")
                if logger is not None and hasattr(logger, "add_scalar"):
                    logger.add_scalar("Loss/tot", v, cur_epoch * len(train_loader) + cur_step + 1, intermediate=True)
                interval.zero_()
        if self.weakly and self.opts.phase == 2:
            self.check_label_overflow()
        reg = torch.zeros_like(total)
        if dist.is_initialized():
            dist.reduce(total, dst=0)
            dist.reduce(reg, dst=0)
        epoch_loss = total / world_size() / max(n, 1)
        logger.info(f"Epoch {cur_epoch}, Class Loss={epoch_loss}, Reg Loss={reg}")
        return (epoch_loss, reg)

    def _leave_graph_mode(self):
        """Eager code after graph-replayed training must not trust the derived-tensor cache: during a replay the host-side
        `CACHE.touch` calls of the optimizer / BN kernels do not run, so entries built by an earlier eager pass (eval-mode
        packed weights `wp1`, folded BN scale/shift of the instance branch) would silently describe old weights. Every
        eval entry point drops them first."""
        from .nn import CACHE
        if self._graph_active:
            CACHE.clear()
            self._graph_active = False

    def validate(self, valid_loader, opts):
        """train.py:588-651 — instance-segmentation validation: eval forward (optionally [image, flipped image]), outputs to
        the ground-truth size, get_ins_map on the device (cl4_val_seg + cl4_instance_map), VOC mask AP at IoU 0.5:0.05:0.9
        on the host (cl4wsis_amd.metrics: restatement of the chainercv helpers the reference imports). Returns
        {'ap': per-class AP averaged over the thresholds, 'map': their mean} like the reference (on every rank)."""
        from . import metrics as MT
        iou_thresholds = np.arange(0.5, 0.95, 0.05).tolist()
        acc = [MT.DetectionAccumulator() for _ in iou_thresholds]
        model, device = self.model, self.device
        self._leave_graph_mode()
        model.eval()
        for img, seg, mask, mask_label, fname in valid_loader:
            gt_mask = np.asarray(mask)[0].astype(bool)
            gt_label = np.asarray(mask_label)[0] - 1                      # -1 removes the background class
            pred_label, pred_mask, pred_score = self.predict_instances(img, tuple(mask.shape[-2:]), opts)
            for a, thr in zip(acc, iou_thresholds):
                a.update(gt_label, gt_mask, pred_label, pred_mask, pred_score, thr)
        ap_masks = np.zeros((len(iou_thresholds), self.tot_classes - 1))
        for i, a in enumerate(acc):
            ap = a.ap()
            ap_masks[i, :len(ap)] = ap[:self.tot_classes - 1]
            ap_masks[i, len(ap):] = np.nan
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)      # classes absent from predictions and ground truth: nan
            ap_05_95 = np.nanmean(ap_masks, axis=0)
            result = {"ap": ap_05_95, "map": np.nanmean(ap_05_95)}
        model.train()
        return result

    def predict_instances(self, img, target_size, opts, cls_label=None):
        """One validation image -> (pred_label [R], pred_mask [R,H,W] bool, pred_score [R]) — train.py:616-626 +
        dataset/utils.py:795-892. `img` [1,3,h,w]; `target_size` the ground-truth (original image) size."""
        self._leave_graph_mode()
        images = img.to(self.device, dtype=torch.float32).contiguous()
        if images.shape[0] != 1:
            raise RuntimeError("validate: batch size 1 expected (train.py:608-618)")
        if opts.val_flip:
            images = torch.cat([images, ops.flip_w(images)], dim=0)
        out = self.model(images)[0]                                       # interpolate=True: align_corners=True to the input size
        up = {k: ops.resize_bilinear(v, tuple(target_size), False) for k, v in out.items()}
        lab = None
        if opts.val_clean and cls_label is not None:
            lab = cls_label.to(self.device, dtype=torch.float32).reshape(-1).contiguous()
        seg_map, prob = ops.val_seg(up["seg"], flip=opts.val_flip, cls_label=lab)
        center = up["center"]
        if opts.val_flip:
            center = ops.scale_(ops.add_act(center[:1].contiguous(), ops.flip_w(center[1:].contiguous()), ops.ACT_NONE, 0.0), 0.5)
        ins, table = ops.instance_map(seg_map, prob, center[0].contiguous(), up["offset"][0].contiguous(), thresh=opts.val_thresh,
                                      nms_kernel=opts.val_kernel, beta=opts.beta, ignore=opts.val_ignore)
        ins, table = ins.cpu().numpy(), table.cpu().numpy()
        keep = [i for i in range(len(table)) if table[i, 2] > 0]
        if not keep:                                                      # dataset/utils.py:882-885
            return np.zeros(1, dtype=np.int64), np.zeros((1,) + tuple(target_size), dtype=bool), np.zeros(1)
        return (np.array([int(table[i, 0]) for i in keep]), np.stack([ins == i for i in keep], 0),
                np.array([float(table[i, 1]) for i in keep]))

    def validate_semseg(self, loader, metrics, opts, eval_pseudolabeler=False):
        """train.py:695-745 — semantic validation: eval forward (optionally [image, flipped image]), logits to image size,
        flip average, softmax, arg-max, confusion matrix. The matrix is accumulated on the device (cl4_semseg_hist) and
        read back once; `metrics` is a cl4wsis_amd.metrics.StreamSegMetrics (same results dict as the reference's)."""
        metrics.reset()
        model, device = self.model, self.device
        self._leave_graph_mode()
        model.eval()
        if eval_pseudolabeler:
            self.pseudolabeler.eval()
        hist = metrics.device_hist(device)
        n = 0
        for batch in loader:
            img, seg = batch[0], batch[1]
            images = img.to(device, dtype=torch.float32).contiguous()
            labels = seg.to(device, dtype=torch.long).contiguous()
            if opts.val_flip:
                if images.shape[0] != 1:
                    raise RuntimeError("validate_semseg: --val_flip expects batch size 1 (train.py:709-711)")
                images = torch.cat([images, ops.flip_w(images)], dim=0)   # batch assembly only
            size = tuple(images.shape[-2:])
            if eval_pseudolabeler:
                lo = self.pseudolabeler(model(images, as_feature_extractor=True)["body"])
                masks = ops.resize_bilinear(lo, size, False)
            else:
                masks = model(images)[0]["seg"]       # interpolate=True: align_corners=True upsampling of the reference
            ops.semseg_hist_(hist, masks, labels, metrics.n_classes, flip=opts.val_flip,
                             zero_below=self.old_classes if eval_pseudolabeler else 0)
            n += labels.shape[0]
        metrics.flush_device(n)
        metrics.synch(device)
        return metrics.get_results()

    # ------------------------------------------------------------------------------------------ checkpoints
    @staticmethod
    def _strip(sd):
        return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}

    def load_step_ckpt(self, path):
        """train.py:747-771 — weights of the previous step into the new model (non strict) and the old model."""
        import os.path as osp
        if self.opts.step_ckpt is not None:
            path = self.opts.step_ckpt
        if not osp.exists(path):
            if getattr(self.opts, "debug", False):           # train.py:766-768
                self.logger.info(f"[!] WARNING: Unable to find of step {self.opts.step - 1}! Do you really want to do from scratch?")
                return
            raise FileNotFoundError(path)
        ck = torch.load(path, map_location="cpu")
        state = self._strip(ck["model_state"])
        self.model.load_state_dict(state, strict=False)
        if getattr(self.opts, "init_balanced", False):       # train.py:752-754
            self.model.init_new_classifier(self.device)
        strict = self.opts.branch not in ("none", "seg")
        self.model_old.load_state_dict(state, strict=strict)
        self._realias()

    def load_ckpt(self, path):
        """train.py:773-795 — resume."""
        import os.path as osp
        assert osp.isfile(path), f"Error, ckpt not found in {path}"
        ck = torch.load(path, map_location="cpu")
        self.model.load_state_dict(self._strip(ck["model_state"]), strict=True)
        self._realias()
        self.optimizer.load_state_dict(ck["optimizer_state"])
        self.scheduler.load_state_dict(ck["scheduler_state"])
        if "pseudolabeler" in ck and self.pseudolabeler is not None:
            self.pseudolabeler.load_state_dict(self._strip(ck["pseudolabeler"]))
        if "peakgenerator" in ck and self.peakgenerator is not None:
            self.peakgenerator.load_state_dict(self._strip(ck["peakgenerator"]))
        return ck["epoch"] + 1, ck["best_score"]

    def load_seg_ckpt(self, path):
        """train.py:797-812 — phase-1 semantic weights (body / head / cls) + PseudoLabeler + PeakGenerator."""
        import os.path as osp
        assert osp.isfile(path), f"Error, ckpt not found in {path}"
        ck = torch.load(path, map_location="cpu")
        # the whole model state, non-strict, exactly like the reference: a phase-1 checkpoint (branch 'none') simply holds no
        # instance-branch keys, so decoder / instance_head keep the step-0 weights loaded before
        self.model.load_state_dict(self._strip(ck["model_state"]), strict=False)
        self._realias()
        if "pseudolabeler" in ck and self.pseudolabeler is not None:
            self.pseudolabeler.load_state_dict(self._strip(ck["pseudolabeler"]))
        if "peakgenerator" in ck and self.peakgenerator is not None:
            self.peakgenerator.load_state_dict(self._strip(ck["peakgenerator"]))

    def _realias(self):
        """load_state_dict copies in place, so the flat-buffer aliasing survives; only derived caches are stale."""
        from .nn import CACHE
        CACHE.clear()
