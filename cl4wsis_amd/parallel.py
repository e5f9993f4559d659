"""Data-parallel plumbing: one process per GPU, gradients of the trainable sub-net live in ONE flat fp32 buffer
that is averaged across ranks with a few large RCCL all-reduces (backend "nccl" = RCCL over xGMI), issued on a
side stream as soon as each bucket's gradients are final so they overlap the remaining backward kernels.

Replaces the three DistributedDataParallel wrappers of the reference (train.py:187-196). Unlike the reference,
whose phase-2 path bypasses DDP.forward and therefore never all-reduces, gradients ARE averaged.
The same code runs on CPU tensors over gloo (tests/test_parallel_cpu.py).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Bootstrap from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*). Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    solo = os.environ.get("CL4_FORCE_COLLECTIVES") == "1"     # world-size-1 rehearsal of the N > 1 path (see BucketedAllReduce.solo)
    if world > 1 or solo:
        # The host driver of the MI355X pool only supports dmabuf IPC: without this RCCL's peer mappings (and CUDA-tensor sharing between
        # processes) fail with `hipIpcGetMemHandle: invalid argument`. The HSA runtime reads it when the process makes its first HIP call,
        # so it is set here, before anything below touches the GPU — run.py and bench.py both come through this function and therefore
        # start RCCL identically; an explicit setting of the caller wins.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if (world > 1 or solo) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("CL4_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        elif torch.cuda.is_available():
            local = local % torch.cuda.device_count()
        dist.init_process_group(backend, init_method="env://", rank=rank, world_size=world, **kw)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def agree_on_direct_comm(make_lib, dev, world, rank_, group=None):
    """Set up the library's own RCCL communicator on every rank, or on none. `make_lib()` returns an object with world(), destroy(),
    unique_id() -> 128 bytes, init(rank, world, id) and probe() (a collective on the new communicator); any of them may raise on
    any rank. The decision is COLLECTIVE: every rank takes part in every torch.distributed call below whatever happened to it
    locally (rank 0 broadcasts a zeroed id plus a failure flag when it could not make one), the ranks enter the collective init /
    probe only when all of them can, and they agree on the verdict with an all-reduce(MIN) of their success flags — either every
    rank returns True, or every rank destroys what it created and returns False (torch.distributed's all_reduce then carries the
    gradients). A partial fallback would pair cl4_allreduce_bucket on some ranks with dist.all_reduce on the others and deadlock
    the first training step."""
    import warnings

    def all_agree(flag, op=dist.ReduceOp.MIN):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=op, group=group)
        return bool(int(t.item()))

    ok, why, lib = True, "", None
    try:
        lib = make_lib()
    except Exception as e:      # noqa: BLE001 - any failure must degrade to the torch.distributed collectives, on every rank
        ok, why = False, str(e)
    have_world = -1
    if ok:
        try:
            have_world = lib.world()
        except Exception as e:  # noqa: BLE001
            ok, why = False, str(e)
    # (1) does any rank need a new communicator? then every rank makes a new one
    if all_agree((not ok) or have_world != world, dist.ReduceOp.MAX):
        if ok and have_world not in (0, -1):
            try:
                lib.destroy()
            except Exception as e:  # noqa: BLE001
                ok, why = False, str(e)
        raw, flag = bytes(128), 1
        if rank_ == 0:
            try:
                if not ok:
                    raise RuntimeError(why)
                raw = lib.unique_id()
            except Exception as e:  # noqa: BLE001
                ok, why, flag, raw = False, str(e), 0, bytes(128)
        t = torch.frombuffer(bytearray(raw[:128]) + bytearray([flag]), dtype=torch.uint8).to(dev)
        dist.broadcast(t, src=0, group=group)                         # always executed, on every rank
        got = bytes(t.cpu().numpy().tobytes())
        if got[128] == 0 and ok:
            ok, why = False, "rank 0 could not create an RCCL id"
        # (2) ncclCommInitRank is itself a collective: enter it together or not at all
        if all_agree(ok):
            try:
                lib.init(rank_, world, got[:128])
            except Exception as e:  # noqa: BLE001
                ok, why = False, str(e)
        elif ok:
            ok, why = False, "a peer rank could not initialise"
    # (3) the probe is a collective on the new communicator: only if every rank holds one
    if all_agree(ok):
        try:
            lib.probe()
        except Exception as e:  # noqa: BLE001
            ok, why = False, str(e)
    elif ok:
        ok, why = False, "a peer rank holds no communicator"
    # (4) the verdict, agreed
    if all_agree(ok):
        return True
    try:
        if lib is not None and lib.world() != 0:
            lib.destroy()
    except Exception:           # noqa: BLE001
        pass
    warnings.warn(f"cl4wsis_amd: direct RCCL communicator not used ({why or 'a peer rank failed'}); every rank uses torch.distributed all_reduce")
    return False


class FlatBuffer:
    """Parameters (and gradients) of a list of tensors packed into one contiguous fp32 buffer, in the given order.

    `views[i]` aliases buffer[offset_i : offset_i + numel_i] with the tensor's shape."""

    def __init__(self, shapes, device):
        self.sizes = [int(torch.Size(s).numel()) for s in shapes]
        self.offsets = [0]
        for n in self.sizes:
            self.offsets.append(self.offsets[-1] + n)
        self.numel = self.offsets[-1]
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.views = [self.flat[o:o + n].view(s) for o, n, s in zip(self.offsets, self.sizes, shapes)]


class GradBuckets:
    """Bucketed, asynchronous gradient averaging over a FlatBuffer.

    Buckets are contiguous index ranges of the flat gradient buffer. `ready(i)` marks tensor i final; once all tensors
    of a bucket are final the bucket is all-reduced (SUM then scaled by 1/world inside `finish`, or AVG natively on
    RCCL) on `comm_stream`. `finish()` joins the communication before the optimizer reads the gradients."""

    def __init__(self, flat, bucket_bytes=32 << 20, process_group=None):
        self.fb = flat
        self.group = process_group
        self.world = world_size()
        # single-rank rehearsal of the collective path (a 1-GPU box can still run RCCL with world size 1): behave as N > 1
        self.solo = self.world == 1 and dist.is_initialized() and os.environ.get("CL4_FORCE_COLLECTIVES") == "1"
        self.distributed = self.world > 1 or self.solo
        self.bounds = []            # (first tensor, last tensor exclusive)
        start, acc = 0, 0
        for i, n in enumerate(flat.sizes):
            acc += n * 4
            if acc >= bucket_bytes:
                self.bounds.append((start, i + 1))
                start, acc = i + 1, 0
        if start < len(flat.sizes):
            self.bounds.append((start, len(flat.sizes)))
        self.bucket_of = {}
        for b, (lo, hi) in enumerate(self.bounds):
            for i in range(lo, hi):
                self.bucket_of[i] = b
        self.is_cuda = flat.flat.is_cuda
        self.native_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.comm_stream = torch.cuda.Stream() if (self.is_cuda and self.distributed) else None
        # The bucket all-reduce goes through the library's own RCCL communicator (cl4_comm_init / cl4_allreduce_bucket,
        # include/cl4wsis.h) when the job runs on RCCL; torch.distributed then only bootstraps (it carries the 128-byte id).
        # CL4_COMM=torch keeps torch.distributed's all_reduce on the data path (also the gloo rehearsal path).
        self.direct = False
        if self.is_cuda and self.distributed and self.native_avg and process_group is None and os.environ.get("CL4_COMM", "native") != "torch":
            self.direct = self._init_direct()
        self.timing = False
        self._events = []
        self.reset()

    def _init_direct(self):
        """One RCCL communicator per process (cl4_comm_*, include/cl4wsis.h), agreed on by every rank: see `agree_on_direct_comm`."""
        import ctypes as C
        from . import _hip
        dev = self.fb.flat.device
        world = self.world

        class _Lib:
            def __init__(self):
                self.lib = _hip.load()

            def world(self):
                return self.lib.cl4_comm_world()

            def destroy(self):
                _hip.call("cl4_comm_destroy")

            def unique_id(self):
                buf = (C.c_char * 128)()
                _hip.call("cl4_comm_unique_id", buf)
                return bytes(buf.raw)

            def init(self, r, w, raw):
                with torch.cuda.device(dev):
                    _hip.call("cl4_comm_init", r, w, C.c_char_p(raw))

            def probe(self):
                probe = torch.ones(8, dtype=torch.float32, device=dev)
                s = torch.cuda.current_stream(dev)
                _hip.call("cl4_allreduce_bucket", C.c_void_p(probe.data_ptr()), 8, 0, C.c_void_p(s.cuda_stream))
                s.synchronize()
                if not bool((probe == float(world)).all()):
                    raise RuntimeError(f"probe all-reduce returned {probe.tolist()} for world size {world}")

        return agree_on_direct_comm(_Lib, dev, world, rank())

    def reset(self):
        self.pending = [hi - lo for lo, hi in self.bounds]
        self.handles = []
        self.launched = [False] * len(self.bounds)

    def _launch(self, b):
        lo, hi = self.bounds[b]
        chunk = self.fb.flat[self.fb.offsets[lo]:self.fb.offsets[hi]]
        self.launched[b] = True
        if not self.distributed:
            return
        if self.is_cuda:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            if self.direct:
                import ctypes as C
                from . import _hip
                _hip.call("cl4_allreduce_bucket", C.c_void_p(chunk.data_ptr()), chunk.numel(), 1,
                          C.c_void_p(self.comm_stream.cuda_stream))
                return
            with torch.cuda.stream(self.comm_stream):
                if self.native_avg:
                    dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group)      # RCCL
                else:                                                                   # gloo rehearsal on a GPU
                    dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
                    chunk.div_(self.world)
        else:
            self.handles.append((dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True), chunk))

    def comm_stats(self):
        """Exposed all-reduce time per step (ms) since `timing` was switched on — needs a device synchronisation first."""
        if not self._events:
            return None
        ms = [a.elapsed_time(b) for a, b in self._events]
        return {"exposed_ms_per_step": sum(ms) / len(ms), "steps": len(ms), "buckets": len(self.bounds),
                "bucket_mb": [round(4e-6 * (self.fb.offsets[hi] - self.fb.offsets[lo]), 2) for lo, hi in self.bounds],
                "path": "cl4_allreduce_bucket (direct RCCL)" if self.direct else "torch.distributed.all_reduce"}

    def ready(self, i):
        b = self.bucket_of[i]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def finish(self):
        for b in range(len(self.bounds)):
            if not self.launched[b]:
                self._launch(b)
        if self.distributed:
            if self.is_cuda:
                cur = torch.cuda.current_stream()
                if self.timing:      # compute stream reaches the join at e0, may proceed at e1: e1 - e0 = exposed communication
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                    cur.wait_stream(self.comm_stream)
                    e1.record(cur)
                    self._events.append((e0, e1))
                else:
                    cur.wait_stream(self.comm_stream)
            else:
                for h, chunk in self.handles:
                    h.wait()
                    chunk.div_(self.world)
        self.reset()
