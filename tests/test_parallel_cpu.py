"""CPU, world_size 2 over gloo: the data-parallel gradient path (flat buffer + bucketed async all-reduce) averages
gradients across ranks exactly like DDP would, whatever order the tensors become ready in."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from cl4wsis_amd.parallel import FlatBuffer, GradBuckets, init_distributed
    r, w, _ = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    shapes = [(7, 3), (11,), (2, 5, 5), (1,), (64, 9)]
    fb = FlatBuffer(shapes, torch.device("cpu"))
    gb = GradBuckets(fb, bucket_bytes=256)
    ok = True
    for it in range(3):
        g = torch.Generator().manual_seed(100 * it + rank)
        for v in fb.views:
            v.copy_(torch.randn(v.shape, generator=g))
        expect = []
        for i, s in enumerate(shapes):
            acc = torch.zeros(s)
            for rr in range(world):
                gg = torch.Generator().manual_seed(100 * it + rr)
                vals = [torch.randn(sh, generator=gg) for sh in shapes]
                acc += vals[i]
            expect.append(acc / world)
        for i in ([4, 2, 0, 3, 1] if it % 2 else [0, 1, 2, 3, 4]):   # readiness order differs per iteration
            gb.ready(i)
        gb.finish()
        ok &= all(torch.allclose(v, e, atol=1e-6) for v, e in zip(fb.views, expect))
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_gloo_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


class _FakeComm:
    """A stand-in for the library's RCCL entry points whose calls fail where a scenario says so: {"make" | "id" | "init" | "probe":
    set of ranks}. The fake 'communicator' is a gloo all-reduce, so a probe that only some ranks enter would hang — exactly what
    the agreement protocol must never let happen."""

    def __init__(self, rank, world, fail):
        self.rank, self.world_, self.fail, self.have = rank, world, fail, 0
        if rank in fail.get("make", ()):
            raise RuntimeError("library missing on this rank")

    def world(self):
        return self.have

    def destroy(self):
        self.have = 0

    def unique_id(self):
        if self.rank in self.fail.get("id", ()):
            raise RuntimeError("ncclGetUniqueId failed")
        return bytes(range(128))

    def init(self, r, w, raw):
        assert raw == bytes(range(128)), "every rank must receive rank 0's id"
        if self.rank in self.fail.get("init", ()):
            raise RuntimeError("ncclCommInitRank failed")
        self.have = w

    def probe(self):
        t = torch.ones(8)
        dist.all_reduce(t)              # collective: hangs unless every rank calls it
        if self.rank in self.fail.get("probe", ()):
            raise RuntimeError("probe returned a wrong sum")
        assert bool((t == float(self.world_)).all())


_SCENARIOS = [({}, True), ({"make": {1}}, False), ({"make": {0}}, False), ({"id": {0}}, False), ({"init": {1}}, False),
              ({"init": {0}}, False), ({"probe": {1}}, False), ({"probe": {0, 1}}, False)]


def _agree_worker(rank, world, port, ret):
    import warnings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from cl4wsis_amd.parallel import agree_on_direct_comm, init_distributed
    init_distributed(backend="gloo")
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for fail, _ in _SCENARIOS:
            holder = []

            def make():
                holder.append(_FakeComm(rank, world, fail))
                return holder[0]

            verdict = agree_on_direct_comm(make, torch.device("cpu"), world, rank)
            left = holder[0].have if holder else 0
            out.append((verdict, left))
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_direct_comm_fallback_is_agreed_across_ranks_gloo_world2():
    """a failure of the direct RCCL set-up on ONE rank (no id on rank 0, init or probe failing on some ranks) must send
    EVERY rank to the torch.distributed path — with no rank left waiting in a broadcast or a collective — and leave no half-made
    communicator behind."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_agree_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        for (fail, expect), (verdict, left) in zip(_SCENARIOS, ret[r]):
            assert verdict == expect, (r, fail, verdict)
            assert left == (world if expect else 0), (r, fail, left)
