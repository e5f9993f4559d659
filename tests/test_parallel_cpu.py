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
