"""cl4_fill_zero (api.hip fill_bytes_kernel) on odd offsets / sizes against a canary-padded buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cl4wsis_amd import _hip
from cl4wsis_amd._hip import call, stream
import ctypes as C
dev = torch.device("cuda:0")
for off in (0, 1, 3, 15, 16, 17, 64):
    for n in (1, 2, 15, 16, 17, 31, 32, 33, 255, 4097, 1 << 20, (1 << 20) + 5, 63 * 1000 * 1000 + 3):
        buf = torch.full((off + n + 64,), 7, dtype=torch.uint8, device=dev)
        call("cl4_fill_zero", C.c_void_p(buf.data_ptr() + off), n, stream())
        torch.cuda.synchronize()
        assert bool((buf[:off] == 7).all()) and bool((buf[off:off + n] == 0).all()) and bool((buf[off + n:] == 7).all()), (off, n)
print("fill ok")
