"""cl4wsis_amd — MI355X (gfx950) native phase-2 training path of CL4WSIS.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed only).
Compute: hand-written HIP kernels in cl4wsis_amd/csrc behind the C ABI of include/cl4wsis.h.
"""
__version__ = "0.1.0"
