"""ORACLE helper (test infrastructure): deterministic, name-keyed synthetic weights.

The reference checkpoints are not available offline, and default initialisation depends on module
construction order. Filling every state_dict entry from a generator seeded with crc32(key) gives the
reference model, the oracle and the HIP modules bit-identical weights from the key names alone.
"""
import zlib

import torch


def fill_(sd, seed=0):
    """In-place deterministic fill of a state_dict (tensors are modified, the dict is returned)."""
    for k in sorted(sd.keys()):
        t = sd[k]
        g = torch.Generator().manual_seed((zlib.crc32(k.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        if k.endswith("num_batches_tracked"):
            t.zero_()
        elif k.endswith("running_var"):
            t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75)
        elif k.endswith("running_mean"):
            t.copy_(torch.randn(t.shape, generator=g) * 0.1)
        elif t.dim() == 1 and k.endswith("weight"):      # BN / ABN gamma
            t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75)
        elif t.dim() == 1:                                # biases
            t.copy_(torch.randn(t.shape, generator=g) * 0.1)
        else:                                             # conv weights: variance-preserving
            fan_in = t[0].numel()
            t.copy_(torch.randn(t.shape, generator=g) * (1.0 / fan_in) ** 0.5)
    return sd
