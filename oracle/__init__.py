"""ORACLE — test infrastructure only.

CPU restatement (PyTorch-CPU fp32 + numpy/scipy) of the CL4WSIS phase-2 hot path, pinned by golden
vectors generated from the upstream reference (tests/golden/). Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; cl4wsis_amd never does.
"""
