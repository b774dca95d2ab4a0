"""CPU oracle for the TaichiSLAM dense-mapping hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (taichislam_amd) never does.  Parity: pinned to the reference's own source executed on tools/ti_seq (oracle/tsl_oracle.h), not to Taichi itself.
"""
from .binding import *  # noqa: F401,F403
