"""CPU oracle for the TaichiSLAM dense-mapping hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (taichislam_amd) never does.  PARITY UNPINNED (see oracle/tsl_oracle.h).
"""
from .binding import *  # noqa: F401,F403
