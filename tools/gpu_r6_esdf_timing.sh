#!/bin/bash
cd $GRAFT_REPO_ROOT
TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_timing.so python tools/esdf_rounds_probe.py 1.0 24 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -45
