#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 200 python tools/burst_probe.py 2>/dev/null | tail -6
bash tools/gpu_timeline.sh 2>&1 | tail -75
