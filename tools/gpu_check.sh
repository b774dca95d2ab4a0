#!/bin/bash
# quick state check: GPU tests (optionally -k "$1"), default-size bench line on stdout only
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q ${1:+-k "$1"} 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -6
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | wc -l
