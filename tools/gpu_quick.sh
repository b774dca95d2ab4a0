#!/bin/bash
# quick GPU check: smoke + parity tests (+ optional extra command)
mkdir -p gpurun_out; O=gpurun_out
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -q "$@" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/smoke.log; tail -30 $O/pytest_gpu.log
