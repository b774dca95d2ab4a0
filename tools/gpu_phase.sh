#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/phase.log; : > $O
timeout 600 python -m pytest tests/test_tsdf_parity_gpu.py -m gpu -q -x -k "$TESTK" 2>&1 | tail -15 >> $O
for a in "$@"; do
  echo "$a: $(timeout 120 python tools/host_overhead_probe.py $a 2>&1 | tail -1)" >> $O
done
cat $O
