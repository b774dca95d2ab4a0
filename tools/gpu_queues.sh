#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/queues.log; : > $O
for q in 4 5 6 8; do for a in "overlap=3" "overlap=4" "overlap=4 phases=1"; do
  echo "GPU_MAX_HW_QUEUES=$q $a: $(GPU_MAX_HW_QUEUES=$q timeout 120 python tools/host_overhead_probe.py $a 2>&1 | tail -1)" >> $O
done; done
cat $O
