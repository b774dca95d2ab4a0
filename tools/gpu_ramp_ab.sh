#!/bin/bash
# developer A/B: the ramp-up of a 20-frame burst (first batches after the pipeline ran dry)
cd $GRAFT_REPO_ROOT
for o in "ramp=2 ramp_size=4" "ramp=1 ramp_size=4" "ramp=0" "ramp=3 ramp_size=4" "ramp=2 ramp_size=2" "ramp=1 ramp_size=2" "ramp=2 ramp_size=3" "ramp=4 ramp_size=4" "ramp=2 ramp_size=5" "ramp=2 ramp_size=6"; do
  echo "== $o"; timeout 100 python tools/burst_probe.py $o 2>&1 | grep profiling
done
