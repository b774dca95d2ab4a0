#!/bin/bash
# round 6 soak: the new paths repeated -- ESDF wavefront against the full recompute (small + C2 streams), ESDF / merge / fusion / Octomap tests in a loop
cd $GRAFT_REPO_ROOT; F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL\|initialized'
bad=0
for i in $(seq 1 ${1:-8}); do
  r=$(python tools/esdf_wave_check.py small c2 2>&1 | grep FRAMES_WITH); echo "esdf check $i: $r"; [ "$r" = "FRAMES_WITH_DIFFERENCES 0" ] || bad=$((bad+1))
done
for i in $(seq 1 ${2:-4}); do
  r=$(timeout 900 python -m pytest tests/test_esdf_gpu.py tests/test_octomap_gpu.py tests/test_merge_gpu.py tests/test_fusion_mesh_gpu.py -x -q 2>&1 | grep -v "$F" | tail -1); echo "tests $i: $r"; echo "$r" | grep -q failed && bad=$((bad+1))
done
echo "SOAK_FAILURES $bad"
