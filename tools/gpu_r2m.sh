#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/esdf_probe.py << 'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
frames = list(syn.sphere_room_stream(12))
md = float(sys.argv[1])
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
for i in range(12):
    m.recast_depth_to_map(frames[i][0], frames[i][1], frames[i][2], None); m.update_esdf(max_dist=md)
print("stats", m.esdf_stats())
PY
for a in 1.0 0.05; do
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_esdf -o p -- python /tmp/esdf_probe.py $a > $O/r2m_probe.log 2>&1
  grep "stats" $O/r2m_probe.log
  f=$(find $O/prof_esdf -name "*kernel_trace.csv" | head -1)
  python - "$f" << 'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "esdf" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-19:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(f'{r["Kernel_Name"].split("(")[0][-22:]:22s} start {(int(r["Start_Timestamp"])-t0)/1e3:8.1f} us  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  grid {r.get("Grid_Size_X","?")} lds {r.get("LDS_Block_Size","?")} vgpr {r.get("VGPR_Count","?")} scratch {r.get("Scratch_Size","?")}')
PY
  rm -rf $O/prof_esdf
done
