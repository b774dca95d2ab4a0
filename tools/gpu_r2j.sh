#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_esdf_gpu.py tests/test_tsdf_parity_gpu.py -m gpu -x -q 2>&1 | tail -8 > $O/r2j_esdf.log; cat $O/r2j_esdf.log
cat > /tmp/esdf_probe.py << 'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
frames = list(syn.sphere_room_stream(40))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
md = float(sys.argv[1]); full = int(sys.argv[2])
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH); m.set_option("esdf_full", full)
for i in range(20):
    m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None); m.update_esdf(max_dist=md)
m.sync(); t0 = time.perf_counter(); te = 0.0
for i in range(20, 40):
    m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
    m.sync(); a = time.perf_counter(); m.update_esdf(max_dist=md); te += time.perf_counter() - a
dt = time.perf_counter() - t0
print(f"max_dist {md} full {full}: {20/dt:.0f} frames/s with an ESDF update per frame, {1e3*te/20:.3f} ms per update, stats {m.esdf_stats()}")
PY
for a in "1.0 0" "0.4 0"; do
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_esdf -o p -- python /tmp/esdf_probe.py $a > $O/r2j_probe.log 2>&1
  grep "max_dist" $O/r2j_probe.log
  f=$(find $O/prof_esdf -name "*kernel_stats.csv" | head -1)
  python - "$f" << 'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    n = r["Name"].split("(")[0].replace("void ", "")[:50]
    print(f"{n:50s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
  rm -rf $O/prof_esdf
done
