#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of access patterns with a known byte count (tools/ubench/fetch_calib.hip) -> gpurun_out/calib/fetch_calibration.txt
O=$GRAFT_REPO_ROOT/gpurun_out/calib; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/fetch_calib.hip -o /tmp/fetch_calib || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p_$c -o p -- /tmp/fetch_calib > $O/run_$c.log 2>&1
  cp $(find $O/p_$c -name "*counter_collection.csv" | head -1) $O/$c.csv; rm -rf $O/p_$c
done
python - $O <<'PY' | tee $O/fetch_calibration.txt
import csv, sys, os
O = sys.argv[1]
known = [l.strip().split(",") for l in open(os.path.join(O, "run_FETCH_SIZE.log")) if l.count(",") == 3 and not l.startswith("kernel")]
def counters(name):
    rows = sorted(csv.DictReader(open(os.path.join(O, name + ".csv"))), key=lambda r: int(r["Dispatch_Id"]))
    return [(r["Kernel_Name"].split("(")[0].replace("void ", ""), float(r["Counter_Value"])) for r in rows if r["Counter_Name"] == name and not r["Kernel_Name"].startswith("__amd")]
f, w = counters("FETCH_SIZE"), counters("WRITE_SIZE")
print("rocprofv3 FETCH_SIZE / WRITE_SIZE (reported in KiB) against known byte counts, 1 GiB buffer, every address touched once (gfx950, ROCm 7)")
dur = {}
rows = sorted(csv.DictReader(open(os.path.join(O, "FETCH_SIZE.csv"))), key=lambda r: int(r["Dispatch_Id"]))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if r["Counter_Name"] == "FETCH_SIZE" and not r["Kernel_Name"].startswith("__amd")]
print(f"{'pattern':24s} {'requested MB':>13s} {'128-B lines':>12s} {'counter MB':>11s} {'ctr/requested':>14s} {'B per 128-B line':>17s} {'us':>8s} {'TB/s if 128 B per line':>23s}")
for i, (nm, req, l64, l128) in enumerate(known):
    c = (w if nm.startswith("write") else f)[i][1] * 1024.0
    req, l128 = float(req), float(l128)
    print(f"{nm:24s} {req / 1e6:13.1f} {l128 / 1e6:11.2f}M {c / 1e6:11.1f} {c / req:14.3f} " + f"{c / l128:17.1f} {durs[i]:8.1f} {l128 * 128 / durs[i] / 1e6:23.2f}")
print("""reading: FETCH_SIZE = 64 B per distinct 128-byte line requested, whatever the width of the load -- a coalesced stream reports half its bytes (the
guide's calibration), a 4-byte gather reports 64 B per line it touches, i.e. the memory side moves 128-byte lines and 2 x FETCH_SIZE is the traffic
for narrow gathers too (the durations agree: the 4-byte-per-128-byte gather touches 1 GiB of lines in the time the 1 GiB stream takes).  WRITE_SIZE = the bytes of full-line streaming
stores, and 32 B (one sector) per partial store.  profiles/r04_traffic.json and bench.py's `traffic` use 2 x FETCH_SIZE + WRITE_SIZE.""")
PY
