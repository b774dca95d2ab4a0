cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/roles; mkdir -p $O
cd /tmp && TSL_SEQ_SPLIT_ROLES=1 TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_testhooks.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o p -- python $GRAFT_REPO_ROOT/tools/seq_probe.py --frames 72 > $O/prof.log 2>&1
python - "$(find $O/t -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rp = [r for r in rows if "k_seq_replay" in r["Kernel_Name"]]
for i in range(0, len(rp), 2):
    a, b = rp[i], rp[i+1]
    print("batch", i//2, "short %.0f us (grid %s)" % ((int(a["End_Timestamp"])-int(a["Start_Timestamp"]))/1e3, a.get("Grid_Size_X") or a.get("Grid_Size")), "long %.0f us (grid %s)" % ((int(b["End_Timestamp"])-int(b["Start_Timestamp"]))/1e3, b.get("Grid_Size_X") or b.get("Grid_Size")))
PY
rm -rf $O/t
