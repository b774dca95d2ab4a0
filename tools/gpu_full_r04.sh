cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/full; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/tests.log
timeout 200 python tools/seq_probe.py --frames 400 --warmup 40 2>&1 | grep -v "^TSDF\|^Export" | tee $O/probe_steady.log
