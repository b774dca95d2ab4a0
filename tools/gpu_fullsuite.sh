#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/suite; mkdir -p $O
timeout 1700 python -m pytest tests -x -q -m gpu --timeout=300 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc" >> $O/pytest.log; tail -4 $O/pytest.log
if [ $rc -ne 0 ]; then grep -v "^$" $O/pytest.log | grep -B2 -A25 "Error\|assert" | head -80; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
