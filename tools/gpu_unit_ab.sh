#!/bin/bash
# default bench (steady leg) over values of the "unit" option (segments of a batch above which a brick is walked in parts)
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  opts=""; for kv in ${spec//,/ }; do opts="$opts --opt $kv"; done
  timeout 200 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --steady 300 $opts 2>/dev/null | python tools/bench_brief.py "[$spec]"
done
