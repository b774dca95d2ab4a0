#!/bin/bash
# default bench (steady leg) with variant libraries lib/libtaichislam_hip_<tag>.so ("base" = the default library)
cd $GRAFT_REPO_ROOT
for tag in "$@"; do
  if [ "$tag" = base ]; then unset TSL_LIB; else export TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_$tag.so; fi
  for i in 1 2; do timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --steady 300 2>/dev/null | python tools/bench_brief.py "[$tag]"; done
done
