#!/bin/bash
# usage: gpu_env_ab.sh "ENV=1 probe args" ...   (first token may be an env assignment or '-')
mkdir -p gpurun_out; O=gpurun_out/envab.log; : > $O
for a in "$@"; do
  set -- $a; e=$1; shift
  if [ "$e" = "-" ]; then r=$(timeout 120 python tools/host_overhead_probe.py "$@" 2>&1 | tail -1); else r=$(env $e timeout 120 python tools/host_overhead_probe.py "$@" 2>&1 | tail -1); fi
  echo "[$a] $r" >> $O
done
cat $O
