cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -5
timeout 300 python bench.py --config 4 --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('config4', round(j['value'],1), 'fps', {k:(round(v,3) if isinstance(v,float) else v) for k,v in j['config'].items() if k!='workload'})"
