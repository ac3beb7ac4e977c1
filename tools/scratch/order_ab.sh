cd $GRAFT_REPO_ROOT
for g in 0 3 4 6 0 4; do
  CMH_GEMM_ORDER=$g python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-dense-text --no-input-pipeline --no-config-legs --no-map-eval --no-precision-legs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); r = d['roofline']
print('order group $g', d['value'], 'pairs/s', d['ms_per_step'], 'ms', 'frac', r['frac'], 'avg us', r['avg_launch_us'])"
done
