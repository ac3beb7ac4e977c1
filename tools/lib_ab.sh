# bench headline with two builds of libcmh.so on one box: bash tools/lib_ab.sh <path of the B library>
cd $GRAFT_REPO_ROOT
for lib in ${AB_SEQ:-A B A B}; do
  if [ $lib = A ]; then unset CMH_LIB; else export CMH_LIB=$1; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-dense-text --no-input-pipeline --no-config-legs --no-map-eval --no-precision-legs --no-towers-ab 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); r = d['roofline']
print('lib $lib', d['value'], 'pairs/s', d['ms_per_step'], 'ms', 'frac', r['frac'], 'avg us', r['avg_launch_us'])"
done
