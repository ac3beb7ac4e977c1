# fp8 leg with and without the 160-row e4m3-output tile, one box: bash tools/fp8_ab.sh
cd $GRAFT_REPO_ROOT
for bm in auto 128 auto 128; do
  if [ $bm = auto ]; then unset CMH_GEMM_BM; else export CMH_GEMM_BM=$bm; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-dense-text --no-input-pipeline --no-config-legs --no-map-eval 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
f = d['fp8_mode']
print('CMH_GEMM_BM=$bm', 'bf16', d['value'], 'fp8', f['pairs_per_s'], 'x', f['speedup_vs_headline'], 'frac', f['roofline']['frac'], 'gemm ms', f['roofline']['gemm_ms_per_step_serialized'])"
done
