# round 3: store-cache-policy sweep of the wide GEMM's 16-byte output stores (diagnostic builds build_sp<n>)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_sp; mkdir -p $O
C=$R/clip-based-cross-modal-hashing_amd/csrc
B="--steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-dense-text --no-input-pipeline --no-precision-legs --no-map-eval"
cd $R
for rep in 1 2; do
for v in 0 1 2 3 4 5 6; do
  L=$C/build/libcmh.so; [ $v != 0 ] && L=$C/build_sp$v/libcmh.so
  CMH_LIB=$L python3 bench.py $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sp$v overlapped value', d['value'], 'gemm_ms', d['roofline']['gemm_ms_per_step_serialized'], 'TF', d['roofline']['achieved'])" >> $O/summary.txt
  CMH_LIB=$L python3 bench.py $B --no-overlap-towers 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sp$v serialized value', d['value'], 'ms', d['ms_per_step'], 'gemm_ms', d['roofline']['gemm_ms_per_step_serialized'])" >> $O/summary.txt
  CMH_LIB=$L python3 tools/gemm_bench2.py --sets vision,text 2>&1 | grep block | sed "s/^/sp$v /" >> $O/summary.txt
done
done
cat $O/summary.txt
