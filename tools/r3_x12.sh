# round 3, experiments X1 (event method) and X2 (store cache policy); usage on the GPU box: bash tools/r3_x12.sh
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_x12; mkdir -p $O
C=$R/clip-based-cross-modal-hashing_amd/csrc
B="--steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-dense-text --no-input-pipeline --no-precision-legs --no-map-eval"
cd $R
CMH_GEMM_PROF_DUMP=1 CMH_GEMM_PROF_BRACKET=1 python3 bench.py $B > $O/plain_bracket.json 2> $O/plain_bracket.err
for v in plain nt sc1 plain nt sc1; do
  L=$C/build/libcmh.so; [ $v = nt ] && L=$C/build_nt/libcmh.so; [ $v = sc1 ] && L=$C/build_sc1/libcmh.so
  CMH_LIB=$L CMH_GEMM_PROF_DUMP=1 python3 bench.py $B >> $O/$v.json 2>> $O/$v.err
  CMH_LIB=$L python3 tools/gemm_bench2.py --sets vision,text >> $O/${v}_gemm2.txt 2>&1
done
grep -h -o '"value": [0-9.]*\|"achieved": [0-9.]*\|gemm_ms_per_step_serialized": [0-9.]*' $O/*.json
