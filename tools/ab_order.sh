for o in 0 1 2 3; do
  CMH_GEMM_ORDER=$o timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-map-eval > gpurun_out/ab_$o.log 2>&1
  echo "order=$o $(tail -1 gpurun_out/ab_$o.log | grep -o '"value": [0-9.]*') $(tail -1 gpurun_out/ab_$o.log | grep -o '"achieved": [0-9.]*')"
done
