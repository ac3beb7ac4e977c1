# configs[0] step with the rows kernel taking more rows: bash tools/dchmt_ab.sh
cd $GRAFT_REPO_ROOT
for mm in 512 2048 4096 512 2048; do
  CMH_GEMM_ROWS_MAX_M=$mm python3 -c "
import sys, json, torch
sys.argv=['bench.py']
import bench, bench_configs
d = bench_configs.dchmt_epoch(torch.device('cuda:0'), cpu_sample=False)
print('CMH_GEMM_ROWS_MAX_M=$mm', d['ms_per_step'], 'ms per step, valid', d['valid_s'], 'first loss', d['first_loss'], 'mAP', d['mAP_i2t'])" 2>/dev/null | tail -1
done
