# The bf16 training step with the residual-gradient stream carried as bf16 (CMH_GRAD_STREAM16=1, the default) against f32 (=0): bench.py's
# 20-step DSPH training leg, alternating on one box.  usage (GPU box): bash tools/grad_stream16_ab.sh
R=$GRAFT_REPO_ROOT
for s in 1 0 1 0; do
  CMH_GRAD_STREAM16=$s python3 $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-config-legs --no-input-pipeline --no-map-eval \
    --no-dense-text --no-towers-ab --no-precision-legs 2>$R/gpurun_out/stream16_err.txt |
    python3 -c "
import json, sys
r = json.loads([ln for ln in sys.stdin.read().splitlines() if ln.startswith('{') and '\"metric\"' in ln][-1])
print('stream16=$s  train_step ms', r['train_step']['ms'], ' loss after 20 steps', r['train_step']['loss'])"
done
