# The gradient buckets of a training step issued from INSIDE the backward pass (dist_utils.GradSync: a tower's slice leaves the moment
# its part of the backward has written it) against all of them issued behind it (CMH_GRADSYNC_DEFER=1), on RCCL in a forced group of
# one rank (CMH_FORCE_DIST=1): bench.py's 20-step DSPH training leg, alternating.  usage (GPU box): bash tools/gradsync_overlap.sh
R=$GRAFT_REPO_ROOT
for d in 0 1 0 1; do
  CMH_FORCE_DIST=1 CMH_DIST_BACKEND=nccl CMH_GRADSYNC_DEFER=$d MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29500 + d + RANDOM % 200)) RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
    python3 $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-config-legs --no-input-pipeline --no-map-eval --no-dense-text --no-towers-ab --no-precision-legs 2>$R/gpurun_out/gradsync_err.txt |
    python3 -c "
import json, sys
r = json.loads([ln for ln in sys.stdin.read().splitlines() if ln.startswith('{') and '\"metric\"' in ln][-1])
print('deferred=$d  collectives', r['collectives'], ' train_step ms', r['train_step']['ms'], ' loss', r['train_step']['loss'])"
done
