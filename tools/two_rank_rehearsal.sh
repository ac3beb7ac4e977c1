#!/bin/bash
# Two trainer processes on ONE GPU over gloo (the 1-GPU box has no second card): checks the one-process-per-GPU flow of main.py —
# sharded batches, loss-side all-gather, gradient means started inside backward, sharded evaluation — by comparing the two ranks' logs.
# usage: two_rank_rehearsal.sh <dir for the logs>   (checkpoints stay under /tmp)
LOGS=$(realpath -m "${1:-/tmp/two_rank_logs}"); mkdir -p "$LOGS"
cd "$(dirname "$0")/../clip-based-cross-modal-hashing_amd"
export CMH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
RUN=/tmp/two_rank_run; CK=/tmp/two_rank_vitb32.pt
python - <<PY
import sys, torch
sys.path.insert(0, "../tests/golden"); sys.path.insert(0, ".")
import recipe
from model.base.model import CLIP
torch.manual_seed(0)
torch.save(CLIP(**recipe.CLIP_VITB32).state_dict(), "$CK")          # random-init ViT-B/32 (no checkpoint download here)
PY
rc=0
for M in ${METHODS:-DSPH MITH}; do
  rm -rf "$RUN/$M"
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    main.py -clip-path $CK --method $M --dataset synthetic --output-dim 64 --epochs 2 --synthetic-size 640 --train-num 256 --query-num 64 \
    --batch-size 32 --num-workers 0 --save-dir "$RUN/$M" --gemm-dtype bf16 > "$LOGS/$M.out" 2>&1 || rc=1
  for f in $(find "$RUN/$M" -name "train*.log"); do cp "$f" "$LOGS/$M.$(basename $f)"; done
  if [ $rc != 0 ]; then grep -v "Warning\|^\[W" "$LOGS/$M.out" | grep -B14 "Error" | head -60; exit 1; fi
  echo "== $M rank 0"; grep -h "MAP(i->t)" "$LOGS/$M.train.log" | sed 's/MAX MAP.*//' | cut -c30-110
  echo "-- $M rank 1"; grep -h "MAP(i->t)" "$LOGS/$M.train.rank1.log" | sed 's/MAX MAP.*//' | cut -c30-110
  a=$(grep -h "MAP(i->t)" "$LOGS/$M.train.log" | sed 's/.*INFO//; s/MAX MAP.*//'); b=$(grep -h "MAP(i->t)" "$LOGS/$M.train.rank1.log" | sed 's/.*INFO//; s/MAX MAP.*//')
  ca=$(grep -h "replica checksum" "$LOGS/$M.train.log" | sed 's/.*INFO//'); cb=$(grep -h "replica checksum" "$LOGS/$M.train.rank1.log" | sed 's/.*INFO//')
  echo "$ca" | cut -c1-80
  [ -n "$ca" ] && [ "$ca" == "$cb" ] && echo "replica weights identical" || { echo "REPLICA WEIGHTS DIFFER: $cb"; rc=1; }
  [ "$a" == "$b" ] && echo "ranks agree" || { echo "RANKS DIFFER"; rc=1; }
done
exit $rc
