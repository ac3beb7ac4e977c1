#!/bin/bash
# Host tokenizer under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: the pool has no GPU sanitizers).
# The tokenizer source is compiled as HIP host code only (--cuda-host-only): it contains no kernels.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(dirname "$HERE")
OUT=${TMPDIR:-/tmp}/cmh_asan_bpe
/opt/rocm/bin/hipcc -std=c++17 -O1 -g --cuda-host-only -fsanitize=address,undefined -fno-omit-frame-pointer \
  "$ROOT/clip-based-cross-modal-hashing_amd/csrc/bpe_tokenizer.hip" -x hip "$HERE/asan_bpe_main.cpp" -lpthread -o "$OUT"
ASAN_OPTIONS=detect_leaks=1 "$OUT" "$1"
