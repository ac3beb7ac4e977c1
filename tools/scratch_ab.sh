cd /tmp && export TMPDIR=/tmp
for l in build build_old build build_old; do
  CMH_LIB=$GRAFT_REPO_ROOT/clip-based-cross-modal-hashing_amd/csrc/$l/libcmh.so rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab_$l -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py --steps 4 > /dev/null 2>&1
  echo $l; python3 - $GRAFT_REPO_ROOT/gpurun_out/ab_$l <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "attention_bwd" in r["Name"]:
            print("  ", r["Name"][:44], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
PY
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/ab_$l
done
