# Kernel trace of one bench leg: bash tools/leg_trace.sh <leg> <tag>  -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>.txt
set -e
R=$GRAFT_REPO_ROOT; LEG=$1; TAG=${2:-leg}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG.d -- python3 $R/tools/run_leg.py $LEG > $R/gpurun_out/$TAG.txt 2>&1
cp $(ls $R/gpurun_out/$TAG.d/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv; rm -rf $R/gpurun_out/$TAG.d
tail -1 $R/gpurun_out/$TAG.txt | cut -c1-600
