# Round-end evidence in one GPU call: the plain bench line, the serialized kernel trace of the same command, and kernel traces of
# the input-pipeline and training-step tools.  usage (GPU box): bash tools/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd $R && python3 bench.py > $O/bench_line.json 2> $O/bench_line.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --steps 10 --warmup 3 --repeats 1 --towers pair --no-towers-ab --no-cpu-baseline --no-train-step --no-dense-text --no-input-pipeline --no-precision-legs --no-config-legs > $O/bench_serialized_under_rocprof.json 2> $O/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prep -- python3 $R/tools/prep_bench.py --cpu-sample 8 > $O/prep_bench.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/tools/train_bench.py > $O/train_bench_under_rocprof.txt 2>&1
cd $R && python3 tools/train_bench.py > $O/train_bench.txt 2>&1
for d in bench prep train; do cp $(ls $O/$d/*/*kernel_stats.csv | head -1) $O/${d}_kernel_stats.csv; rm -rf $O/$d; done
tail -c 600 $O/bench_line.json; tail -1 $O/train_bench.txt; tail -3 $O/prep_bench.txt
