# HBM-side traffic of the GEMM kernel inside bench.py: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; no tracing
# domains), averaged per launch over every gemm_wide_kernel dispatch.  Corrections per MI355X_MICROARCH.md (HBM section): both
# counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of 16-B/lane streams at 64 B -> doubled.
# usage (GPU box): bash tools/pmc_bench_traffic.sh  -> gpurun_out/gemm_traffic.json (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmct; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $R/bench.py --steps 3 --warmup 1 --repeats 1 --towers pair --no-towers-ab --no-cpu-baseline --no-map-eval --no-train-step --no-dense-text --no-input-pipeline --no-precision-legs --no-config-legs > $O/$c.log 2>&1 || echo "pass $c failed"
done
python3 - <<'PY'
import csv, glob, json, os
R = os.environ["GRAFT_REPO_ROOT"]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(f"{R}/gpurun_out/pmct/{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "gemm_wide_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    tot[c] = (sum(vals) / max(len(vals), 1), len(vals))
fetch_kib, n = tot["FETCH_SIZE"]; write_kib, _ = tot["WRITE_SIZE"]
out = {"kernel": "cmh::gemm_wide_kernel", "launches": n, "FETCH_SIZE_KiB_per_launch": round(fetch_kib, 1),
       "WRITE_SIZE_KiB_per_launch": round(write_kib, 1),
       "traffic_bytes_per_launch": round((2 * fetch_kib + write_kib) * 1024),
       "correction": "(2*FETCH_SIZE + WRITE_SIZE) * 1024: counters in KiB, gfx950 FETCH_SIZE counts wide reads at 1/2",
       "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --towers pair --no-towers-ab --no-cpu-baseline --no-map-eval --no-train-step --no-dense-text --no-input-pipeline --no-precision-legs --no-config-legs"}
json.dump(out, open(f"{R}/gpurun_out/gemm_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
