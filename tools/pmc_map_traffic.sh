# HBM-side traffic of the ranking kernel at one dataset scale: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as in
# tools/pmc_bench_traffic.sh.  usage (GPU box): bash tools/pmc_map_traffic.sh nuswide -> gpurun_out/map_traffic_<scale>.txt
S=${1:-nuswide}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcm_$S; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $R/tools/map_bench.py $S > $O/$c.log 2>&1 || echo "pass $c failed"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/map_bench.py $S > $O/trace.log 2>&1 || echo "trace failed"
python3 - $S <<'PY'
import csv, glob, os, sys
R = os.environ["GRAFT_REPO_ROOT"]; S = sys.argv[1]
out = open(f"{R}/gpurun_out/map_traffic_{S}.txt", "w")
rows = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{R}/gpurun_out/pmcm_{S}/{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "map_query_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                rows.setdefault(c, []).append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(f"{R}/gpurun_out/pmcm_{S}/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "map_query_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
for i in range(len(rows.get("FETCH_SIZE", []))):
    fe, wr = rows["FETCH_SIZE"][i], rows["WRITE_SIZE"][i] if i < len(rows.get("WRITE_SIZE", [])) else float("nan")
    d = dur[i] if i < len(dur) else float("nan")
    line = f"launch {i}: FETCH_SIZE {fe:.0f} KiB (x2 = {2*fe/1048576:.2f} GiB)  WRITE_SIZE {wr:.0f} KiB ({wr/1048576:.2f} GiB)  {d:.2f} ms  -> {(2*fe+wr)*1024/1e9/(d*1e-3)/1e3:.2f} TB/s"
    print(line); out.write(line + "\n")
PY
