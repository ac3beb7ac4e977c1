# MFMA-pipe utilisation of the multi-problem wgrad launch inside the training step: one rocprofv3 --pmc pass over tools/train_bench.py.
# usage (GPU box): bash tools/pmc_wgrad_multi.sh > gpurun_out/pmc_wgrad_multi.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcw; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/a -- python3 $R/tools/train_bench.py --steps 3 > $O/a.log 2>&1 || echo "pmc pass failed"
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
disp = collections.defaultdict(dict)
for f in glob.glob(f"{R}/gpurun_out/pmcw/a/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_wide" in r["Kernel_Name"]:
            d = disp[int(r["Dispatch_Id"])]
            d["name"] = "tn_multi" if "tn_multi" in r["Kernel_Name"] else ("tn single" if ", true, false, false>" in r["Kernel_Name"] else "nt")
            d["grid"] = int(r["Grid_Size"]) // 512
            d[r["Counter_Name"]] = float(r["Counter_Value"])
groups = collections.defaultdict(list)
for d in disp.values():
    if "GRBM_GUI_ACTIVE" in d:
        groups[(d["name"], d["grid"] if d["name"] == "tn_multi" else 0)].append(d)
print("# tools/pmc_wgrad_multi.sh: gemm_wide dispatches of tools/train_bench.py (3 steps), medians per launch")
for (name, grid), ds in sorted(groups.items()):
    med = lambda k: sorted(x[k] for x in ds)[len(ds) // 2]
    cyc = med("GRBM_GUI_ACTIVE") / 8
    print(f"{name:10s} workgroups {grid:4d}  n={len(ds):4d}  launch {cyc:8.0f} cycles  MFMA busy {100 * med('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / cyc:5.1f} % of the chip's pipes"
          f"  waves waiting {100 * med('SQ_WAIT_ANY') / med('SQ_WAVE_CYCLES'):5.1f} %")
PY
