# MFMA-pipe utilisation of the GROUPED launches (both towers' tiles in one persistent grid, csrc/gemm_wide.hip GRP) inside the real
# encode step: one rocprofv3 --pmc pass over `bench.py --towers pair` (no tracing domains), every gemm_wide_kernel dispatch classified
# by its instantiation and length:
#   <1,1,5,false,true>  = QKV (shorter) and c_fc (longer) of a layer, both towers;  <1,1,4,false,true> = out_proj, both towers;
#   <...,false,false>   = the ungrouped launches (c_proj of either tower, patch embedding, pooled tail).
# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), as tools/pmc_gemm_shapes.sh.
# usage (GPU box): bash tools/pmc_grouped.sh > gpurun_out/pmc_grouped.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcg; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/a -- python3 $R/bench.py --steps 4 --warmup 2 --repeats 1 --towers pair --no-towers-ab --no-cpu-baseline --no-map-eval --no-train-step --no-dense-text --no-input-pipeline --no-precision-legs --no-config-legs > $O/a.log 2>&1 || echo "pmc pass failed"
python3 - <<'PY'
import csv, glob, collections, os, re
R = os.environ["GRAFT_REPO_ROOT"]
disp = collections.defaultdict(dict)
for f in glob.glob(f"{R}/gpurun_out/pmcg/a/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_wide_kernel" in r["Kernel_Name"]:
            d = disp[int(r["Dispatch_Id"])]
            d["name"] = re.search(r"gemm_wide_kernel<([^>]*)>", r["Kernel_Name"]).group(1).replace(" ", "")
            d[r["Counter_Name"]] = float(r["Counter_Value"])
groups = collections.defaultdict(list)
for i in sorted(disp):
    d = disp[i]
    if "GRBM_GUI_ACTIVE" in d:
        groups[d["name"]].append(d)
print("# tools/pmc_grouped.sh: gemm_wide_kernel dispatches of `bench.py --towers pair`, per instantiation (and per length cluster)")
def report(tag, ds):
    if not ds:
        return
    med = lambda k: sorted(x[k] for x in ds)[len(ds) // 2]
    cyc = med("GRBM_GUI_ACTIVE") / 8
    print(f"{tag:46s} n={len(ds):4d}  launch {cyc:8.0f} cycles  MFMA busy {100 * med('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / cyc:5.1f} %  "
          f"waves waiting {100 * med('SQ_WAIT_ANY') / med('SQ_WAVE_CYCLES'):5.1f} % of wave cycles")
for name, ds in sorted(groups.items()):
    if name.endswith("true") and name.startswith("1,1,5"):
        cyc = sorted(x["GRBM_GUI_ACTIVE"] for x in ds)
        cut = (cyc[len(cyc) // 4] + cyc[3 * len(cyc) // 4]) / 2        # two clusters of equal size: QKV and c_fc
        report(f"<{name}> QKV, both towers", [x for x in ds if x["GRBM_GUI_ACTIVE"] < cut])
        report(f"<{name}> c_fc (+QuickGELU), both towers", [x for x in ds if x["GRBM_GUI_ACTIVE"] >= cut])
    elif name.endswith("true"):
        report(f"<{name}> out_proj, both towers", ds)
    else:
        report(f"<{name}> ungrouped", ds)
PY
