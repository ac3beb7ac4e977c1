# usage: ONLY=v_fc2 bash tools/pmc_quick.sh   -> MFMA busy fraction and effective clock for the GEMM kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcq; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/a -- python3 $R/tools/gemm_bench.py --iters 3 --only ${ONLY:-v_fc2,v_fc1} > $O/a.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
dur={}
for f in glob.glob(R+"/gpurun_out/pmcq/a/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+"/gpurun_out/pmcq/a/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            k=(r["Grid_Size"], r["Kernel_Name"][:36]); agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); agg[k]["_us"].append(dur.get(r["Dispatch_Id"],0))
for k,v in agg.items():
    m={c:sum(x)/len(x) for c,x in v.items()}
    cyc=m["GRBM_GUI_ACTIVE"]/8
    print(k, "us=%.1f clk=%.2fGHz mfma_busy=%.1f%% wait_any=%.1f%% wait_inst=%.1f%% active=%.1f%%" % (m["_us"], cyc/m["_us"]/1e3, 100*m["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/cyc, 100*m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"], 100*m["SQ_WAIT_INST_ANY"]/m["SQ_WAVE_CYCLES"], 100*m["SQ_ACTIVE_INST_ANY"]/m["SQ_WAVE_CYCLES"]))
PY
