# PMC counters for the GEMM kernel (separate passes; rocprofv3 --pmc only, no tracing domains)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc/$tag -- python3 $R/tools/gemm_bench.py --iters 3 --only ${ONLY:-v_qkv,v_fc2,t_fc1} > $R/gpurun_out/pmc/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+"/gpurun_out/pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            agg[(r["Grid_Size"], r["Kernel_Name"][:40])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:34s} n={len(vals):3d} mean={sum(vals)/len(vals):16.1f}")
PY
