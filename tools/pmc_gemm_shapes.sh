# Per-shape PMC counters of gemm_wide_kernel with the encoder's real epilogues and the packed text row count: one rocprofv3 --pmc
# pass per (shape, counter set) - no tracing domains - so that every row belongs to a known shape.  Derived per shape:
#   MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs = 1024) / (GRBM_GUI_ACTIVE / 8 XCDs)      (busy cycles of the matrix pipes / launch cycles)
#   LDS read  = SQ_LDS_IDX_ACTIVE / 256 CUs / (GRBM_GUI_ACTIVE / 8)                          (LDS-array cycles of ds_read instructions)
#   L2 hit    = TCC_HIT / TCC_REQ;   waits as a share of SQ_WAVE_CYCLES
# usage (GPU box): bash tools/pmc_gemm_shapes.sh > gpurun_out/pmc_shapes.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcs; rm -rf $O; mkdir -p $O
for shape in v_qkv v_out v_fc1 v_fc2 t_qkv t_out t_fc1 t_fc2; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $O/$shape/$i -- python3 $R/tools/pmc_shape.py $shape 30 > $O/$shape.$i.log 2>&1 || echo "pass $shape/$i failed"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
print("# tools/pmc_gemm_shapes.sh: medians per launch over 30 launches per shape (counters are sums over the chip)")
for shape in "v_qkv v_out v_fc1 v_fc2 t_qkv t_out t_fc1 t_fc2".split():
    c = collections.defaultdict(list)
    for f in glob.glob(f"{R}/gpurun_out/pmcs/{shape}/*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "gemm_wide_kernel" in r["Kernel_Name"]:
                c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sorted(v)[len(v) // 2] for k, v in c.items()}
    if "GRBM_GUI_ACTIVE" not in m:
        print(shape, "no data"); continue
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(f"{shape}: launch {cyc:9.0f} cycles | MFMA busy {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:5.1f} % | LDS read {100 * m['SQ_LDS_IDX_ACTIVE'] / 256 / cyc:5.1f} % "
          f"| bank conflicts {m['SQ_LDS_BANK_CONFLICT']:.0f} | L2 hit {100 * m['TCC_HIT_sum'] / max(m['TCC_REQ_sum'], 1):5.1f} % "
          f"| of wave cycles: waiting (s_waitcnt / barrier) {100 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:5.1f} %, issue stalls {100 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:5.1f} %, "
          f"issuing {100 * m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:5.1f} %")
    print("     " + "  ".join(f"{k}={v:.0f}" for k, v in sorted(m.items())))
PY
