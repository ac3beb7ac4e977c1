# Which operand does gemm_wide_kernel re-fetch from beyond the L2s?  (read side 2.33x the algorithmic bytes)
# One rocprofv3 --pmc pass per shape (FETCH_SIZE alone: it takes three of the four TCC slots): a family of launches with the SAME X
# operand (12 800 x 768 bf16 = 19.7 MB) and N = 256 ... 3072 weight rows, plus the packed-text family.  Per launch
#   reads = 2 x FETCH_SIZE x 1024 B      (gfx950: FETCH_SIZE reports half of a wide coalesced read, MI355X_MICROARCH.md "HBM")
# and with X counted once, the rest is W-side traffic: (reads - X bytes) / W bytes = how many times W left the fabric.
# usage (GPU box): bash tools/pmc_traffic_split.sh > gpurun_out/traffic_split.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmct; rm -rf $O; mkdir -p $O
SHAPES="12800x256x768 12800x768x768 12800x1536x768 12800x2304x768 12800x3072x768 12800x768x3072 10499x512x512 10499x1536x512 10499x2048x512 10499x512x2048"
for s in $SHAPES; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$s/f -- python3 $R/tools/pmc_shape.py $s 30 > $O/$s.f.log 2>&1 || echo "pass $s fetch failed"
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/$s/w -- python3 $R/tools/pmc_shape.py $s 30 > $O/$s.w.log 2>&1 || echo "pass $s write failed"
done
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
print("# tools/pmc_traffic_split.sh: medians per launch over 30 back-to-back launches per shape; bf16 operands and output, bias epilogue")
print("# shape MxNxK | reads MB (2 x FETCH_SIZE KB) | X MB | W MB | (reads - X) / W | writes MB | out MB | L2 hit %")
for s in os.environ.get("SHAPES_PY", "12800x256x768 12800x768x768 12800x1536x768 12800x2304x768 12800x3072x768 12800x768x3072 10499x512x512 10499x1536x512 10499x2048x512 10499x512x2048").split():
    c = collections.defaultdict(list)
    for f in glob.glob(f"{R}/gpurun_out/pmct/{s}/*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "gemm_wide_kernel" in r["Kernel_Name"]:
                c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sorted(v)[len(v) // 2] for k, v in c.items()}
    if "FETCH_SIZE" not in m:
        print(s, "no data"); continue
    M, N, K = (int(v) for v in s.split("x"))
    reads = 2 * m["FETCH_SIZE"] * 1024 / 1e6
    X, W, out = M * K * 2 / 1e6, N * K * 2 / 1e6, M * N * 2 / 1e6
    hit = 100 * m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)
    print(f"{s:16s} | reads {reads:7.1f} | X {X:5.1f} | W {W:5.2f} | W-side fetches {(reads - X) / W:6.1f} x | writes {m.get('WRITE_SIZE', 0) * 1024 / 1e6:6.1f} | out {out:5.1f} | L2 hit {hit:4.1f} %")
PY
