"""Per-dispatch durations of the few-workgroup GEMM launches (M = batch: the pooled rows of the last block, the final projections)
from a rocprofv3 kernel trace: python tools/small_gemm_trace.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if "gemm_wide" not in name and "splitk" not in name and "gather_rows" not in name:
        continue
    grid = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    if grid > 64 and "gemm_wide" in name:
        continue
    agg[(name.split("(")[0][-44:], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in sorted(agg.items()):
    v.sort()
    print(f"{n:46s} workgroups {g:4d} launches {len(v):4d} median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}  max {v[-1]:7.1f}")
