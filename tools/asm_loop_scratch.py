"""Scratch (spill) instructions of gemm_wide_kernel variants by LOOP DEPTH, from the -save-temps .s file (depth 2 = the K loop):
   hipcc ... -save-temps -c gemm_wide.hip; python tools/asm_loop_scratch.py gemm_wide-hip-amdgcn-amd-amdhsa-gfx950.s [DT OK MF]"""
import re, sys
L = open(sys.argv[1]).read().split('\n')
want = tuple(sys.argv[2:5])
i = 0
while i < len(L):
    m = re.match(r'_ZN3cmh16gemm_wide_kernelILi(\d)ELi(\d)ELi(\d)ELb(\d)ELb(\d)(?:ELb(\d))?', L[i])
    if not m:
        i += 1
        continue
    j = next(k for k in range(i + 1, len(L)) if L[k].startswith('.Lfunc_end'))
    key = m.groups()
    depth, by_depth, mf_by_depth = 0, {}, {}
    for l in L[i:j]:
        if re.match(r'\.LBB\d+_\d+:', l):
            d = re.search(r'Depth=(\d+)', l)
            depth = int(d.group(1)) if d else 0
        elif 'Loop Header' in l or 'in Loop' in l or 'Inner Loop' in l:       # continuation comment lines of a block label
            d = re.search(r'Depth=(\d+)', l)
            if d: depth = max(depth, int(d.group(1)))
        if 'scratch_' in l: by_depth[depth] = by_depth.get(depth, 0) + 1
        if 'v_mfma' in l: mf_by_depth[depth] = mf_by_depth.get(depth, 0) + 1
    if not want or key[:3] == want:
        print('DT OK MF TN GRP =', ' '.join(key), '| scratch ops by loop depth', dict(sorted(by_depth.items())), '| MFMAs by depth', dict(sorted(mf_by_depth.items())))
    i = j
