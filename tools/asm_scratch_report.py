"""Where a wide-GEMM variant touches scratch (spills) relative to its barriers and MFMAs, from the -save-temps .s file:
   cd /tmp/x && hipcc ... -save-temps -c gemm_wide.hip; python tools/asm_scratch_report.py /tmp/x/gemm_wide-hip-amdgcn-amd-amdhsa-gfx950.s 1 1 5"""
import re, sys
s = open(sys.argv[1]).read()
want = tuple(sys.argv[2:5])
for f in re.split(r'\n(?=_ZN3cmh16gemm_wide_kernel\w+:)', s):
    m = re.match(r'(_ZN3cmh16gemm_wide_kernelILi(\d)ELi(\d)ELi(\d)E\w+):', f)
    if not m:
        continue
    lines = f.split('\n')
    bar = [i for i, l in enumerate(lines) if 's_barrier' in l]
    scr = [(i, l.strip()) for i, l in enumerate(lines) if 'scratch_' in l]
    mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
    print(m.group(2), m.group(3), m.group(4), 'lines', len(lines), 'barriers', bar, 'scratch ops', len(scr), 'mfma lines', (mf[0], mf[-1]) if mf else None)
    if (m.group(2), m.group(3), m.group(4)) == want:
        for i, l in scr:
            print('    ', i, l)
