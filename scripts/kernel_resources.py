"""VGPRs / scratch / occupancy of selected kernels from `hipcc -Rpass-analysis=kernel-resource-usage` output:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -o /tmp/x.o pysdc_amd/csrc/sdcmi.hip -Rpass-analysis=kernel-resource-usage 2> /tmp/res.txt
   python scripts/kernel_resources.py /tmp/res.txt k_spec_zILi1024ELi5 k_fftx_invILi1024"""
import re
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
KEYS = (('VGPR', 'VGPRs'), ('AGPR', 'AGPRs'), ('SGPR', 'SGPRs'), ('scratch', r'ScratchSize \[bytes/lane\]'),
        ('occ', r'Occupancy \[waves/SIMD\]'), ('LDS', r'LDS Size \[bytes/block\]'))
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split(' ')[0]
    if pats and not any(p in name for p in pats):
        continue
    vals = []
    for label, k in KEYS:
        m = re.search(k + r': (\d+)', b)
        vals.append(f'{label} {m.group(1) if m else "?":>4}')
    print(f'{name[:100]:100s} ' + ' '.join(vals))
