"""instruction mix of kernels in the device assembly:
   hipcc -S --offload-arch=gfx950 --cuda-device-only -O3 -std=c++17 -o /tmp/dev.s pysdc_amd/csrc/sdcmi.hip
   python scripts/isa_mix.py /tmp/dev.s <mangled name> ..."""
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    i = txt.index(name + ':')
    j = txt.index('s_endpgm', i)
    ins = []
    for l in txt[i:j].split('\n'):
        t = l.strip()
        if not l.startswith('\t') or not t or t.startswith(('.', ';')):
            continue
        ins.append(t.split()[0])
    c = Counter(ins)
    grp = lambda pre: sum(v for k, v in c.items() if k.startswith(pre))   # noqa: E731
    print(name[:60], 'instructions', len(ins), 'f64', sum(v for k, v in c.items() if 'f64' in k), 'valu', grp('v_'), 'salu', grp('s_'),
          'global', grp('global_'), 'ds', grp('ds_'), 'scratch', grp('scratch_'), 'waitcnt', c.get('s_waitcnt', 0),
          'branches', sum(v for k, v in c.items() if k.startswith(('s_cbranch', 's_branch'))))
    print('   top:', c.most_common(12))
