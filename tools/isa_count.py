#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing (used for the per-element VALU counts quoted in DESIGN.md).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S --cuda-device-only -o x.s llm-qat_amd/csrc/fq_bf16.hip
    python tools/isa_count.py x.s 'row_reg_kernel<1, 512, 3, false, true, true, true, false, 0>' [--dump out.s]
"""
import re
import subprocess
import sys
from collections import Counter


def kernel_text(s, sub):
    names = re.findall(r'^(_Z\S+):\s', s, re.M)
    d = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    for n, dn in zip(names, d):
        if sub in dn:
            a = s.index('\n' + n + ':')
            return dn, s[a:s.index('.Lfunc_end', a)]
    raise SystemExit(f'no kernel matching {sub!r}')


def main():
    s = open(sys.argv[1]).read()
    name, k = kernel_text(s, sys.argv[2])
    ins = []
    for line in k.split('\n'):
        t = line.strip()
        if line.startswith('\t') and t and not t.startswith(('.', ';')):
            ins.append(t.split()[0])
    c = Counter(ins)
    valu = sum(v for op, v in c.items() if op.startswith('v_'))
    print(name)
    print(f'{len(ins)} instructions, {valu} VALU, {sum(v for op, v in c.items() if op.startswith("s_"))} SALU/ctl, '
          f'{sum(v for op, v in c.items() if op.startswith(("global_", "buffer_", "flat_", "ds_")))} memory')
    for op, v in c.most_common(int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 30):
        print(f'  {op:28s} {v}')
    if '--dump' in sys.argv:
        open(sys.argv[sys.argv.index('--dump') + 1], 'w').write(k)


if __name__ == '__main__':
    main()
