"""Instruction counts per opcode of a kernel, or the difference between two builds, from hipcc's assembly listing
(measurement tool).

    hipcc -O3 --offload-arch=gfx950 -DMSM_REAL=float -DMSM_SUFFIX=f32 -DMSM_TQ --cuda-device-only -S msm_amd/csrc/engine.hip -o /tmp/a.s
    python tools/isa_hist.py /tmp/a.s k5_psi_rtILi0 7k5_kick            # counts of one listing
    python tools/isa_hist.py /tmp/a.s /tmp/b.s k5_psi_rtILi0 7k5_kick   # b minus a, opcodes that moved by 8 or more

Kernels are named by a substring of their mangled name.  What to look for: address arithmetic (v_ashrrev / v_add3 /
v_lshlrev per LDS access, v_add_co + v_addc per global access), v_mov / v_xor / v_cndmask that assemble operands, s_nop
between dependent packed instructions, scratch_ accesses.
"""
import collections
import re
import sys


def load(path):
    lines = open(path).read().split('\n')
    names = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(_ZN3msm\S+):', l)
        if m:
            names[m.group(1)] = i
    return lines, names


def count(lines, names, key):
    cands = [n for n in names if key in n]
    if not cands:
        raise SystemExit(f"no kernel matches {key}")
    h = collections.Counter()
    for l in lines[names[cands[0]] + 1:]:
        t = l.strip()
        if t.startswith('.Lfunc_end'):
            break
        if not t or t.startswith(('.', ';')) or t.endswith(':'):
            continue
        h[t.split()[0]] += 1
    return cands[0], h


def main():
    args = sys.argv[1:]
    files = [a for a in args if a.endswith('.s')]
    keys = [a for a in args if not a.endswith('.s')]
    a = load(files[0])
    b = load(files[1]) if len(files) > 1 else None
    for k in keys:
        name, ha = count(*a, k)
        if b is None:
            print(name, sum(ha.values()))
            print('   ', ha.most_common(30))
        else:
            _, hb = count(*b, k)
            scratch = sum(v for op, v in hb.items() if op.startswith('scratch_'))
            print(name, sum(ha.values()), '->', sum(hb.values()), f'(scratch accesses in the second: {scratch})')
            d = {op: hb.get(op, 0) - ha.get(op, 0) for op in set(ha) | set(hb)}
            print('   ', sorted((v, op) for op, v in d.items() if abs(v) >= 8))


if __name__ == "__main__":
    main()
