#!/usr/bin/env python3
"""Instruction mix of one kernel in pypbr_amd/csrc/cook_torrance.gfx950.s (`make -C pypbr_amd/csrc asm`).
usage: tools/isa_mix.py LIGHT WF TI TO VEC MULTI NT   e.g.  1 0 f f 4 0 1"""
import collections
import os
import re
import sys

ASM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pypbr_amd", "csrc", "cook_torrance.gfx950.s")


def main():
    light, wf, ti, to, vec, multi, nt = sys.argv[1:8]
    tmap = {"f": "f", "h": "6__half", "H": "S1_"}
    sym = f"_ZN3pbr20cook_torrance_kernelILi{light}ELi{wf}E{tmap[ti]}{tmap[to]}Li{vec}ELb{multi}ELb{nt}EEEvNS_5KArgsE"
    s = open(ASM).read()
    a = s.index("\n" + sym + ":")
    b = s.index(".Lfunc_end", a)
    ops = collections.Counter()
    for line in s[a:b].splitlines():
        m = re.match(r"^\s+([sv]_\w+|global_\w+|buffer_\w+|ds_\w+)", line)
        if m:
            ops[m.group(1)] += 1
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    trans = sum(v for k, v in ops.items() if re.match(r"v_(rcp|rsq|sqrt|log|exp)_", k))
    print(f"{sym}: {sum(ops.values())} instructions, {valu} VALU ({trans} transcendental), "
          f"{sum(v for k, v in ops.items() if k.startswith('s_'))} SALU, "
          f"{sum(v for k, v in ops.items() if k.startswith('global_'))} global")
    for k, v in ops.most_common(int(sys.argv[8]) if len(sys.argv) > 8 else 40):
        print(f"{v:6d} {k}")


if __name__ == "__main__":
    main()
