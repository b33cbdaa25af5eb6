#!/usr/bin/env python3
"""SGPR-spill reloads (v_readlane) inside the innermost loops of a kernel of build/asm/crt_kernels.s (make asm):
    python tools/loop_spills.py [mangled-name substring]"""
import collections
import re
import sys

name = sys.argv[1] if len(sys.argv) > 1 else "shade_tiles_kernelILb1ELi6ELi8ELb0ELb0ELb1E"
s = open(sys.argv[2] if len(sys.argv) > 2 else "build/asm/crt_kernels.s").read()
i = s.index(name)
i = s.index("\n", s.index(name + s[i + len(name):].split(":")[0] + ":", i))
j = s.index(".Lfunc_end", i)
blocks, pos = [], {}
for l in s[i:j].split("\n"):
    l = l.strip()
    m = re.match(r"(\.LBB\d+_\d+):", l)
    if m:
        pos[m.group(1)] = len(blocks)
        blocks.append([m.group(1), []])
    elif l and not l.startswith((".", ";", "//")) and blocks:
        blocks[-1][1].append(l)
loops = set()
for bi, (_, ins) in enumerate(blocks):
    for l in ins:
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in pos and pos[m.group(1)] <= bi:
            loops.add((pos[m.group(1)], bi))
total = sum(len(b[1]) for b in blocks)
rl_all = sum(1 for b in blocks for l in b[1] if l.startswith("v_readlane"))
print("%s: %d instructions, %d v_readlane, %d loops" % (name, total, rl_all, len(loops)))
inner = [lp for lp in loops if not any(o != lp and o[0] >= lp[0] and o[1] <= lp[1] for o in loops)]
for a, b in sorted(inner):
    ins = [l for k in range(a, b + 1) for l in blocks[k][1]]
    c = collections.Counter(x.split()[0] for x in ins)
    if c["v_readlane_b32"] or c["v_writelane_b32"]:
        print("  innermost loop %s..%s: %d instructions, v_readlane %d, v_writelane %d" % (blocks[a][0], blocks[b][0], len(ins), c["v_readlane_b32"], c["v_writelane_b32"]))
in_loops = set(k for a, b in loops for k in range(a, b + 1))
print("  v_readlane inside any loop: %d" % sum(1 for k in in_loops for l in blocks[k][1] if l.startswith("v_readlane")))
