#!/usr/bin/env python3
"""Static VALU instruction mix of one kernel from the ISA (`make -C cudaraytracer_amd/csrc asm` -> build/asm/crt_kernels.s):
    python tools/isa_mix.py [kernel-name-substring] [asm-file]
Classes follow the issue rates measured on gfx950 (profiles/r1_ubench_valu*.txt, r2_ubench_valu3/4.txt):
  fast  ~2.3 cycles per wave64 instruction: VGPR-only v_mul / v_add / v_sub / v_xor / v_and / v_or / v_mov (VOP1/VOP2)
  fma   ~2.9-3.6: v_fma_f32, v_fmac_f32
  slow  ~4.1: any VALU form with an SGPR or literal operand, v_cmp*, v_cndmask, conversions, v_min / v_max, shifts, VOP3 integer forms
  trans ~8.2: v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos
The mix is STATIC (every instruction counted once; 'inside loops' = between a label and a branch back to it); the
dynamic truth is the counter ratio SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU (tools/summarise_pmc.py: valu_cycles_per_instruction)."""
import re
import sys

name = sys.argv[1] if len(sys.argv) > 1 else "shade_tiles_kernelILb1ELi6ELi8ELb0ELb0ELb1"
path = sys.argv[2] if len(sys.argv) > 2 else "build/asm/crt_kernels.s"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and name in l and l.split()[0].endswith(":"))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
depth = [0] * len(body)
for i, l in enumerate(body):           # a branch back to an earlier label: everything in between is inside one more loop
    m = re.match(r"\s+s_cbranch\w*\s+(\.LBB\d+_\d+)|\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        tgt = labels.get(m.group(1) or m.group(2))
        if tgt is not None and tgt < i:
            for k in range(tgt, i + 1):
                depth[k] += 1
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")
FAST = ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_mov_b32", "v_add_u32", "v_sub_u32",
        "v_mul_legacy_f32")
counts = {}
by_depth = {}
for i, l in enumerate(body):
    m = re.match(r"\s+(v_[a-z0-9_]+)\s*(.*)", l)
    if not m:
        continue
    op, args = m.group(1), m.group(2).split(";")[0]
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_accvgpr", "v_nop")):
        cls = "lane/acc"
    elif op.startswith(TRANS):
        cls = "trans"
    elif op.startswith(("v_fma_", "v_fmac_", "v_pk_fma")):
        cls = "fma"
    else:
        operands = [a.strip() for a in args.split(",")]
        srcs = operands[1:]
        sgpr_or_lit = any(re.match(r"^-?\|?(s\d+|s\[|vcc|exec|0x|[0-9.]+(e[-+]?\d+)?$|-?[0-9]+$)", s) and not re.match(r"^-?(0|1|2|4|0\.5|1\.0|2\.0|4\.0)$", s)
                          for s in srcs)
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        cls = "fast" if (base in FAST and not sgpr_or_lit and not op.endswith("_e64")) else "slow"
    w = depth[i]
    counts[cls] = counts.get(cls, 0) + 1
    by_depth.setdefault(w, {}).setdefault(cls, 0)
    by_depth[w][cls] += 1
total = sum(v for k, v in counts.items() if k != "lane/acc")
print("kernel %s: %d lines, %d VALU instructions (+%d lane moves)" % (body[0][:80], len(body), total, counts.get("lane/acc", 0)))
CYC = {"fast": 2.25, "fma": 3.2, "slow": 4.1, "trans": 8.2}
in_loops = {}
for d, cs in by_depth.items():
    if d >= 1:
        for k, v in cs.items():
            in_loops[k] = in_loops.get(k, 0) + v
for label, cs in [("all", counts), ("inside loops", in_loops)]:
    n = sum(v for k, v in cs.items() if k in CYC)
    if n == 0:
        continue
    cyc = sum(CYC[k] * v for k, v in cs.items() if k in CYC)
    print("  %-13s n %6d  fast %5.1f%%  fma %5.1f%%  slow %5.1f%%  trans %4.1f%%   mean issue %.2f cycles; mix bound max(2.2 N, 4.1 N_slow) = %.2f cycles per instruction"
          % (label, n, 100.0 * cs.get("fast", 0) / n, 100.0 * cs.get("fma", 0) / n, 100.0 * cs.get("slow", 0) / n, 100.0 * cs.get("trans", 0) / n,
             cyc / n, max(2.2 * n, 4.1 * (cs.get("slow", 0) + cs.get("trans", 0) * 2 + cs.get("fma", 0) * 0.8)) / n))
