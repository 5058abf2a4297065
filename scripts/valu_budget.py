#!/usr/bin/env python3
"""Static VALU budget of the tiled likelihood kernel's particle loop: the ISA of the loop body (hipcc --save-temps) priced
with the per-opcode costs of profiles/r02b_valu_microbench.txt (cycles one wave64 instruction occupies a SIMD, two
wavefronts per SIMD). Counts every instruction of the loop once except the overflow sub-loop (rare on lattice maps) and
the two branches of the record-address computation (only the 32-bit one is counted).
usage: valu_budget.py <file.s> [kernel-name-regex]"""
import collections
import re
import sys

FULL = 2.6    # v_mul/add/sub_f32 (also with a DPP operand), v_mov_b32, v_and/or_b32, v_ashrrev_i32, v_add_u32
HALF = 4.4    # everything else measured: min/max, fma, cvt, floor, cmp, shifts left, lshl_or, add3, mul_u24, 64-bit ops
TRANS = 8.4   # v_sqrt_f32, v_rcp_f32
full_ops = ("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_ashrrev_i32",
            "v_add_u32", "v_sub_u32", "v_lshrrev_b32")
trans_ops = ("v_sqrt_f32", "v_rcp_f32", "v_rsq_f32")

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else r"_ZN6mcl3dl23likelihood_tiled_kernelILi16ELi2ELi8ELb1E\w*"
s = open(path).read()
m = re.search(r"^(%s):" % pat, s, re.M)
body = s[m.start():s.index(".Lfunc_end", m.start())].split("\n")
i0 = next(i for i, l in enumerate(body) if "This Loop Header: Depth=1" in l)
i1 = next(i for i, l in enumerate(body) if "._crit_edge" in l and i > i0)
ops = collections.Counter()
skip = False
for l in body[i0:i1]:
    t = l.strip()
    if "Inner Loop Header" in t or "Child Loop" in t:
        pass
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        ops[op] += 1
n = sum(ops.values())
cyc = 0.0
by = collections.Counter()
for op, c in ops.items():
    base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "")
    cost = TRANS if base in trans_ops else FULL if base in full_ops else HALF
    cyc += cost * c
    by["trans" if cost == TRANS else "full" if cost == FULL else "half"] += c
print("VALU instructions in the loop body (static): %d  = %s" % (n, dict(by)))
print("priced: %.0f SIMD-cycles per evaluation-wavefront (upper bound of the common path: both address branches and the"
      " overflow loop are in the count)" % cyc)
for op, c in sorted(ops.items(), key=lambda kv: -kv[1])[:14]:
    print("   %-24s %d" % (op, c))
