#!/usr/bin/env python3
"""VALU census of a kernel's per-token loop from a `hipcc -S --cuda-device-only` listing.
    tools/valu_census.py file.s <kernel name substring> [--elems N] [--exclude LBB0_145,LBB0_262,...] [--loop LBB32_23]
Walks the basic blocks of the kernel, keeps the blocks inside its outermost loop (the token loop: everything between the
first backward-branch target and the last backward branch), drops the blocks of the rare exact-division fallback (they contain
v_div_scale_f32 next to v_rndne: taken for ~3e-5 of the dwords), and sorts every VALU instruction into a category.
Static counts of straight-line code = dynamic counts per loop trip when every kept block runs once per token (true for the
kernels of this library: their inner loops are fully unrolled); blocks that are alternatives of one another (clamp / no-clamp
quantiser, two-tile / one-tile GEMM) are listed separately so that the reader adds the ones of the route in question."""
import re
import sys
from collections import Counter, OrderedDict

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
elems = int(sys.argv[sys.argv.index("--elems") + 1]) if "--elems" in sys.argv else None
# --exclude L1,L2,...: blocks of routes NOT taken (e.g. the clamp quantiser when the census is of the no-clamp route)
exclude = set(sys.argv[sys.argv.index("--exclude") + 1].split(",")) if "--exclude" in sys.argv else set()
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])

CATS = OrderedDict([
    ("quantiser (asm blocks: fma/mad_u32_u16/lshl_add/min3_u16/med3/xor/cmp)", None),   # by ASM region
    ("extrema (max3/min3/max/min, DPP and permlane steps)", re.compile(r"v_(max3|min3|max|min)_f(32|16)|v_pk_(max|min)_f16|v_permlane|_dpp")),
    ("fp32->fp16/bf16 conversions, fp16->fp32", re.compile(r"v_cvt_")),
    ("post-scale / butterflies / other float arithmetic", re.compile(r"v_(pk_)?(mul|add|sub|fma|fmac|mad)_(f32|f16|legacy_f32)|v_fma_mix")),
    ("division, reciprocal (scale = m / 7, 1 / scale)", re.compile(r"v_div_|v_rcp_|v_rsq_|v_ldexp|v_frexp")),
    ("accumulator / register copies (v_mov, v_accvgpr)", re.compile(r"v_mov_|v_accvgpr_|v_swap")),
    ("address and index arithmetic (integer add/shift/mul/and/or/bfe/perm)", re.compile(r"v_(add|sub|subrev|lshl|lshr|ashr|mul|mad|and|or|xor|bfe|bfi|perm|alignbit|lshl_add|lshl_or|add3|and_or|bitop3|mbcnt|bcnt|not)_?[a-z0-9_]*(u32|i32|b32|u24|i24|u64|b64|u16|i16)")),
    ("compares, selects, lane reads (v_cmp, v_cndmask, v_readfirstlane/readlane/writelane)", re.compile(r"v_cmp|v_cndmask|v_readfirstlane|v_readlane|v_writelane")),
])

# blocks
blocks, cur, in_asm = [], None, False
for i in range(start + 1, end + 1):
    t = src[i].strip()
    if not t:
        continue
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if t.startswith((";", ".")) and not t.startswith(".LBB"):
        continue
    if t.split()[0].endswith(":"):
        cur = {"name": t.split()[0][:-1], "ins": [], "line": i}
        blocks.append(cur)
        continue
    if cur is None:
        cur = {"name": "entry", "ins": [], "line": i}
        blocks.append(cur)
    cur["ins"].append((t, in_asm))
names = {b["name"]: k for k, b in enumerate(blocks)}
# outermost loop = [min target of a backward branch, max index of a backward branch]
lo, hi, best_span = len(blocks), -1, 0
for k, b in enumerate(blocks):
    for t, _ in b["ins"]:
        if t.startswith(("s_cbranch", "s_branch")):
            tgt = names.get(t.split()[1])
            if tgt is not None and tgt <= k:
                span = k - tgt
                if span > best_span:      # the token loop = the widest backward branch (meeting spins and search loops are short)
                    best_span, lo, hi = span, tgt, k
if "--loop" in sys.argv:   # --loop LBB32_23: the loop whose header is that block (up to the last branch back to it)
    hdr = sys.argv[sys.argv.index("--loop") + 1]
    lo = names["." + hdr.lstrip(".")]
    hi = max(k for k, b in enumerate(blocks) for t, _ in b["ins"]
             if t.startswith(("s_cbranch", "s_branch")) and t.split()[1] == "." + hdr.lstrip(".") and k >= lo)
loop = blocks[lo:hi + 1]


def classify(op, in_asm):
    if "mfma" in op:
        return "MFMA"
    if not op.startswith("v_"):
        return None
    if in_asm and re.match(r"v_(fma_f32|mad_u32_u16|lshl_add_u32|min3_u16|min_u16|med3_[fi]32|xor_b32|add_u32|cmp_|fma_mix|cvt_pk_f16_f32|pk_add_f16|pk_max_f16|pk_min_f16|perm_b32|lshrrev_b32|bfi_b32|mul_f32|subrev_f32|max3_f32|cvt_pk_bf16)", op):
        return next(iter(CATS))
    for name, rx in CATS.items():
        if rx is not None and rx.search(op):
            return name
    return "other VALU"


tot, rare, per_block = Counter(), Counter(), []
for b in loop:
    if b["name"].lstrip(".") in exclude or b["name"] in exclude:
        continue
    ops = [t.split()[0] for t, _ in b["ins"]]
    is_rare = any(o == "v_div_scale_f32" for o in ops) and any(o == "v_rndne_f32_e32" for o in ops) and sum(o.startswith("v_div_scale") for o in ops) >= 8
    c = Counter()
    for t, a in b["ins"]:
        k = classify(t.split()[0], a)
        if k:
            c[k] += 1
    per_block.append((b["name"], is_rare, c, len(b["ins"])))
    (rare if is_rare else tot).update(c)

nvalu = sum(v for k, v in tot.items() if k != "MFMA")
print(f"kernel {key}: token loop = blocks {loop[0]['name']} .. {loop[-1]['name']} ({len(loop)} blocks), "
      f"{nvalu} VALU + {tot['MFMA']} MFMA outside the exact-division fallback ({sum(v for k, v in rare.items() if k != 'MFMA')} VALU inside it)")
for name in list(CATS) + ["other VALU"]:
    if tot[name]:
        extra = f"   {tot[name] / elems:5.2f} per lane-element" if elems else ""
        print(f"  {tot[name]:6d}  {name}{extra}")
if elems:
    print(f"  {nvalu:6d}  total   {nvalu / elems:5.2f} per lane-element ({elems} elements per lane and token)")
print("  blocks with >= 24 VALU (alternatives of one another are NOT de-duplicated: add the ones of the route in question):")
for name, is_rare, c, n in per_block:
    v = sum(x for k, x in c.items() if k != "MFMA")
    if v >= 24 or c["MFMA"]:
        top = ", ".join(f"{x} {k.split(' (')[0]}" for k, x in c.most_common(3))
        print(f"    {name:10s} {'(fallback)' if is_rare else '          '} VALU {v:4d}  MFMA {c['MFMA']:3d}   {top}")
