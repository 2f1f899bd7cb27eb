#!/usr/bin/env python3
"""ISA checks for kernels that load registers with loads the compiler does not track (inline-asm global_load_dwordx4 in
fq_kron_tall.hip).

(1) HAZARDS: between such a load and the wait that covers it no instruction may READ or COPY the destination registers (the data
    has not arrived; a register copy there — a phi, a spill, an AGPR move — silently takes the old contents). Waits:
      s_waitcnt vmcnt(0)              covers everything;
      s_waitcnt vmcnt(4)  (--deep 2)  covers all but the four loads issued last (a two-register-set build);
      s_waitcnt vmcnt(k)  (--deep 1)  a counted wait behind the token's stores (k = stores issued since): covers all. Also the mode for
                                      fq_kron64_linear_kernel (round 6: two weight tiles in flight, counts chosen at run time inside one asm
                                      statement): every wait is taken to cover everything, i.e. what is checked is "no read or copy between a
                                      request and the NEXT wait" — where the first build's copies sat (the loop's back edge).
    (the two --deep forms belong to prefetch variants measured and removed in round 5; kept for the next experiment of that kind)
(2) LOOP WAITS (round 5): inside the token loop the ONLY vmcnt waits may be the kernel's own explicit ones and the vmcnt(0) directly
    behind a compiler-tracked load of a grouped launch's clip / offset arrays. Rounds 3-4 shipped a build whose compiler-inserted
    waits for the factor fragments (loaded once, in the prologue, and sunk below the prologue's wait by the scheduler) sat in front
    of GEMM 2's MFMAs: s_waitcnt vmcnt(11) ... vmcnt(0) every iteration — and vmcnt is one counter, so every token waited there
    for the rows of the NEXT token requested a few instructions earlier. Checked: no `s_waitcnt vmcnt(n)` with n > 0 in a basic
    block of the token loop (the compiler marks them "in Loop" / "Loop Header") other than the kernel's own.

    tools/check_untracked_loads.py file.s [kernel-name-substring] [--deep 0|1|2]      exit code 1 and a listing if anything is found."""
import re
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
deep = 0
if "--deep" in sys.argv:
    deep = int(sys.argv[sys.argv.index("--deep") + 1])
    args = [a for a in args if a != str(deep)] if str(deep) in args[2:] else args
src = open(args[0]).read().split("\n")
key = args[1] if len(args) > 1 else ""
bad = 0
kernel = None
pending = []      # [(register numbers, line)] in issue order
in_asm = False
loop_waits = []   # compiler waits with n > 0 inside the token loop
in_loop = False


def flush_loop_waits():
    global bad, loop_waits
    if loop_waits:
        bad += 1
        print(f"{kernel[:60]}: {len(loop_waits)} compiler vmcnt waits inside the token loop (lines {loop_waits[0] + 1}..{loop_waits[-1] + 1}): "
              "they also wait for the rows requested ahead")
    loop_waits = []


for i, l in enumerate(src):
    t = l.strip()
    if l.startswith("_Z") and l.rstrip().endswith(":") or (l.startswith("_Z") and ":" in l.split(";")[0]):
        if kernel is not None and key in kernel:
            flush_loop_waits()
        kernel = l.split(":")[0]
        pending, loop_waits, in_loop = [], [], False
    if kernel is None or key not in kernel:
        continue
    if t.startswith(".LBB"):
        in_loop = "Loop" in t
        continue
    if t.startswith("s_endpgm"):
        in_loop = False
    if t.startswith(";;#ASMSTART"):
        in_asm = True
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if not t or t.startswith((";", ".")):
        continue
    m = re.match(r"global_load_dwordx[24] v\[(\d+):(\d+)\]", t)
    if in_asm and m:
        pending.append((set(range(int(m.group(1)), int(m.group(2)) + 1)), i))
        continue
    m1 = re.match(r"global_load_(?:ushort|dword) v(\d+),", t)      # (round 6: the fused decode launch's column scales / biases)
    if in_asm and m1:
        pending.append(({int(m1.group(1))}, i))
        continue
    mw = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", t)
    if mw:
        n = int(mw.group(1))
        if n == 0 or deep == 1:
            pending = []
        elif deep == 2 and n == 4:
            pending = pending[-4:]
        if n > 0 and in_loop and not in_asm and not (deep == 1) and not (deep == 2 and n == 4):
            loop_waits.append(i)
        continue
    if pending and not t.startswith(("s_", "global_load", "ds_")):
        ops = t.split(None, 1)[1] if " " in t else ""
        srcs = ops.split(",", 1)[1] if "," in ops else ""        # everything after the destination
        if t.startswith(("global_store", "buffer_store", "flat_store")):
            srcs = ops                                            # a store reads all of its operands
        regs = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", srcs):
            regs.update(range(int(a), int(b) + 1))
        regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", srcs))
        for rs, ln in pending:
            hit = regs & rs
            if hit:
                bad += 1
                print(f"{kernel[:60]} line {i + 1}: `{t}` reads v{sorted(hit)[0]}.. loaded at line {ln + 1} before its wait")
                break
if kernel is not None and key in kernel:
    flush_loop_waits()
print("findings:", bad)
sys.exit(1 if bad else 0)
