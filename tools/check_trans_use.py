#!/usr/bin/env python3
"""gfx940+ VALU-trans-use hazard at the edge of an inline-asm statement: tools/check_trans_use.py file.s [kernel name substring]
The result of a transcendental VALU instruction (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos, f32 and f16) may not be read by the
VALU instruction issued right after it (one wait state). hipcc pads the hazard between instructions it schedules itself, but it does not look
inside an asm statement: a v_exp_f32 the compiler emitted followed directly by an inline-asm v_fma_mix_f32 that multiplies by its result reads
a stale register now and then (fq_kv_decode_kernel's fp16 path, round 6: six NaNs in 4096 outputs). This walks a `hipcc -S --cuda-device-only`
listing and reports every transcendental whose destination is a source of the very next instruction (an s_nop in between clears it).
Exit status 1 if any is found."""
import re
import sys

TRANS = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f32|f16|legacy_f32)")


def regs(tok):
    """registers a textual operand names: v7, v[4:7], -v3, |v2| -> {7}, {4..7}, ..."""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", tok))
    return out


def main():
    src = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2] if len(sys.argv) > 2 else None
    kernel, bad, pending = None, [], None     # pending: (line number, text, destination registers) of a transcendental just issued
    for n, raw in enumerate(src, 1):
        line = raw.split(";")[0].strip()
        if raw.startswith("_Z") and raw.rstrip().split(":")[0].startswith("_Z") and ":" in raw:
            kernel, pending = raw.split(":")[0], None
            continue
        if not line or line.startswith(".") or line.endswith(":") or line.startswith(("#", "//")):
            continue
        if key and (kernel is None or key not in kernel):
            continue
        op = line.split()[0]
        ops = line[len(op):].split(",")
        if pending is not None:
            if op.startswith("v_"):
                srcs = set()
                for t in ops[1:]:
                    srcs |= regs(t)
                if "mfma" in op or op.startswith(("v_fmac", "v_mac", "v_dot")):      # (accumulating forms read their destination too)
                    srcs |= regs(ops[0])
                if srcs & pending[2]:
                    bad.append((kernel, pending[0], pending[1], n, line))
            pending = None                                                           # any instruction (an s_nop too) is the wait state
        if TRANS.match(op):
            pending = (n, line, regs(ops[0]))
    for k, n0, l0, n1, l1 in bad:
        print(f"{k[:80]}: line {n0}: {l0}   ->   line {n1}: {l1}")
    print(f"{len(bad)} transcendental result(s) read by the next VALU instruction")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
