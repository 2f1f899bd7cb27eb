"""Vendor int8 GEMM (torch._int_mm -> hipBLASLt) on the same shapes, as a practical ceiling for fq_gemm_i4 (GPU box)."""
import torch
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K in ((16384, 4096, 4096), (16384, 4096, 14336), (16384, 8192, 8192)):
    for name, scale in (("multiples of 16 (as fq_gemm_i4 feeds)", 16), ("full-range int8", 1)):
        hi = 8 if scale == 16 else 128
        a = (torch.randint(-hi, hi, (M, K), generator=g, device="cuda", dtype=torch.int32) * scale).clamp(-128, 127).to(torch.int8)
        b = (torch.randint(-hi, hi, (K, N), generator=g, device="cuda", dtype=torch.int32) * scale).clamp(-128, 127).to(torch.int8)
        try:
            for _ in range(3):
                torch._int_mm(a, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                torch._int_mm(a, b)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            print(f"torch._int_mm M={M} N={N} K={K} {name}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TOP/s")
        except Exception as ex:  # noqa: BLE001
            print("torch._int_mm failed:", type(ex).__name__, str(ex)[:120])
            break
