"""what does torch-ROCm compute for an fp16 [rows,1] tensor times a 0-dim fp32 DEVICE tensor?
 (a) fp16(fp32(x) * s32)              -- the CPU result (s keeps its fp32 value: 'wrapped number' semantics)
 (b) fp16(fp32(x) * fp32(fp16(s32)))  -- s cast to the result dtype on load
Run on the GPU box; prints how many of 65536 extrema agree with each."""
import torch
x = torch.arange(1, 65536, dtype=torch.int32).to(torch.int16).view(torch.float16)
x = x[torch.isfinite(x)].reshape(-1, 1)
for c in (4.0, 2.3, -0.7, 1.0):
    s = torch.sigmoid(torch.tensor(c, dtype=torch.float32))
    gpu = (x.cuda() * s.cuda()).cpu()
    cpu = x * s
    a = (x.float() * s).half()
    b = (x.float() * s.half().float()).half()
    print(f"clip {c}: gpu dtype {gpu.dtype}; gpu==cpu {int((gpu == cpu).sum())}/{x.numel()}; gpu==(a) {int((gpu == a).sum())}; gpu==(b) {int((gpu == b).sum())}; cpu==(a) {int((cpu == a).sum())}")
# the same for bf16 rows
xb = torch.arange(1, 65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
xb = xb[torch.isfinite(xb)].reshape(-1, 1)
s = torch.sigmoid(torch.tensor(2.3, dtype=torch.float32))
gpu = (xb.cuda() * s.cuda()).cpu()
print("bf16:", gpu.dtype, int((gpu == (xb.float() * s).bfloat16()).sum()), int((gpu == (xb.float() * s.bfloat16().float()).bfloat16()).sum()), xb.numel())
