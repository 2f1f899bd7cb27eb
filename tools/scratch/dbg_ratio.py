import torch
g = torch.Generator().manual_seed(77)
x = (torch.randn(600, 4096, generator=g)*3).half().cuda()
m = torch.max(torch.abs(x), dim=-1)[0].unsqueeze(1)
s0 = (m/7).to(torch.float16)
print("s0 == fp16(m32/7):", int((s0 == (m.float()/7).half()).sum()), "s0 == fp16(m32 * fl(1/7)):", int((s0 == (m.float()*torch.tensor(1/7, dtype=torch.float32).cuda()).half()).sum()))
for ratio in (0.9, 0.83):
    want = s0 * ratio
    r32 = torch.tensor(ratio, dtype=torch.float32).cuda()
    r16 = r32.half().float()
    a = (s0.float()*r32).half()
    c = (s0.float()*r16).half()
    d = (s0.double()*ratio).half()
    e = (s0.double()*float(r32)).half()
    print(ratio, "two-round f32 ratio", int((want==a).sum()), "fp16 ratio", int((want==c).sum()), "exact double ratio", int((want==d).sum()), "exact f32 ratio", int((want==e).sum()))
    # per-element scalar mul through a 0-dim tensor
    w2 = s0 * torch.tensor(ratio).cuda()
    print("   0-dim device tensor:", int((w2==a).sum()), int((w2==c).sum()), int((w2==e).sum()))
