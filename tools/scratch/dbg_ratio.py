import torch, sys
sys.path.insert(0,'/root/repo')
from flatquant_amd import deploy
g = torch.Generator().manual_seed(77)
x = (torch.randn(600, 4096, generator=g)*3).half().cuda()
for ratio in (0.9, 0.83, 1.0):
    qz = deploy.nn.Quantizer(input_clip_ratio=ratio).cuda()
    p = qz(x)
    m = torch.max(torch.abs(x), dim=-1)[0].unsqueeze(1)
    want = (m / 7).to(torch.float16) * ratio
    ulp = (p.scales_x.view(torch.int16).int() - want.view(torch.int16).int()).abs()
    s0 = (m.float()/7).half()
    two = (s0.float()*ratio).half()      # two roundings (float32 ratio?)
    r32 = torch.tensor(ratio, dtype=torch.float32).cuda()
    two32 = (s0.float()*r32).half()
    print(ratio, int(ulp.max()), int((ulp!=0).sum()), 'ours==two32', int((p.scales_x==two32).sum()), 'want==two32', int((want==two32).sum()), 'want==double', int((want==(s0.double()*ratio).half()).sum()), 's0 eq', int(((m/7).half()==s0).sum()))
