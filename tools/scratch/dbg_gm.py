import numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from flatquant_amd import ops
F,T,R16,P=2,4,8,1
for dtype in ("f16","bf16"):
    td = torch.bfloat16 if dtype=="bf16" else torch.float16
    M,N=32,64
    rng=np.random.default_rng(M*N)
    rows,G=700,40
    x=torch.from_numpy((rng.standard_normal((rows,M*N))).astype(np.float32)).to(td).cuda()
    Lg=torch.from_numpy((rng.standard_normal((G,M,M))/np.sqrt(M)).astype(np.float32)).to(td).cuda()
    Rg=torch.from_numpy((rng.standard_normal((G,N,N))/np.sqrt(N)).astype(np.float32)).to(td).cuda()
    cuts=np.sort(rng.integers(0,rows+1,size=G-1)); offs=np.concatenate([[0],cuts,[rows]]).astype(np.int64)
    smax=torch.full((G,),0.9,device="cuda"); smin=torch.full((G,),0.8,device="cuda")
    o=ops.kron_quant_grouped(x,Lg,Rg,torch.from_numpy(offs).cuda(),smax,smin,F|T|R16)
    bad=[]
    for g in range(G):
        a,b=int(offs[g]),int(offs[g+1])
        if b==a: continue
        one=ops.kron_quant(x[a:b],Lg[g].contiguous(),Rg[g].contiguous(),[(0.9,0.8)],F|T|R16)
        d=(o.y[a:b].view(torch.int16)!=one.y.view(torch.int16))
        if d.any(): bad.append((g,a,b,int(d.sum()), d.any(dim=1).nonzero().flatten().tolist()[:6]))
    print(dtype, "bad groups", bad[:10])
# ratio
from flatquant_amd import deploy
g = torch.Generator().manual_seed(77)
x = (torch.randn(600, 4096, generator=g)*3).half().cuda(); x[5]=0
for ratio in (0.9,1.0):
  for shaped in (x, x.reshape(2,300,4096)):
    qz=deploy.nn.Quantizer(input_clip_ratio=ratio).cuda(); p=qz(shaped)
    want=(torch.max(torch.abs(shaped),dim=-1)[0].unsqueeze(1)/7).to(torch.float16)*ratio
    print(ratio, p.scales_x.shape, want.shape, int((p.scales_x.reshape(-1)!=want.reshape(-1)).sum()), p.scales_x.reshape(-1)[5].item(), want.reshape(-1)[5].item())
