import torch
x = torch.zeros(64, device="cuda")
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): x.add_(1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(200): x.add_(1)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)*1e3/200)
    print("dependent tiny kernels in a graph: %.2f us each" % sorted(ts)[3])
    # eager back-to-back
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2000): x.add_(1)
    e1.record(); torch.cuda.synchronize()
    print("eager stream: %.2f us each" % (e0.elapsed_time(e1)*1e3/2000))
