import torch
x = torch.randn(2 * 32 * 64, 96, device="cuda").to(torch.bfloat16)
big = torch.randn(8, 96, 128, 256, device="cuda")
def fb():
    y = x * 1.0
    return [y.float().mean(), big.mean(), big.sum(), y.float().sum()]
ref = [t.clone() for t in fb()]
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): fb()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = fb()
for it in range(3):
    g.replay(); torch.cuda.synchronize()
    print("replay", it, [f"{float(a):.6g}" for a in out], "ref", [f"{float(a):.6g}" for a in ref])
