"""Launch the attention kernels at the PanoSwin-T stage-0 shape (B = 8, bf16) a few times: target for rocprofv3 --pmc."""
import sys, ctypes, torch
sys.path.insert(0, ".")
from panoswintransformerobjectdetection_amd import ops, _lib
dev = "cuda:0"
lib = _lib.load()
H, W, heads, B = 128, 256, 3, 8
C = heads * 32
tiles = ops.window_dist_tiles(H, W, 3, dev)
nW = tiles.n
n = B * nW
qkv = torch.randn(n * 49, 3 * C, device=dev).to(torch.bfloat16)
alpha = torch.randn(169, heads, device=dev) * 0.02
beta = torch.randn(169, heads, device=dev) * 0.02
out = torch.empty(n * 49, C, device=dev, dtype=torch.bfloat16)
lse = torch.empty(n, heads, 64, device=dev)
dout = torch.randn(n * 49, C, device=dev).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
vp = ctypes.c_void_p
st = vp(torch.cuda.current_stream().cuda_stream)
es = 2
q, k, v = qkv.data_ptr(), qkv.data_ptr() + C * es, qkv.data_ptr() + 2 * C * es
dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + C * es, dqkv.data_ptr() + 2 * C * es
chf = lib.pswin_attn_suggest_chunks(n, nW, heads, 0)
chb = lib.pswin_attn_suggest_chunks(n, nW, heads, 1)
gs = torch.empty(chb * nW, heads, 64, 64, device=dev)
flush = torch.empty(512 * 1024 * 1024 // 4, device=dev)     # 512 MiB: evict the 256 MiB Infinity Cache between launches
for it in range(4):
    flush.fill_(float(it))
    assert lib.pswin_attn_fwd(vp(q), vp(k), vp(v), 3 * C, vp(tiles.fwd.data_ptr()), nW, vp(alpha.data_ptr()), vp(beta.data_ptr()), None, 0,
                              vp(out.data_ptr()), C, vp(lse.data_ptr()), 0, n, nW, heads, 32 ** -0.5, 1, st) == 0
    flush.fill_(float(it) + 0.5)
    assert lib.pswin_attn_bwd(vp(q), vp(k), vp(v), 3 * C, vp(tiles.bwd.data_ptr()), nW, vp(alpha.data_ptr()), vp(beta.data_ptr()), None, 0,
                              vp(dout.data_ptr()), C, vp(lse.data_ptr()), vp(dq), vp(dk), vp(dv), 3 * C, vp(gs.data_ptr()),
                              chb, n, nW, heads, 32 ** -0.5, 1, st) == 0
torch.cuda.synchronize()
print("algorithmic bytes: fwd", n * heads * 4 * 49 * 32 * 2, "bwd", n * heads * 7 * 49 * 32 * 2, "chunks", chf, chb)
