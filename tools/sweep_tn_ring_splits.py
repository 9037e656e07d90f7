"""Row-split sweep of the ring weight-gradient kernel (pswin_gemm_tn_ring) + the fixed-order sum of its partial slabs, per Linear shape
of PanoSwin-T at batch 8 and batch 2: time of (kernel + reduction of its slabs) against the number of workgroups the splits are chosen
for.  Fewer splits = fewer partial-slab bytes (256 workgroups always write 256 x 192 x 192 x 2 B = 18.9 MB whatever the shape) but
idle CUs.  Ten launches per measurement are captured in one hipGraph and replayed (no host launch gaps).
usage: python tools/sweep_tn_ring_splits.py [batch ...]   ->  gpurun_out/r04_tn_ring_split_sweep.txt"""
import sys

import torch

sys.path.insert(0, ".")
from panoswintransformerobjectdetection_amd import _lib, ops

dev = "cuda:0"
lib = _lib.load()


def graph_time(fn, reps=10, replays=5):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


def shapes(batch):
    out = []
    for st, C in enumerate((96, 192, 384, 768)):
        H, W = 128 >> st, 256 >> st
        Hp, Wp, nW = _lib.window_grid(_lib.MODE_PANO, H, W)
        Mw, Mt = batch * nW * 49, batch * H * W
        out += [(f"s{st} qkv", Mw, 3 * C, C), (f"s{st} proj", Mw, C, C), (f"s{st} fc1", Mt, 4 * C, C), (f"s{st} fc2", Mt, C, 4 * C)]
        if st < 3:
            out.append((f"m{st + 1} red", Mt // 4, 2 * C, 4 * C))
    return out


def main():
    batches = [int(a) for a in sys.argv[1:]] or [8, 2]
    targets = (64, 96, 128, 160, 192, 256, 384, 512)
    lines = []
    for batch in batches:
        lines.append(f"batch {batch}: us per (pswin_gemm_tn_ring + sum of its slabs), bf16 slabs; columns = workgroups the splits are chosen for "
                     f"(splits in brackets); * = the current default (256)")
        tot = {t: 0.0 for t in targets}
        for name, M, N, K in shapes(batch):
            if not lib.pswin_gemm_tn_ring_supported(M, N, K):
                lines.append(f"{name:8s} M={M:6d} N={N:4d} K={K:4d}: not on the ring kernel")
                continue
            dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
            x = torch.randn(M, K, device=dev).to(torch.bfloat16)
            out = torch.empty(N * K, dtype=torch.float32, device=dev)
            row = []
            for t in targets:
                sp = int(lib.pswin_gemm_tn_ring_splits(M, N, K, t))

                def run():
                    part = ops.gemm_tn_ring(dy, x, sp, torch.bfloat16 if sp > 1 else torch.float32)
                    if sp > 1:
                        ops.sum_rows(part, sp, N * K, out=out)
                us = graph_time(run)
                tot[t] += us
                row.append(f"{us:6.1f}[{sp:3d}]{'*' if t == 256 else ' '}")
            lines.append(f"{name:8s} M={M:6d} N={N:4d} K={K:4d}: " + " ".join(row))
            print(lines[-1], flush=True)
        lines.append("sum over the shapes (one of each): " + " ".join(f"{t}: {v:.0f}" for t, v in tot.items()))
        print(lines[-1], flush=True)
    with open("gpurun_out/r04_tn_ring_split_sweep.txt", "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
