"""Calibrate FETCH_SIZE for the attention kernels' access pattern (64-byte row segments, rows 576 B apart) against a wide
streaming read: known byte counts, run under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`."""
import torch
dev = "cuda:0"
n = 275576
a = torch.randn(n, 288, device=dev).to(torch.bfloat16)            # 158.7 MB, rows of 576 B
flush = torch.empty(512 * 1024 * 1024 // 4, device=dev)
outs = []
for it in range(3):
    flush.fill_(float(it)); b = a.clone()                          # wide streaming read: 158.7 MB
    flush.fill_(float(it) + .1); c = a[:, 32:64].contiguous()      # one 64-byte segment per row: 17.6 MB useful
    flush.fill_(float(it) + .2); d = a[:, 0:64].contiguous()       # one 128-byte segment per row: 35.3 MB useful
    flush.fill_(float(it) + .3); e = a[:, 0:96].float().sum(1)     # 192 B per row
    outs += [b, c, d, e]
torch.cuda.synchronize()
print("bytes: wide", a.numel() * 2, "seg64", n * 64, "seg128", n * 128, "seg192", n * 192)
