"""HBM-side fetch traffic of the ring weight-gradient kernel PER PRODUCT: every Linear shape of PanoSwin-T at batch 8 launched alone with
the row splits of the grouped rule (ops.grouped_wgrad_splits), under `rocprofv3 --pmc FETCH_SIZE`; the summary divides the counted bytes by
the product's operand bytes 2 M (N + K).  A one-tile product (stage-0 qkv: 288 x 96 in one <1,8> tile) cannot re-read anything: its ratio
calibrates the counter for this access pattern (rows of 128-byte pieces by LDS-DMA, 16 B per lane).

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/tnp -- python3 tools/pmc_tn_products.py run
  python3 tools/pmc_tn_products.py summarize gpurun_out/tnp/*/*counter_collection.csv gpurun_out/tn_products.json
"""
import csv
import json
import sys

sys.path.insert(0, ".")
REPS = 3


def shapes(batch):
    from panoswintransformerobjectdetection_amd import _lib
    out = []
    for st, C in enumerate((96, 192, 384, 768)):
        H, W = 128 >> st, 256 >> st
        Hp, Wp, nW = _lib.window_grid(_lib.MODE_PANO, H, W)
        Mw, Mt = batch * nW * 49, batch * H * W
        out += [(f"s{st} qkv", Mw, 3 * C, C), (f"s{st} proj", Mw, C, C), (f"s{st} fc1", Mt, 4 * C, C), (f"s{st} fc2", Mt, C, 4 * C)]
        if st < 3:
            out.append((f"m{st + 1} red", Mt // 4, 2 * C, 4 * C))
    return out


def run():
    import torch
    from panoswintransformerobjectdetection_amd import _lib, ops
    lib = _lib.load()
    dev = "cuda:0"
    done = []
    for name, M, N, K in shapes(8):
        if not lib.pswin_gemm_tn_ring_supported(M, N, K):
            continue
        dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        sp = ops.grouped_wgrad_splits(M)
        for _ in range(REPS):
            ops.gemm_tn_ring(dy, x, sp, torch.bfloat16 if sp > 1 else torch.float32)
        torch.cuda.synchronize()
        done.append({"name": name, "M": M, "N": N, "K": K, "splits": sp})
        del dy, x
    json.dump({"reps": REPS, "products": done}, open("gpurun_out/tn_products.json", "w"))


def summarize(csv_path, json_path):
    meta = json.load(open(json_path))
    rows = [r for r in csv.DictReader(open(csv_path)) if r["Counter_Name"] == "FETCH_SIZE" and "gemm_tn_ring_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    reps = meta["reps"]
    assert len(rows) == reps * len(meta["products"]), (len(rows), len(meta["products"]))
    print(f"{'product':8s} {'M':>7s} {'N':>5s} {'K':>5s} splits geometry           grid  | counted MB (raw KiB x 1024) per launch | operands MB | raw ratio | x2 ratio")
    for i, p in enumerate(meta["products"]):
        mine = rows[i * reps:(i + 1) * reps]
        raw = [float(r["Counter_Value"]) * 1024.0 / 1e6 for r in mine]
        algo = 2.0 * p["M"] * (p["N"] + p["K"]) / 1e6
        geom = mine[0]["Kernel_Name"].split("gemm_tn_ring_kernel")[1].split("(")[0]
        grid = int(mine[0]["Grid_Size"]) // int(mine[0]["Workgroup_Size"]) if "Grid_Size" in mine[0] else -1
        last = raw[-1]
        print(f"{p['name']:8s} {p['M']:7d} {p['N']:5d} {p['K']:5d} {p['splits']:6d} {geom:18s} {grid:5d} | " + " ".join(f"{v:8.1f}" for v in raw) +
              f" | {algo:8.1f} | {last / algo:6.3f} | {2 * last / algo:6.3f}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summarize(sys.argv[2], sys.argv[3])
