#!/usr/bin/env python3
"""Headline benchmark: panoramas/s of PanoSwin-T (512x1024) backbone training steps on N MI355X GPUs.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward + backward + gradient average (RCCL, N > 1) + AdamW update of the PanoSwin-T backbone on one
synthetic batch of 8 panoramas per GPU (BASELINE.json configs[1]), bf16 compute with fp32 accumulation /
master weights.  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

`roofline` is measured live in this run: HIP-event pairs around every launch of the hand-written kernels (on the stream they are
launched on) in `--kernel-steps` EAGER steps run right after the timed region -- nodes of a replayed hipGraph cannot be bracketed by
host-recorded events; same kernels, shapes and data, a spin kernel in front of each such step keeps the launches back to back (the
line's `kernel_timing` field says so too).  The dominant kernel is computed (largest ms/step), its algorithmic bytes / FLOP are
SURVEY.md section 8(d)'s per-unit figures x the units of the launch (stated next to each call in ops.py; split partial slabs are
reported as `partial_bytes_per_launch`, never as algorithmic bytes).
`cpu_baseline` times the CPU oracle (oracle/panoswin_oracle.py, "port") on this box's host cores, rank 0, N = 1.
"""
import argparse
import json
import os
import sys
import time

# A fresh box has no MIOpen user find-db, so the first bf16 NHWC PatchEmbed convolutions would trigger a ~45 s
# solver search in every process (MIOPEN_FIND_MODE=FAST avoids the search but falls back to naive kernels, 30x
# slower).  The results of that search for the bench shapes are shipped with the package and used when present.
_MIOPEN_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "panoswintransformerobjectdetection_amd", "miopen_db")
if os.path.isdir(_MIOPEN_DB):
    os.environ.setdefault("MIOPEN_USER_DB_PATH", _MIOPEN_DB)
# hipBLASLt solution choice for the ~60 GEMM shapes of the step: results of a PyTorch TunableOp tuning run on MI355X
# are shipped as one table and only READ here (tuning off): +5 % panoramas/s.
_TUNABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "panoswintransformerobjectdetection_amd", "tunableop",
                        "tunableop_results.csv")


def use_shipped_gemm_table(table=_TUNABLE):
    """TunableOp reads <name><device ordinal>.csv: give every ordinal of the node a copy of the ONE shipped table."""
    if not os.path.isfile(table) or "PYTORCH_TUNABLEOP_FILENAME" in os.environ:
        return
    import shutil
    # one deterministic cache directory (not a fresh temporary directory per import): ~/.cache, or the scratch directory of the run
    d = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "pswin_tunableop")
    try:
        os.makedirs(d, exist_ok=True)
    except OSError:
        d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out", "pswin_tunableop")
        os.makedirs(d, exist_ok=True)
    for ordinal in range(8):
        dst = os.path.join(d, f"tunableop_results{ordinal}.csv")
        if not os.path.isfile(dst) or os.path.getmtime(dst) < os.path.getmtime(table) or os.path.getsize(dst) != os.path.getsize(table):
            tmp = f"{dst}.{os.getpid()}.tmp"
            shutil.copyfile(table, tmp)
            os.replace(tmp, dst)                  # atomic: the ranks of one node import this module at the same time
    os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "0")
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(d, "tunableop_results.csv")


use_shipped_gemm_table()

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TCFG = dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, ape=True,
            drop_path_rate=0.2, pano_mode=True)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md; ~6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak (MI355X_MICROARCH.md)
TIMED = ("pswin_win_attn_fused_fwd", "pswin_qkv_attn_fused_fwd", "pswin_attn_fwd", "pswin_attn_bwd", "pswin_attn_bwd_ex", "pswin_window_gather", "pswin_window_scatter_add",
         "pswin_scatter_add_ln_fwd", "pswin_ln_gather_fwd", "pswin_ln_gather_bwd", "pswin_bias_gelu_fwd", "pswin_bias_gelu_bwd",
         "pswin_adamw_flat", "pswin_gemm_skinny", "pswin_gemm_nt", "pswin_gemm_nt_gelu_fwd", "pswin_gemm_nt_gelu_bwd", "pswin_gemm_tn_ring", "pswin_roi_align_fwd", "pswin_roi_align_bwd", "pswin_fc1_gelu_fwd", "pswin_fc1_gelu_bwd", "pswin_mlp0_fwd", "pswin_mlp0_bwd", "pswin_stem_conv2_fwd", "pswin_stem_conv3_fwd",
         "pswin_stem_conv3_bwd_stats", "pswin_stem_conv3_bwd_data", "pswin_stem_conv3_wgrad", "pswin_stem_conv2_wgrad",
         "pswin_stem_conv2_bwd", "lib_gemm_fwd", "lib_gemm_dgrad", "lib_gemm_wgrad")
# hardware MFMA-pipe utilisation of the window-attention kernels: SQ_VALU_MFMA_BUSY_CYCLES of a separate rocprofv3 --pmc pass
# (tools/pmc_fused.py / tools/pmc_attn.py), summarised into this committed file; the bench line quotes it with its source
MFMA_BUSY_FILE = "profiles/r04_pmc_window_attention_mfma.json"


def cpu_baseline(threads):
    """CPU oracle (oracle/panoswin_oracle.py, a port: the reference itself cannot travel) on this box's host cores, fp32,
    as SURVEY.md section 8(d) states: BASELINE configs[0] (forward only, batch 2, no_grad) and the same batch fwd+bwd; 1
    warm-up + 5 runs each, median.  About 25 s of CPU work."""
    import statistics
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import panoswin_oracle as po
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = po.SimplePanoSwinTransformerOracle(**{**TCFG, "drop_path_rate": 0.0})
    m.init_weights(None)
    x = torch.randn(2, 3, 512, 1024)

    def timed(fn, runs=5):
        fn()
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    m.eval()

    def fwd():
        with torch.no_grad():
            m(x)
    t_fwd = timed(fwd)
    m.train()

    def fwd_bwd():
        m.zero_grad(set_to_none=True)
        sum(o.float().mean() for o in m(x)).backward()
    t_fb = timed(fwd_bwd)
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except (OSError, IndexError):
        cpu = "unknown"
    return {"value": round(2 / t_fb, 4), "unit": "panoramas/s", "cores": threads, "kind": "port",
            "sample": "PanoSwin-T fwd+bwd, batch 2 x 3x512x1024 fp32, median of 5 after 1 warm-up, CPU oracle",
            "forward_only": {"value": round(2 / t_fwd, 4), "unit": "panoramas/s",
                             "sample": "BASELINE configs[0]: PanoSwin-T forward, eval mode, batch 2 x 3x512x1024 fp32, median of 5"},
            "cpu_model": cpu}


def _lib_digest():
    """sha1 over the kernel sources: a PMC summary is quoted only for the code it was collected on."""
    import hashlib
    d = os.path.join(ROOT, "panoswintransformerobjectdetection_amd", "csrc")
    h = hashlib.sha1()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".hpp", ".inc")):
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="panoramas per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=32.0)
    ap.add_argument("--eager", action="store_true", help="do not capture the step into a HIP graph")
    ap.add_argument("--no-capture", action="store_true",
                    help="the benched step -- flat gradient buffer, deferred reductions, grouped weight gradients, FlatAdamW -- launched from Python "
                         "instead of replayed from hipGraphs (rocprofv3 --pmc passes: counter collection on a replayed graph does not finish)")
    ap.add_argument("--model", default="T", choices=["T", "S"], help="PanoSwin-T (depths 2-2-6-2, the headline) or -S (2-2-18-2)")
    ap.add_argument("--height", type=int, default=512, help="panorama height; width = 2 * height (headline: 512)")
    ap.add_argument("--graph-heads", type=int, default=1, help="--config maskrcnn: capture the head stand-ins into a hipGraph too (1) or run them eagerly (0)")
    ap.add_argument("--torch-adamw", action="store_true", help="torch.optim.AdamW(fused=True) instead of the one-launch HIP update (A/B)")
    ap.add_argument("--single-group", action="store_true", help="AdamW without the reference's parameter groups (decay on every element: A/B)")
    ap.add_argument("--sustain-steps", type=int, default=200,
                    help="extra steps run and timed AFTER the K timed ones (reported as sustained_ms_per_step: the clock / thermal state "
                         "of a 0.25 s burst is not the state of a training run)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); 'gloo' lets two "
                    "ranks rehearse the N > 1 code path on one GPU together with --device")
    ap.add_argument("--device", type=int, default=None, help="force this HIP device index for every rank (rehearsal only)")
    ap.add_argument("--no-defer", action="store_true",
                    help="issue every parameter-gradient reduction where autograd produces it (A/B of the grouped launch)")
    ap.add_argument("--split-backward", action="store_true",
                    help="use the two-graph step (backward in two pieces, all-reduce overlapped) also on one GPU")
    ap.add_argument("--kernel-steps", type=int, default=3,
                    help="eager steps run after the timed region to time individual kernels with HIP events")
    ap.add_argument("--config", default="backbone", choices=["backbone", "maskrcnn"],
                    help="backbone = BASELINE configs[1] (the headline); maskrcnn = configs[2]: PanoSwin-T + a minimal Mask R-CNN "
                         "head stack (detector.py), synthetic COCO-shaped targets, end-to-end step")
    args = ap.parse_args()
    if args.config == "maskrcnn":
        return main_maskrcnn(args)

    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer, _lib
    from panoswintransformerobjectdetection_amd.dp import GradReducer, init_distributed

    rank, local_rank, world = init_distributed(args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev = torch.device("cuda", local_rank if args.device is None else args.device)
    torch.cuda.set_device(dev)
    # Everything from here on -- parameter allocation, warm-up, capture, replay launches of uncaptured work -- runs on ONE side
    # stream: autograd's AccumulateGrad nodes remember the stream they were created on, and a node created on the default
    # stream makes the capturing backward pass synchronise with it (torch warns; harmless here, but a needless cross-stream edge)
    cap_stream = torch.cuda.Stream()
    torch.cuda.set_stream(cap_stream)
    _lib.load()                                  # fail loudly if the HIP library is missing

    torch.manual_seed(0)
    cd = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cfg = dict(TCFG, depths=[2, 2, 18, 2]) if args.model == "S" else TCFG
    model = SimplePanoSwinTransformer(**cfg, compute_dtype=cd)
    model.init_weights(None)
    model = model.to(dev).train()
    reducer = GradReducer(model, bucket_mb=args.bucket_mb, pack=not args.eager)
    # pack mode reads param.grad only after backward() has returned: the parameter-gradient reductions of the pass can
    # then be issued as one grouped launch at its end (ops.set_deferred_reductions)
    from panoswintransformerobjectdetection_amd import ops as _ops
    _ops.set_deferred_reductions(not args.eager and not args.no_defer)
    reducer.broadcast_parameters(model)
    # The optimizer runs over ONE flat parameter / gradient / state buffer (dp.GradReducer.flatten_parameters): a single fused
    # element-wise launch per step.  The reference's parameter groups (configs/swin/*.py paramwise_cfg: decay_mult = 0 for every
    # parameter whose name contains 'norm') live in a byte map over that buffer (optim.FlatAdamW(paramwise_cfg=...)).
    opt_params = [reducer.flatten_parameters(model, cd if cd != torch.float32 else None)] if not args.eager else list(model.parameters())
    if args.eager or args.torch_adamw:
        opt = torch.optim.AdamW(opt_params, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.05, fused=True, capturable=not args.eager)
    else:
        # the update of the flat buffer and the bf16 copy of the new weights in one HIP launch (the arithmetic of torch.optim.AdamW)
        from panoswintransformerobjectdetection_amd.optim import REFERENCE_PARAMWISE_CFG, FlatAdamW
        opt = FlatAdamW(opt_params[0], lr=1e-4, betas=(0.9, 0.999), weight_decay=0.05, model=model,
                        paramwise_cfg=None if args.single_group else REFERENCE_PARAMWISE_CFG)

    torch.manual_seed(1234 + rank)               # every rank its own shard of synthetic panoramas
    x = torch.randn(args.batch, 3, args.height, 2 * args.height, device=dev)

    # Synthetic objective: a fixed random linear functional of the four feature maps (what a detection head's
    # gradient looks like to the backbone: dense, zero-mean, O(1/numel)).  The mean of the LayerNorm-ed outputs used in
    # earlier rounds is ~0 with a degenerate gradient, and torch's two-pass global reductions of graph-pool tensors
    # return stale values from the second hipGraph replay on (ROCm 7.0 / torch 2.10: tools/repro_graph_stale_reduction.py); a dot
    # product is a single-pass GEMV and replays bit-exactly (tests/test_backbone_gpu.py::test_hipgraph_replay_matches_eager checks every gradient against eager).
    with torch.no_grad():
        loss_w = [torch.randn_like(o).flatten() / o.numel() for o in model(x)]
    for p in model.parameters():
        p.grad = None

    def fwd_bwd():
        reducer.zero_grad()
        outs = model(x)
        loss = sum(o.float().flatten() @ w for o, w in zip(outs, loss_w))
        loss.backward()
        if reducer.pack:
            reducer.pack_grads()
        return loss

    def eager_step():
        loss = fwd_bwd()
        reducer.finish()
        opt.step()
        return loss

    # Split backward (N > 1): stages 2-3 hold ~90 % of the parameters and their gradients are finished first.  The step
    # is captured as TWO graphs sharing one memory pool -- (forward + backward of stages 2-3 + pack) and (backward of
    # stages 0-1 and the stem + pack) -- and the RCCL all-reduce of the late group's buckets (the reducer cuts its buckets
    # at the group boundary: tests/test_dp_gloo.py asserts >= 90 % of the gradient bytes) is launched between the two
    # replays, so it runs on xGMI under the second graph's kernels; only the early group's bucket is exposed.  One xGMI
    # link per GPU pair: at N = 2 the 112 MB all-reduce would otherwise add ~1.7 ms to a 14 ms step.
    from panoswintransformerobjectdetection_amd.dp import BoundaryTap, backward_early, backward_late, split_parameters
    late_params, early_params = split_parameters(model)          # = model.grad_groups(): the reducer's buckets follow it
    tap = BoundaryTap(model.layers[2])
    carry = {}

    def phase1():
        reducer.zero_grad()
        outs = model(x)
        l_early = sum(o.float().flatten() @ w for o, w in zip(outs[:2], loss_w[:2]))
        l_late = sum(o.float().flatten() @ w for o, w in zip(outs[2:], loss_w[2:]))
        xb = tap.x
        g_xb = backward_late(l_late, xb, late_params)
        reducer.pack_grads(late_params)
        carry.update(l_early=l_early, xb=xb, g_xb=g_xb)
        return (l_early + l_late).detach()

    def phase2():
        backward_early(carry["l_early"], carry["xb"], carry["g_xb"], early_params)
        reducer.pack_grads(early_params)
        return carry["g_xb"]

    split = (world > 1 or args.split_backward) and not args.eager and not args.no_capture
    if args.eager:
        step = eager_step
    elif args.no_capture:
        def step():
            loss = fwd_bwd()
            reducer.finish()
            opt.step()
            return loss
    elif split:
        from panoswintransformerobjectdetection_amd.graph import GraphedCallable, GraphedSequence
        seq = GraphedSequence([phase1, phase2], warmup=2, stream=cap_stream)
        g_opt = GraphedCallable(opt.step, warmup=1, stream=seq.stream)
        late_buckets = reducer.buckets_within(late_params)

        def step():
            loss = seq.calls[0]()
            reducer.launch(late_buckets)          # RCCL runs under the second graph
            seq.calls[1]()
            reducer.finish()
            g_opt()
            return loss
    elif world == 1:
        # one process, nothing to exchange between the backward pass and the update: the whole step is ONE hipGraph
        from panoswintransformerobjectdetection_amd.graph import GraphedCallable

        def fwd_bwd_update():
            loss = fwd_bwd()
            opt.step()
            return loss
        step = GraphedCallable(fwd_bwd_update, warmup=2, stream=cap_stream)
    else:
        # One hipGraph for forward+backward, one for the optimizer; the RCCL all-reduce of the flat gradient buffer
        # runs between the two replays (N > 1), so collectives are never part of a captured graph.
        from panoswintransformerobjectdetection_amd.graph import GraphedCallable
        g_fb = GraphedCallable(fwd_bwd, warmup=2, stream=cap_stream)
        g_opt = GraphedCallable(opt.step, warmup=1, stream=g_fb.stream)

        def step():
            loss = g_fb()
            reducer.finish()
            g_opt()
            return loss
    if not split:
        tap.remove()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if args.eager:
        _lib.enable_timing(TIMED)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    sustained = None
    if args.sustain_steps > 0:
        # the driver's K steps last a quarter of a second; the same step for another --sustain-steps, timed the same way
        t1 = time.perf_counter()
        for _ in range(args.sustain_steps):
            loss = step()
        barrier()
        sustained = (time.perf_counter() - t1) / args.sustain_steps * 1e3
    if not args.eager:
        # Nodes of a replayed graph cannot be bracketed by host-recorded events, so the per-kernel HIP-event timing
        # runs on eager steps of the same model / batch right after the timed region (same kernels, shapes, data).
        # An eager step is host-bound (the GPU drains its queue and idles between launches), which would put idle time
        # inside the event pairs.  A spin kernel in front of every step lets the host enqueue the whole step first,
        # so the kernels then run back to back exactly as they do inside the graph.
        spin = 200_000_000 if hasattr(torch.cuda, "_sleep") else 0        # ~80-100 ms of spinning at 2-2.4 GHz
        _lib.enable_timing(TIMED)
        for _ in range(args.kernel_steps):
            if spin:
                torch.cuda._sleep(spin)
            eager_step()
    kern = _lib.disable_timing()
    ksteps = args.steps if args.eager else args.kernel_steps
    if world > 1:
        t = torch.tensor([elapsed, sustained or 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        sustained = float(t[1].item()) if sustained is not None else None
    assert torch.isfinite(loss).item(), "non-finite loss"
    # (outside the timed region) every parameter and every gradient of the last step is finite: a finite loss alone once let
    # non-finite stem weight gradients through
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters()), "non-finite parameter after the timed steps"
    assert bool(torch.isfinite(reducer.flat).all()), "non-finite gradient in the last step"

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * world * args.steps / elapsed
        stats = {}
        for name, recs in kern.items():
            if recs:
                tot_ms = sum(r[0] for r in recs)
                st = {"launches": len(recs), "avg_us": round(tot_ms / len(recs) * 1e3, 2), "max_us": round(max(r[0] for r in recs) * 1e3, 2),
                      "ms_per_step": round(tot_ms / ksteps, 3),
                      "GBps": round(sum(r[1] for r in recs) / (tot_ms * 1e-3) / 1e9, 1)}
                fl = sum(r[2] for r in recs)
                if fl:
                    st["TFLOPs"] = round(fl / (tot_ms * 1e-3) / 1e12, 1)
                stats[name] = st
        digest = _lib_digest()

        def roof(name):
            """roofline entry of one timed kernel.  Per launch the floor is max(algorithmic bytes / 8 TB/s, algorithmic FLOP /
            2.5 PFLOP/s); `bound` says which of the two dominates over the kernel's launches, `achieved` / `peak` are quoted
            in that unit and frac = achieved / peak (for a mix of shapes: sum of the floors / sum of the times)."""
            recs = kern[name]
            t = sum(r[0] for r in recs) * 1e-3
            fl_h = sum(r[1] for r in recs) / (HBM_PEAK_GBS * 1e9)
            fl_m = sum(r[2] for r in recs) / (MFMA_PEAK_TFLOPS * 1e12)
            out = {"kernel": name, "avg_launch_us": stats[name]["avg_us"], "ms_per_step": stats[name]["ms_per_step"],
                   "launches_per_step": round(len(recs) / ksteps, 1)}
            if fl_m > fl_h:
                ach = sum(r[2] for r in recs) / t / 1e12
                out.update(bound="mfma", achieved=round(ach, 1), peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ach / MFMA_PEAK_TFLOPS, 4),
                           algorithmic_flop_per_launch=round(sum(r[2] for r in recs) / len(recs)))
            else:
                ach = sum(r[1] for r in recs) / t / 1e9
                out.update(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                           algorithmic_bytes_per_launch=round(sum(r[1] for r in recs) / len(recs)))
            out["frac_of_mixed_floor"] = round(sum(max(r[1] / (HBM_PEAK_GBS * 1e9), r[2] / (MFMA_PEAK_TFLOPS * 1e12)) for r in recs) / t, 4)
            pb = sum(r[3] for r in recs)
            if pb:               # split partial slabs etc.: an implementation artefact, NOT part of the algorithmic bytes `frac` is computed from
                out["partial_bytes_per_launch"] = round(pb / len(recs))
            return out

        # dominant hand-written kernel = the largest ms/step among the timed C-ABI entry points (computed, not assumed)
        own = [n for n in stats if n.startswith("pswin_")]
        dom = max(own, key=lambda n: stats[n]["ms_per_step"]) if own else None
        roofline = roof(dom) if dom else {"kernel": None, "bound": None, "achieved": None, "peak": None, "unit": None, "frac": None,
                                          "note": "--kernel-steps 0: no per-kernel timing in this run"}
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x correction + WRITE_SIZE, separate rocprofv3 --pmc passes,
        # tools/pmc_summary.py -> profiles/pmc_traffic.json); quoted only when collected on THIS build of the kernels
        traffic, tnote = None, "no PMC summary for this kernel"
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pt = json.load(f)
            if pt.get("lib_digest") != digest:
                tnote = f"profiles/pmc_traffic.json was collected on kernel sources {pt.get('lib_digest')}, this run is {digest}: not quoted"
            elif dom and dom in pt["kernels"]:
                traffic, tnote = round(pt["kernels"][dom]["hbm_bytes_per_launch"]), "profiles/pmc_traffic.json"
        except (OSError, KeyError, ValueError):
            pass
        roofline["traffic"], roofline["traffic_source"] = traffic, tnote
        # the window-attention kernels (what BASELINE's metric string and the north star grade), whichever is dominant
        roofline["window_attention"] = {n: roof(n) for n in ("pswin_win_attn_fused_fwd", "pswin_qkv_attn_fused_fwd", "pswin_attn_fwd", "pswin_attn_bwd_ex") if n in stats}
        try:
            with open(os.path.join(ROOT, MFMA_BUSY_FILE)) as f:
                mb = json.load(f)
            roofline["window_attn_mfma_busy"] = {"source": MFMA_BUSY_FILE, "same_kernel_sources": mb.get("lib_digest") == digest,
                                                 **{k: v for k, v in mb.items() if k not in ("lib_digest",)}}
        except (OSError, ValueError):
            roofline["window_attn_mfma_busy"] = None
        # the library (hipBLASLt through PyTorch) GEMM pool: time per step against its own floor max(bytes / 6.3 TB/s,
        # FLOP / 2.5 PFLOP/s) summed over the launches
        pool = {}
        for n in ("lib_gemm_fwd", "lib_gemm_dgrad", "lib_gemm_wgrad"):
            recs = kern.get(n) or []
            if recs:
                floor_ms = sum(max(r[1] / 6.3e12, r[2] / 2.5e15) for r in recs) * 1e3
                t_ms = sum(r[0] for r in recs)
                pool[n] = {"launches_per_step": len(recs) // ksteps, "ms_per_step": round(t_ms / ksteps, 3),
                           "floor_ms_per_step": round(floor_ms / ksteps, 3), "frac_of_floor": round(floor_ms / t_ms, 3),
                           "TFLOPs": round(sum(r[2] for r in recs) / (t_ms * 1e-3) / 1e12, 1)}
        roofline["library_gemm_pool"] = pool
        roofline["kernels"] = stats
        roofline["kernel_timing"] = ("HIP-event pairs around each launch in eager steps after the timed region (nodes of a replayed graph cannot be "
                                     "bracketed); the same kernels inside graph replays measure 5-12 % shorter under rocprofv3 (profiles/*_kernel_stats*), "
                                     "so every frac here is a lower bound")
        roofline["kernel_sources_digest"] = digest
        line = {
            "metric": f"panoramas/sec PanoSwin-{args.model} {args.height}x{2 * args.height} fwd+bwd", "value": round(value, 2), "unit": "panoramas/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "sustained_ms_per_step": None if sustained is None else round(sustained, 3), "sustain_steps": args.sustain_steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"PanoSwin-{args.model} backbone (embed 96, depths {'-'.join(map(str, cfg['depths']))}, "
                                    f"heads 3-6-12-24, ape, pano mode) fwd+bwd+AdamW on 3x{args.height}x{2 * args.height} "
                                    "panoramas" + (", BASELINE.json configs[1]" if (args.model, args.height) == ("T", 512) else "")),
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "grad_bucket_mb": args.bucket_mb, "hip_graph": not args.eager and not args.no_capture, "overlap_allreduce": bool(split),
                       "optimizer": ("AdamW lr 1e-4 wd 0.05, one launch over the flat parameter buffer"
                                     + (", single group (A/B)" if args.single_group or args.eager or args.torch_adamw else
                                        ", parameter groups of the reference's paramwise_cfg (decay_mult 0 for 'norm' parameters)")),
                       "fused_window_attention": bool(_ops.FUSED_WINDOW_ATTENTION),
                       "device": torch.cuda.get_device_name(dev)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            line["cpu_baseline"] = cpu_baseline(min(cores, 16))      # the box's CPU share for one GPU is 16
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_maskrcnn(args):
    """BASELINE configs[2] / [3]: one end-to-end Mask R-CNN training step (backbone + FPN + RPN + RoI heads + AdamW), batch 2
    per GPU unless --batch says otherwise.  The backbone's forward and backward are two hipGraphs sharing a memory pool; the
    heads (dynamic-shaped PyTorch operators: parity unpinned, see detector.py) run eagerly between the two replays on
    detached feature maps and hand their feature-map gradients to the second graph."""
    from panoswintransformerobjectdetection_amd import _lib, ops as _ops
    from panoswintransformerobjectdetection_amd.detector import MiniMaskRCNN, synthetic_targets
    from panoswintransformerobjectdetection_amd.dp import GradReducer, init_distributed, reduce_loss_scalars
    from panoswintransformerobjectdetection_amd.graph import GraphedCallable, GraphedSequence

    rank, local_rank, world = init_distributed(args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local_rank if args.device is None else args.device)
    torch.cuda.set_device(dev)
    cap_stream = torch.cuda.Stream()             # one stream for allocation, warm-up, capture and replay (see main())
    torch.cuda.set_stream(cap_stream)
    _lib.load()
    batch = 2 if args.batch == 8 else args.batch          # configs[3]: batch = 2 per GPU
    if os.environ.get("PSWIN_MIOPEN_FIND") == "1":
        # one-off: let MIOpen search its solvers for the head stand-ins' convolutions and record the results in
        # MIOPEN_USER_DB_PATH (tools/miopen_find_heads.sh copies them into the shipped find-db); without a record MIOpen's
        # immediate mode runs these shapes on its naive reference kernels
        torch.backends.cudnn.benchmark = True
    H = args.height
    torch.manual_seed(0)
    model = MiniMaskRCNN(dict(TCFG, compute_dtype=torch.bfloat16)).to(dev).train()
    model.backbone.init_weights(None)
    bb = model.backbone
    red = GradReducer(bb, bucket_mb=args.bucket_mb, pack=True)
    _ops.set_deferred_reductions(True)
    red.broadcast_parameters(bb)
    head_params = model.head_parameters()
    if world > 1:
        for p in head_params:
            dist.broadcast(p.data, src=0)
    from panoswintransformerobjectdetection_amd.optim import REFERENCE_PARAMWISE_CFG, FlatAdamW
    opt_bb = FlatAdamW(red.flatten_parameters(bb, torch.bfloat16), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.05, model=bb,
                       paramwise_cfg=REFERENCE_PARAMWISE_CFG)
    opt_hd = torch.optim.AdamW(head_params, lr=1e-4, weight_decay=0.05, fused=True)
    torch.manual_seed(1234 + rank)
    x = torch.randn(batch, 3, H, 2 * H, device=dev)
    targets = synthetic_targets(batch, H, 2 * H, dev, seed=rank)
    with torch.no_grad():
        gbuf = [torch.zeros_like(o) for o in bb(x)]
    for p in bb.parameters():
        p.grad = None
    state = {}

    def phase_fwd():
        red.zero_grad()
        state["outs"] = bb(x)
        return state["outs"]

    def phase_bwd():
        torch.autograd.backward(state["outs"], gbuf)
        red.pack_grads()
        return gbuf[0]

    def phase_heads():
        feats = [o.detach().requires_grad_(True) for o in state["outs"]]
        for p in head_params:
            p.grad = None
        losses = model.heads_loss(feats, targets, (H, 2 * H))
        total = sum(losses.values())
        total.backward()
        for g, f in zip(gbuf, feats):
            g.copy_(f.grad)
        state["losses"] = losses
        return total

    # --graph-heads: the head stand-ins (static shapes, no host synchronisation) captured as a third hipGraph between the two
    # backbone graphs instead of ~2,000 eager launches
    graph_heads = bool(args.graph_heads)
    seq = GraphedSequence([phase_fwd, phase_heads, phase_bwd] if graph_heads else [phase_fwd, phase_bwd], warmup=2, stream=cap_stream)
    g_opt = GraphedCallable(opt_bb.step, warmup=1, stream=seq.stream)
    if graph_heads:
        opt_hd = torch.optim.AdamW(head_params, lr=1e-4, weight_decay=0.05, fused=True, capturable=True)
        g_opt_hd = GraphedCallable(opt_hd.step, warmup=1, stream=seq.stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    parts = []

    def step(record=False):
        if record:
            ev[0].record()
        outs = seq.calls[0]()
        if record:
            ev[1].record()
        if graph_heads:
            total = seq.calls[1]()
            losses = state["losses"]
        else:
            feats = [o.detach().requires_grad_(True) for o in outs]
            opt_hd.zero_grad(set_to_none=True)
            losses = model.heads_loss(feats, targets, (H, 2 * H))
            total = sum(losses.values())
            total.backward()
            for g, f in zip(gbuf, feats):
                g.copy_(f.grad)
        if record:
            ev[2].record()
        seq.calls[-1]()
        if record:
            ev[3].record()
        if world > 1:
            flat = torch.cat([p.grad.flatten() for p in head_params])
            dist.all_reduce(flat, op=dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM)
            if dist.get_backend() != "nccl":
                flat /= world
            at = 0
            for p in head_params:
                p.grad.copy_(flat[at:at + p.numel()].view_as(p))
                at += p.numel()
        red.finish()
        g_opt()
        if graph_heads:
            g_opt_hd()
        else:
            opt_hd.step()
        if world > 1:
            # the logged losses: BaseDetector._parse_losses all-reduces every scalar on its own (mmdet/models/detectors/base.py:213-218);
            # here the five of them travel as ONE small all-reduce per iteration
            losses = reduce_loss_scalars(losses)
        if record:
            ev[4].record()
            torch.cuda.synchronize()
            parts.append([ev[i].elapsed_time(ev[i + 1]) for i in range(4)])
        return total, losses

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total, losses = step()
    barrier()
    elapsed = time.perf_counter() - t0
    for _ in range(3):
        step(record=True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(total).item(), "non-finite loss"
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        br = [sum(p[i] for p in parts) / len(parts) for i in range(4)]
        line = {
            "metric": f"panoramas/sec PanoSwin-T + Mask R-CNN {H}x{2 * H} end-to-end training step", "value": round(batch * world * args.steps / elapsed, 2),
            "unit": "panoramas/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{2 if world == 1 else 3}]: PanoSwin-T backbone (HIP) + FPN + RPN + Mask R-CNN RoI heads (PyTorch "
                                   f"stand-ins + HIP RoIAlign with the config's sampling_ratio = 0; head parity unpinned), synthetic COCO-shaped targets, 3x{H}x{2 * H}, fwd + bwd + AdamW",
                       "batch_per_gpu": batch, "global_batch": batch * world, "parallelism": f"dp{world}",
                       "device": torch.cuda.get_device_name(dev)},
            "step_breakdown_ms": {"backbone_forward_graph": round(br[0], 3), "heads_forward_backward": round(br[1], 3), "heads_in_hipgraph": graph_heads,
                                  "backbone_backward_graph": round(br[2], 3), "allreduce_and_optimizers": round(br[3], 3)},
            "losses": {k: round(float(v), 4) for k, v in losses.items()},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
