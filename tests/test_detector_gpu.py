"""BASELINE.json configs[2] on the MI355X box: PanoSwin backbone (HIP) + the minimal Mask R-CNN head stack, one training
step.  Head parity is unpinned (detector.py); what is asserted: finite losses, gradients for every parameter, and that the
backbone's parameter gradients through the detector equal the backbone-only backward pass fed with the same feature-map
gradients (the two-graph arrangement bench.py --config maskrcnn replays)."""
import pytest
import torch

from _util import TINY

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16])
def test_mask_rcnn_step_and_backbone_gradient_hand_off(cd):
    from panoswintransformerobjectdetection_amd.detector import MiniMaskRCNN, synthetic_targets
    torch.manual_seed(0)
    m = MiniMaskRCNN(dict(TINY, compute_dtype=cd), num_classes=80).to(DEV).train()
    m.backbone.init_weights(None)
    B, H, W = 2, 128, 256
    x = torch.randn(B, 3, H, W, device=DEV)
    tg = synthetic_targets(B, H, W, DEV)

    outs = m.backbone(x)
    for o in outs:
        o.retain_grad()                                        # the feature-map gradients the heads send back
    losses = m.heads_loss(outs, tg, (H, W))
    total = sum(losses.values())
    assert torch.isfinite(total) and all(torch.isfinite(v) and v >= 0 for v in losses.values())
    total.backward()
    G = [o.grad.detach().clone() for o in outs]
    ref = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    assert all(torch.isfinite(g).all() for g in ref.values()) and all(torch.isfinite(g).all() and g.abs().sum() > 0 for g in G)
    assert sum(float(g.abs().sum()) > 0 for k, g in ref.items() if k.startswith("backbone.")) > 100
    assert all(ref[k].abs().sum() > 0 for k in ref if not k.startswith("backbone."))

    # the arrangement bench.py --config maskrcnn replays: backbone forward | heads on detached maps | backbone backward fed
    # with the heads' feature-map gradients.  Same gradients G -> same backbone parameter gradients as the end-to-end pass.
    for p in m.parameters():
        p.grad = None
    outs2 = m.backbone(x)
    torch.autograd.backward(outs2, G)
    from _util import ZERO_GRAD_KEYS, record
    worst = {"stem": (0.0, ""), "rest": (0.0, "")}
    for k, p in m.backbone.named_parameters():
        if any(z in k for z in ZERO_GRAD_KEYS):
            continue
        r = ref["backbone." + k]
        err = float((p.grad - r).norm() / r.norm().clamp_min(1e-30))
        grp = "stem" if k.startswith("patch_embed.proj") else "rest"
        if err > worst[grp][0]:
            worst[grp] = (err, k)
    record(f"detector_hand_off_{'fp32' if cd == torch.float32 else 'bf16'}", stem_rel=worst["stem"][0], stem_key=worst["stem"][1],
           rest_rel=worst["rest"][0], rest_key=worst["rest"][1])
    # behind the stem: this repository's kernels (fixed summation order) + library GEMMs; the small stem of this tiny model
    # runs on MIOpen convolutions, whose backward-weights kernels are not bitwise reproducible from call to call
    assert worst["rest"][0] < 1e-5, worst
    assert worst["stem"][0] < 1e-3, worst
    assert all(p.grad is None for k, p in m.named_parameters() if not k.startswith("backbone."))


def _fixed_keys():
    """A deterministic stand-in for the samplers' torch.rand_like: the same keys in every call of a given size, so that an eager
    step and a replayed (captured) step sample the same anchors / RoIs."""
    cache = {}

    def rand_like(t):
        n = t.numel()
        if n not in cache:
            g = torch.Generator("cpu").manual_seed(1000 + n)
            cache[n] = torch.rand(n, generator=g).to(t.device)
        return cache[n].view_as(t).to(t.dtype)
    return rand_like


def test_configs2_workload_panoswin_t_512x1024_step_and_its_hipgraph_arrangement():
    """BASELINE.json configs[2] AT ITS OWN SIZE: PanoSwin-T backbone (bf16) + the Mask R-CNN head stack on 2 x 3 x 512 x 1024
    synthetic COCO-shaped panoramas, one training step.  (1) property checks of the end-to-end step: finite non-negative losses,
    every parameter of the backbone and of every head reached by a finite gradient; (2) the hand-off bench.py --config maskrcnn
    uses -- backbone forward | heads on detached maps | backbone backward fed with the heads' feature-map gradients -- reproduces
    the end-to-end backbone gradients; (3) that arrangement captured as THREE hipGraphs sharing a memory pool (what the bench
    replays) gives the same losses as the eager step and backbone gradients equal to an eager backward pass fed with the replay's own
    feature-map gradients, first and later replays."""
    # Everything -- parameter allocation, the eager passes, warm-up, capture, replay -- runs on ONE side stream, as in bench.py: autograd's
    # AccumulateGrad nodes remember the stream of the first backward pass, and nodes created on the default stream make a later capture
    # on another stream synchronise with it (torch warns "may break CUDA graph capture"; here hipStreamEndCapture then crashes).
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _configs2_body(side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()


def _configs2_body(side):
    from _util import TCFG, ZERO_GRAD_KEYS, record
    from panoswintransformerobjectdetection_amd import ops as _ops
    from panoswintransformerobjectdetection_amd.detector import MiniMaskRCNN, synthetic_targets
    from panoswintransformerobjectdetection_amd.dp import GradReducer
    from panoswintransformerobjectdetection_amd.graph import GraphedSequence
    torch.manual_seed(0)
    cfg = dict(TCFG, drop_path_rate=0.0, compute_dtype=torch.bfloat16)
    m = MiniMaskRCNN(cfg, num_classes=80).to(DEV).train()
    m.backbone.init_weights(None)
    m.rand_like = _fixed_keys()
    B, H, W = 2, 512, 1024
    x = torch.randn(B, 3, H, W, device=DEV)
    tg = synthetic_targets(B, H, W, DEV)
    bb = m.backbone
    heads = m.head_parameters()

    # (1) end to end
    outs = bb(x)
    assert [tuple(o.shape) for o in outs] == [(B, 96, 128, 256), (B, 192, 64, 128), (B, 384, 32, 64), (B, 768, 16, 32)]
    for o in outs:
        o.retain_grad()
    losses = m.heads_loss(outs, tg, (H, W))
    assert set(losses) == {"loss_rpn_cls", "loss_rpn_bbox", "loss_cls", "loss_bbox", "loss_mask"}
    total = sum(losses.values())
    assert torch.isfinite(total) and all(torch.isfinite(v) and v >= 0 for v in losses.values())
    total.backward()
    G = [o.grad.detach().clone() for o in outs]
    ref = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    assert all(torch.isfinite(g).all() for g in ref.values()) and all(torch.isfinite(g).all() and g.abs().sum() > 0 for g in G)
    assert all(float(g.abs().sum()) > 0 for k, g in ref.items() if not any(z in k for z in ZERO_GRAD_KEYS)), \
        [k for k, g in ref.items() if float(g.abs().sum()) == 0]
    eager_losses = {k: float(v) for k, v in losses.items()}

    # (2) the hand-off, eagerly
    def rel(a, b):
        return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))

    for p in m.parameters():
        p.grad = None
    torch.autograd.backward(bb(x), G)
    worst = max((rel(p.grad, ref["backbone." + k]), k) for k, p in bb.named_parameters() if not any(z in k for z in ZERO_GRAD_KEYS))
    record("configs2_hand_off_bf16", worst_rel=worst[0], key=worst[1])
    assert worst[0] < 1e-5, worst                      # the fused bf16 stem and everything behind it run in a fixed summation order
    assert all(p.grad is None for p in heads)

    # (3) bench.py --config maskrcnn's arrangement: three graphs, one pool
    red = GradReducer(bb, pack=True)
    prev = _ops.set_deferred_reductions(True)
    try:
        gbuf = [torch.zeros_like(g) for g in G]
        state = {}

        def phase_fwd():
            red.zero_grad()
            state["outs"] = bb(x)
            return state["outs"]

        def phase_heads():
            feats = [o.detach().requires_grad_(True) for o in state["outs"]]
            for p in heads:
                p.grad = None
            ls = m.heads_loss(feats, tg, (H, W))
            t = sum(ls.values())
            t.backward()
            for g, f in zip(gbuf, feats):
                g.copy_(f.grad)
            state["losses"] = torch.stack([ls[k] for k in sorted(ls)])
            return t

        def phase_bwd():
            torch.autograd.backward(state["outs"], gbuf)
            red.pack_grads()
            return gbuf[0]

        seq = GraphedSequence([phase_fwd, phase_heads, phase_bwd], warmup=2, stream=side)
        for replay in range(3):
            seq.calls[0]()
            seq.calls[1]()
            seq.calls[2]()
            torch.cuda.synchronize()
            got = dict(zip(sorted(eager_losses), state["losses"].tolist()))
            for k, v in eager_losses.items():
                # same samples (fixed keys); RoIAlign's backward and torch's index / scatter backward passes add with atomics, and the
                # heads' weights are untouched between replays: losses agree to rounding
                assert abs(got[k] - v) <= 2e-3 * max(abs(v), 1e-3), (replay, k, got[k], v)
            # The head stand-ins are NOT run-to-run deterministic on identical feature maps (tools/debug_heads_determinism.py: MIOpen's
            # bf16 convolutions differ in the last bit between calls, and with freshly initialised RPN / box heads thousands of scores
            # are within that bit, so a few of the 2 x 512 sampled RoIs change from call to call: the classification loss moves in its
            # sixth digit, its feature-map gradient by ~12 % in norm).  So the heads' gradients of a replay are compared with the eager
            # step's only loosely; what the ARRANGEMENT has to guarantee is checked exactly: the backbone's captured backward pass, fed
            # with whatever feature-map gradients this replay's heads produced, equals an eager backward pass fed with the same ones.
            head_rel = [rel(g, want) for g, want in zip(gbuf, G)]
            assert all(bool(torch.isfinite(g).all()) for g in gbuf) and max(head_rel) < 0.5, (replay, head_rel)
            got_grads = {k: p.grad.detach().clone() for k, p in bb.named_parameters()}
            fed = [g.clone() for g in gbuf]
            red.zero_grad()
            torch.autograd.backward(bb(x), fed)
            red.pack_grads()
            werr = max((rel(got_grads[k], p.grad), k) for k, p in bb.named_parameters() if not any(z in k for z in ZERO_GRAD_KEYS))
            record(f"configs2_graph_replay{replay}", worst_rel=werr[0], key=werr[1], heads_vs_eager_rel=max(head_rel))
            assert werr[0] < 1e-5, (replay, werr)
            assert all(bool(torch.isfinite(p.grad).all()) for p in heads)
    finally:
        _ops.set_deferred_reductions(prev)


def test_conv_bias_gradients_of_every_width_stay_fresh_across_graph_replays():
    """the heads' convolution bias gradients all go through the C ABI's fixed-order column sums -- also the RPN's 3- and
    12-channel convolutions, whose rows are zero-padded to 8 columns -- because the framework's two-pass reductions return stale results
    from the second replay of a captured hipGraph.  The captured step is replayed four times on different data; every bias gradient of
    every replay must equal the sum of THAT replay's own output gradient."""
    from panoswintransformerobjectdetection_amd.detector import conv_bias
    from panoswintransformerobjectdetection_amd.graph import GraphedCallable
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.manual_seed(0)
        convs = [torch.nn.Conv2d(16, c, k, padding=k // 2).to(DEV) for c, k in ((3, 1), (12, 1), (256, 3), (80, 1))]
        deconv = torch.nn.ConvTranspose2d(16, 20, 2, stride=2).to(DEV)
        x = torch.randn(2, 16, 40, 72, device=DEV)
        wts = [torch.zeros(2, c.out_channels, 40, 72, device=DEV) for c in convs] + [torch.zeros(2, 20, 80, 144, device=DEV)]
        mods = convs + [deconv]

        def step():
            for m in mods:
                m.bias.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                total = sum((conv_bias(m, x).float() * w).sum() for m, w in zip(mods, wts))
            total.backward()
            return total.detach()

        for w in wts:
            w.normal_()
        g = GraphedCallable(step, warmup=2, stream=side, parameters=mods)
        for replay in range(4):
            for w in wts:
                w.normal_()                                         # new output gradients for this replay
            g()
            side.synchronize()
            for m, w in zip(mods, wts):
                want = w.to(torch.bfloat16).double().sum((0, 2, 3))    # dy of the bf16 convolution output is w rounded to bf16
                got = m.bias.grad.double()
                scale = w.abs().sum((0, 2, 3)).double()
                assert torch.all((got - want).abs() <= 1e-3 * scale + 1e-6), (replay, m.out_channels, (got - want).abs().max())
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
