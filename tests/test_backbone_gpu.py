"""The MI355X backbone against the golden fixtures captured from the live reference.  Needs an MI355X."""
import pytest
import torch

from _util import (SCFG, TCFG, TINY, TINY_PITCH, build_filled, compare_to_golden, golden, model_inputs, record,
                   rel_norm_errors, run_and_collect, sub)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(cfg, pano, tag, train=True, **kw):
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
    return build_filled(lambda **c: SimplePanoSwinTransformer(**c, **kw), cfg, pano, tag, train=train).to(DEV)


@pytest.mark.parametrize("fixture,cfg,pano,shape,tag", [
    ("tiny_pano", TINY, True, (2, 3, 64, 128), "tiny"),
    ("tiny_planar", TINY, False, (2, 3, 64, 128), "tiny"),
    ("tiny_planar_odd", TINY, False, (2, 3, 60, 100), "tiny"),
    ("tiny_pano_oddw", TINY, True, (1, 3, 100, 196), "tiny"),
    ("tiny_pitch_pano", TINY_PITCH, True, (2, 3, 64, 128), "tinyp"),
    ("tiny_pitch_planar", TINY_PITCH, False, (2, 3, 60, 100), "tinyp"),
])
def test_tiny_models_fp32(fixture, cfg, pano, shape, tag):
    """fp32 tolerance of the north star: outputs rtol 1e-4 / atol 5e-4 on the LayerNorm-ed maps; gradients 1e-3 of
    their scale (GPU GEMM / conv summation order differs from the CPU reference)."""
    m = _model(cfg, pano, tag)
    res = run_and_collect(m, shape, tag, device=DEV)
    compare_to_golden(res, golden(fixture), rtol=1e-4, atol=5e-4, grad_rtol=2e-3, grad_atol_frac=1e-3)


@pytest.mark.parametrize("fixture,pano,shape", [("tiny_pano_eval", True, (2, 3, 64, 128)),
                                                ("tiny_planar_eval", False, (2, 3, 60, 100))])
def test_tiny_models_eval_mode_fp32(fixture, pano, shape):
    """eval mode (BatchNorm running statistics, DropPath off; HOT:981-983): outputs, dx and every parameter gradient."""
    m = _model(TINY, pano, "tiny", train=False)
    assert not m.training
    res = run_and_collect(m, shape, "tiny", device=DEV)
    compare_to_golden(res, golden(fixture), rtol=1e-4, atol=5e-4, grad_rtol=2e-3, grad_atol_frac=1e-3, report=fixture)


def test_T_512x1024_fp32():
    """fp32 path at full PanoSwin-T size against the live reference's captured outputs / gradients.  SURVEY 8c states
    atol 5e-4 on the LayerNorm-ed outputs; the measured worst error is recorded in gpurun_out/parity_report.json."""
    m = _model(TCFG, True, "T")
    res = run_and_collect(m, (2, 3, 512, 1024), "T", device=DEV, subsample_out=4096)
    # measured (gpurun_out/parity_report.json): outputs max |err| 5.5e-6; parameter gradients 0.26 of (2e-3, 1e-3); the input
    # gradient (through the two train-mode BatchNorms of the MIOpen fp32 stem) 3e-4 in relative 2-norm with isolated
    # near-cancellation elements at 2.9x that pair, hence its own (1e-2, 5e-3)
    compare_to_golden(res, golden("T_512x1024_pano"), rtol=1e-4, atol=5e-4, grad_rtol=2e-3, grad_atol_frac=1e-3,
                      dx_tol=(1e-2, 5e-3), report="T_512x1024_fp32")


def _bf16_report(res, g, name, out_mean, out_max, rel_bound, gnorm_bound, nonstem_rel_bound=None):
    """bf16 path against the fp32 golden: outputs (mean / max abs error on the unit-variance LayerNorm-ed maps), and for
    dx and EVERY parameter gradient the relative error of the stored subsample in the 2-norm plus the ratio of the full
    gradient norms."""
    worst_out = (0.0, 0.0)
    for i in range(4):
        ref = torch.from_numpy(g[f"out{i}_sub"])
        err = (res[f"out{i}_sub"] - ref).abs()
        worst_out = (max(worst_out[0], err.mean().item()), max(worst_out[1], err.max().item()))
    rel = rel_norm_errors(res, g)
    gn = {}
    for k in g.files:
        if k.startswith("gnorm:") and not any(z in k for z in ("proj.0.bias", "proj.3.bias", "k_linear.bias")):
            gn[k] = abs(float(res[k]) / max(float(g[k]), 1e-30) - 1.0)
    wk = max(rel, key=rel.get)
    wg = max(gn, key=gn.get)
    stem = {k: v for k, v in rel.items() if "patch_embed" in k}
    rest = {k: v for k, v in rel.items() if "patch_embed" not in k}
    wr = max(rest, key=rest.get)
    record(name, out_mean_abs=worst_out[0], out_max_abs=worst_out[1], worst_rel=rel[wk], worst_rel_key=wk,
           worst_nonstem_rel=rest[wr], worst_nonstem_key=wr,
           worst_gnorm_dev=gn[wg], worst_gnorm_key=wg, median_rel=sorted(rel.values())[len(rel) // 2],
           worst_stem_rel=max(stem.values()) if stem else 0.0, n_keys=len(rel))
    assert worst_out[0] < out_mean and worst_out[1] < out_max, worst_out
    bad = {k: v for k, v in rel.items() if v > rel_bound}
    assert not bad, sorted(bad.items(), key=lambda t: -t[1])[:10]
    if nonstem_rel_bound is not None:             # everything behind the stem (attention, MLP, norms, merging, tables)
        bad = {k: v for k, v in rest.items() if v > nonstem_rel_bound}
        assert not bad, sorted(bad.items(), key=lambda t: -t[1])[:10]
    badn = {k: v for k, v in gn.items() if v > gnorm_bound}
    assert not badn, sorted(badn.items(), key=lambda t: -t[1])[:10]
    assert len(rel) > 150


def test_T_512x1024_bf16_outputs_and_gradients():
    """The benched precision (bf16 GEMM / attention / stem operands, fp32 accumulation, fp32 residual stream and master
    weights) at full PanoSwin-T size against the fp32 golden of the live reference: outputs AND dx AND every parameter
    gradient (fused stem, fused fc1+GELU recompute, bias gradients routed through the LayerNorm backward kernel).
    Bounds = about 1.5x the measured worst case (gpurun_out/parity_report.json, key T_512x1024_bf16)."""
    m = _model(TCFG, True, "T", compute_dtype=torch.bfloat16)
    res = run_and_collect(m, (2, 3, 512, 1024), "T", device=DEV, subsample_out=4096)
    # measured: outputs mean 5.6e-3 / max 3.2e-2; worst tensor patch_embed.proj.1.bias (BatchNorm-1 shift) 0.139, median
    # over the 198 tensors 7e-3; gradient norms within 6.3 %
    _bf16_report(res, golden("T_512x1024_pano"), "T_512x1024_bf16", out_mean=1e-2, out_max=0.06, rel_bound=0.2,
                 gnorm_bound=0.1, nonstem_rel_bound=0.1)


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16])
def test_T_512x1024_eval_forward(cd):
    """BASELINE.json configs[0]: PanoSwin-T forward only, batch 2, 512x1024, eval mode (the fused stem's running-statistics
    branch at full size in bf16)."""
    m = _model(TCFG, True, "T", train=False, compute_dtype=cd)
    g = golden("T_512x1024_pano_eval")
    with torch.no_grad():
        outs = m(model_inputs((2, 3, 512, 1024), "T").to(DEV))
    worst_mean = worst_max = 0.0
    for i, o in enumerate(outs):
        err = (sub(o, 4096)[0].cpu() - torch.from_numpy(g[f"out{i}_sub"])).abs()
        worst_mean, worst_max = max(worst_mean, err.mean().item()), max(worst_max, err.max().item())
    record(f"T_512x1024_eval_{'fp32' if cd == torch.float32 else 'bf16'}", out_mean_abs=worst_mean, out_max_abs=worst_max)
    if cd == torch.float32:
        assert worst_max < 5e-4, worst_max
    else:
        assert worst_mean < 2e-2 and worst_max < 0.25, (worst_mean, worst_max)


def test_S_1024x2048_fp32():
    """BASELINE.json configs[4] geometry: PanoSwin-S (depths 2-2-18-2) on a 1024x2048 panorama, fp32, outputs + dx +
    every parameter gradient against the live reference's capture."""
    m = _model(SCFG, True, "S")
    res = run_and_collect(m, (1, 3, 1024, 2048), "S", device=DEV, subsample_out=4096)
    compare_to_golden(res, golden("S_1024x2048_pano"), rtol=1e-4, atol=5e-4, grad_rtol=2e-3, grad_atol_frac=1e-3,
                      dx_tol=(1e-2, 5e-3), report="S_1024x2048_fp32")


def test_S_1024x2048_bf16_outputs_and_gradients():
    """configs[4] in its stated precision (bf16) against the fp32 golden."""
    m = _model(SCFG, True, "S", compute_dtype=torch.bfloat16)
    res = run_and_collect(m, (1, 3, 1024, 2048), "S", device=DEV, subsample_out=4096)
    # measured: outputs mean 6.5e-3 / max 3.1e-2; worst tensor 0.091 (patch_embed.proj.1.bias), median 8e-3; norms within 3 %
    _bf16_report(res, golden("S_1024x2048_pano"), "S_1024x2048_bf16", out_mean=1.2e-2, out_max=0.06, rel_bound=0.15,
                 gnorm_bound=0.06, nonstem_rel_bound=0.1)


def test_batch_8_equals_four_batches_of_2_in_eval_mode():
    """The B = 8 workload of configs[1] is benched, the goldens hold B = 2: with BatchNorm in eval mode images are
    independent, so one batch of 8 must equal four batches of 2 (fp32: same kernels per window, bit for bit except the
    batch-looped attention items; bound 1e-5)."""
    m = _model(TCFG, True, "T", train=False)
    x = model_inputs((8, 3, 512, 1024), "B8").to(DEV)
    with torch.no_grad():
        full = m(x)
        parts = [m(x[i:i + 2]) for i in range(0, 8, 2)]
    for lvl in range(4):
        cat = torch.cat([p[lvl] for p in parts], 0)
        assert torch.allclose(full[lvl], cat, rtol=1e-5, atol=1e-5), lvl


def test_interface_and_state_dict():
    import panoswintransformerobjectdetection_amd as pkg
    assert "SimplePanoSwinTransformer" in pkg.BACKBONES.module_dict
    m = pkg.build_backbone(dict(type="SimplePanoSwinTransformer", **TINY)).to(DEV)
    m.init_weights(None)
    assert m.eval() is m
    with torch.no_grad():
        outs = m(torch.randn(1, 3, 64, 128, device=DEV), None)
    assert [tuple(o.shape) for o in outs] == [(1, 32, 16, 32), (1, 64, 8, 16), (1, 128, 4, 8), (1, 256, 2, 4)]
    assert all(o.dtype == torch.float32 and o.is_contiguous() for o in outs)
    with pytest.raises(TypeError):
        m.init_weights(pretrained=1)
    with pytest.raises(pkg.PswinError):
        m(torch.randn(1, 3, 64, 128))           # CPU input: no fallback


def test_drop_path_and_checkpoint_run():
    import panoswintransformerobjectdetection_amd as pkg
    cfg = dict(TINY, drop_path_rate=0.3, use_checkpoint=True)
    m = pkg.SimplePanoSwinTransformer(**cfg).to(DEV).train()
    m.init_weights(None)
    x = torch.randn(2, 3, 64, 128, device=DEV, requires_grad=True)
    sum(o.mean() for o in m(x)).backward()
    assert torch.isfinite(x.grad).all()


def test_deferred_reductions_with_activation_checkpointing():
    """use_checkpoint=True makes every block's backward a nested autograd pass; each pass must flush its own queued
    reductions (the queue is keyed by graph task): same parameter gradients as the immediate mode, bit for bit."""
    import panoswintransformerobjectdetection_amd as pkg
    from panoswintransformerobjectdetection_amd import ops
    cfg = dict(TINY, drop_path_rate=0.0, use_checkpoint=True)
    torch.manual_seed(0)
    m = pkg.SimplePanoSwinTransformer(**cfg, compute_dtype=torch.bfloat16).to(DEV).train()
    m.init_weights(None)
    x = torch.randn(2, 3, 64, 128, device=DEV)

    def grads():
        for p in m.parameters():
            p.grad = None
        sum(o.float().square().mean() for o in m(x)).backward()
        return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    ref = grads()
    prev = ops.set_deferred_reductions(True)
    try:
        got = grads()
    finally:
        ops.set_deferred_reductions(prev)
    assert not ops._ReduceQueue.tasks
    assert ref.keys() == got.keys() and len(ref) > 50
    for k in ref:
        if k.startswith("patch_embed"):      # this small stem runs on library convolutions (not run-to-run bitwise stable)
            assert torch.allclose(ref[k], got[k], rtol=1e-3, atol=1e-3 * float(ref[k].abs().max())), k
        else:
            assert torch.equal(ref[k], got[k]), k


def test_hipgraph_replay_matches_eager():
    """A captured training step (forward + backward of PanoSwin-T, bf16, fused stem) must replay bit-identically to the
    eager step: every output and every parameter gradient, on the first and on later replays (the path contains no
    framework two-pass reduction, whose result goes stale from the second replay of a hipGraph on this stack)."""
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
    from panoswintransformerobjectdetection_amd.graph import GraphedCallable
    cfg = dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, ape=True, drop_path_rate=0.0,
               pano_mode=True)
    torch.manual_seed(0)
    m = SimplePanoSwinTransformer(**cfg, compute_dtype=torch.bfloat16)
    m.init_weights(None)
    m = m.cuda().train()
    x = torch.randn(2, 3, 256, 512, device="cuda")
    with torch.no_grad():
        ws = [torch.randn_like(o).flatten() / o.numel() for o in m(x)]

    def step():
        for p in m.parameters():
            p.grad = None
        outs = m(x)
        loss = sum(o.float().flatten() @ w for o, w in zip(outs, ws))
        loss.backward()
        return [loss] + list(outs)

    ref = [t.detach().clone() for t in step()]
    gref = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    assert all(torch.isfinite(g).all() for g in gref.values())
    # the captured step also postpones the ~120 parameter-gradient reductions to ONE grouped launch at the end of the
    # backward pass (ops.set_deferred_reductions); same kernel, same summation order: still bit-identical
    from panoswintransformerobjectdetection_amd import ops
    prev = ops.set_deferred_reductions(True)
    try:
        eager_deferred = [t.detach().clone() for t in step()]
        for k, p in m.named_parameters():
            assert torch.equal(p.grad, gref[k]), k
        g = GraphedCallable(step, warmup=2)
        for _ in range(3):
            out = g()
            torch.cuda.synchronize()
            for a, b in zip(out, ref):
                assert torch.equal(a, b)
            for k, p in m.named_parameters():
                assert torch.equal(p.grad, gref[k]), k
    finally:
        ops.set_deferred_reductions(prev)
    assert not ops._ReduceQueue.tasks
    del eager_deferred


def test_weight_gradients_land_in_the_flat_gradient_buffer():
    """dp.GradReducer(pack=True) + deferred reductions: the split-K weight gradients of the Linear layers are summed
    straight into their slots of the flat gradient buffer (ops.grad_slot), pack_grads copies only the rest, and the
    buffer holds exactly the gradients of a plain backward pass."""
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer, ops
    from panoswintransformerobjectdetection_amd.dp import GradReducer
    cfg = dict(embed_dim=96, depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24], window_size=7, ape=True, drop_path_rate=0.0,
               pano_mode=True)
    torch.manual_seed(0)
    m = SimplePanoSwinTransformer(**cfg, compute_dtype=torch.bfloat16)
    m.init_weights(None)
    m = m.cuda().train()
    x = torch.randn(2, 3, 256, 512, device="cuda")
    with torch.no_grad():
        ws = [torch.randn_like(o).flatten() / o.numel() for o in m(x)]

    def backward():
        sum(o.float().flatten() @ w for o, w in zip(m(x), ws)).backward()

    backward()
    ref = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    red = GradReducer(m, pack=True)
    prev = ops.set_deferred_reductions(True)
    try:
        for _ in range(2):                                   # twice: the slots are handed out again every pass
            red.zero_grad()
            backward()
            aliased = [k for k, p in m.named_parameters() if p.grad is not None and p.grad.data_ptr() == p._grad_slot.data_ptr()]
            assert len(aliased) >= 20 and all(k.endswith("weight") for k in aliased), aliased
            red.pack_grads()
            red.finish()
            for k, p in m.named_parameters():
                assert p.grad.data_ptr() == p._grad_slot.data_ptr()
                assert torch.equal(p.grad, ref[k]), k
    finally:
        ops.set_deferred_reductions(prev)


def test_two_piece_backward_equals_plain_backward():
    """dp.backward_late + dp.backward_early (the step that overlaps the gradient all-reduce with the early layers'
    backward) produce the gradients of one plain backward pass."""
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
    from panoswintransformerobjectdetection_amd.dp import BoundaryTap, backward_early, backward_late, split_parameters
    cfg = dict(embed_dim=96, depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24], window_size=7, ape=True, drop_path_rate=0.0,
               pano_mode=True)
    torch.manual_seed(0)
    m = SimplePanoSwinTransformer(**cfg, compute_dtype=torch.bfloat16)
    m.init_weights(None)
    m = m.cuda().train()
    x = torch.randn(2, 3, 128, 256, device="cuda")
    with torch.no_grad():
        ws = [torch.randn_like(o).flatten() / o.numel() for o in m(x)]
    outs = m(x)
    sum(o.float().flatten() @ w for o, w in zip(outs, ws)).backward()
    ref = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
    late, early = split_parameters(m, ("layers.2.", "layers.3.", "norm2.", "norm3."))
    assert late and early and len(late) + len(early) == len(ref)
    tap = BoundaryTap(m.layers[2])
    outs = m(x)
    l_early = sum(o.float().flatten() @ w for o, w in zip(outs[:2], ws[:2]))
    l_late = sum(o.float().flatten() @ w for o, w in zip(outs[2:], ws[2:]))
    g_xb = backward_late(l_late, tap.x, late)
    assert all(p.grad is None for p in early)
    backward_early(l_early, tap.x, g_xb, early)
    tap.remove()
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        assert torch.allclose(p.grad, ref[k], rtol=1e-5, atol=1e-7 + 1e-5 * float(ref[k].abs().max())), k


def test_deferred_reductions_stay_correct_when_a_result_is_read_before_the_pass_ends():
    """ops.set_deferred_reductions(True) postpones a parameter-gradient reduction only when nothing can read it before
    the pass ends.  Three cases that would otherwise add or clone unwritten memory: a Linear applied twice in one pass
    (autograd adds the two gradients when the second arrives), accumulation into an existing .grad over two passes, and a
    parameter with a post-accumulate hook (what dp.GradReducer(pack=False) registers)."""
    from panoswintransformerobjectdetection_amd import ops
    from panoswintransformerobjectdetection_amd.backbone import _linear
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 64).to(DEV)
    ln = torch.nn.LayerNorm(64).to(DEV)
    x = torch.randn(2, 8192, 64, device=DEV)

    def net():
        h = ops.layer_norm_gather(x, ln.weight, ln.bias, ln.eps, out_dtype=torch.bfloat16)
        h = _linear(h, lin, torch.bfloat16)
        h = ops.layer_norm_gather(h.float(), ln.weight, ln.bias, ln.eps, out_dtype=torch.bfloat16)   # LayerNorm twice
        return _linear(h, lin, torch.bfloat16).float().square().mean()                              # Linear twice

    params = [lin.weight, lin.bias, ln.weight, ln.bias]

    def run(passes, hook):
        for p in params:
            p.grad = None
        seen = []
        hs = [p.register_post_accumulate_grad_hook(lambda q: seen.append(q.grad.detach().clone())) for p in params] if hook else []
        for _ in range(passes):
            net().backward()
        for h in hs:
            h.remove()
        return [p.grad.detach().clone() for p in params], seen

    for passes, hook in ((1, False), (2, False), (1, True)):
        prev = ops.set_deferred_reductions(False)
        ref, ref_seen = run(passes, hook)
        ops.set_deferred_reductions(True)
        try:
            got, got_seen = run(passes, hook)
        finally:
            ops.set_deferred_reductions(prev)
        assert not ops._ReduceQueue.tasks
        for a, b in zip(ref, got):
            assert torch.isfinite(b).all() and torch.equal(a, b), (passes, hook)
        for a, b in zip(ref_seen, got_seen):        # what a hook sees at accumulation time is already the final value
            assert torch.equal(a, b), (passes, hook)


@pytest.mark.gpu
def test_training_steps_with_the_flat_hip_adamw_match_torch_adamw():
    """Three bf16 training steps of a small PanoSwin with optim.FlatAdamW (the update and the bf16 weight copy in one HIP launch; the
    forward pass no longer refreshes the copy) against the same steps with torch.optim.AdamW on the flat parameter and the
    per-forward refresh: same parameters (up to Adam's sign noise on near-zero gradients), same outputs of the fourth forward pass."""
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
    from panoswintransformerobjectdetection_amd.dp import GradReducer
    from panoswintransformerobjectdetection_amd.optim import FlatAdamW
    cfg = dict(embed_dim=96, depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24], window_size=7, ape=True, drop_path_rate=0.0, pano_mode=True)
    x = torch.randn(2, 3, 128, 256, device="cuda", generator=torch.Generator("cuda").manual_seed(3))

    def run(native):
        torch.manual_seed(0)
        m = SimplePanoSwinTransformer(**cfg, compute_dtype=torch.bfloat16)
        m.init_weights(None)
        m = m.cuda().train()
        red = GradReducer(m, pack=True)
        flat = red.flatten_parameters(m, torch.bfloat16)
        kw = dict(lr=1e-3, betas=(0.9, 0.999), weight_decay=0.05)
        opt = FlatAdamW(flat, model=m, **kw) if native else torch.optim.AdamW([flat], **kw)
        assert bool(m.__dict__.get("_lowp_external")) == native
        with torch.no_grad():
            ws = [torch.randn_like(o, generator=None).flatten() for o in m(x)]
            ws = [torch.linspace(-1, 1, w.numel(), device="cuda") / w.numel() for w in ws]
        for _ in range(3):
            red.zero_grad()
            sum(o.float().flatten() @ w for o, w in zip(m(x), ws)).backward()
            red.pack_grads()
            red.finish()
            opt.step()
        with torch.no_grad():
            outs = [o.float() for o in m(x)]
        return flat.data.clone(), outs

    pa, oa = run(True)
    pb, ob = run(False)
    # Adam's update is ~ lr * sign(gradient) wherever a gradient is noise around zero, and the two runs' gradients differ in the last
    # bits from the second step on (their parameters differ by the f32 rounding of the two update kernels): a few elements move by
    # up to 2 lr per step in opposite directions, everything else agrees to rounding
    d = (pa - pb).abs()
    assert float(d.max()) <= 3 * 2 * 1e-3 * 1.01, d.max()
    assert float((d > 1e-5).float().mean()) < 0.02, (d > 1e-5).float().mean()
    for a, b in zip(oa, ob):
        assert torch.allclose(a, b, rtol=2e-2, atol=2e-2), (a - b).abs().max()


def test_benched_workload_is_bitwise_reproducible():
    """BASELINE configs[1] at full size (PanoSwin-T, batch 8, 3x512x1024, bf16, train mode): two forward + backward passes on the
    same input give bit-identical outputs and parameter gradients -- every reduction of the path (split-K partial sums, column
    sums, table gradients, LayerNorm partials, the attention kernels' dS sums) runs in a fixed order, none uses atomics.  The
    library's batched GEMMs and the shipped TunableOp table are part of that claim."""
    m = _model(TCFG, True, "T", compute_dtype=torch.bfloat16)
    x = model_inputs((8, 3, 512, 1024), "B8").to(DEV)
    with torch.no_grad():
        ws = [torch.linspace(-1, 1, o.numel(), device=DEV) / o.numel() for o in m(x)]

    def run():
        for p in m.parameters():
            p.grad = None
        outs = m(x)
        sum(o.float().flatten() @ w for o, w in zip(outs, ws)).backward()
        return [o.detach().clone() for o in outs], {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    o1, g1 = run()
    o2, g2 = run()
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    assert g1.keys() == g2.keys() and len(g1) > 150
    diff = [k for k in g1 if not torch.equal(g1[k], g2[k])]
    assert not diff, diff[:8]
    assert all(bool(torch.isfinite(v).all()) for v in g1.values())


@pytest.mark.gpu
def test_weights_changed_behind_the_flat_optimizer_reach_the_bf16_kernels():
    """once optim.FlatAdamW keeps the bf16 weight shadow fresh, the forward pass stops re-casting it -- so a weight change
    that does NOT go through its step() (load_state_dict after the optimizer is built, init_weights, a p.data write announced with
    mark_weights_changed(), the optimizer being dropped) must be noticed by the next forward pass."""
    import gc
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
    from panoswintransformerobjectdetection_amd.dp import GradReducer
    from panoswintransformerobjectdetection_amd.optim import FlatAdamW
    cfg = dict(embed_dim=96, depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24], window_size=7, ape=True, drop_path_rate=0.0, pano_mode=True)
    x = torch.randn(2, 3, 128, 256, device="cuda", generator=torch.Generator("cuda").manual_seed(3))

    def fresh(seed):
        torch.manual_seed(seed)
        m = SimplePanoSwinTransformer(**cfg, compute_dtype=torch.bfloat16)
        m.init_weights(None)
        return m.cuda().eval()

    def fwd(m):
        with torch.no_grad():
            return [o.clone() for o in m(x)]

    other = fresh(1)
    want = fwd(other)
    m = fresh(0)
    red = GradReducer(m, pack=True)
    flat = red.flatten_parameters(m, torch.bfloat16)
    opt = FlatAdamW(flat, lr=1e-3, model=m)
    before = fwd(m)
    assert not torch.equal(before[0], want[0])
    m.load_state_dict(other.state_dict())                      # (1) tracked by the parameters' version counters
    for a, b in zip(fwd(m), want):
        assert torch.equal(a, b)
    with torch.no_grad():                                      # (2) a raw write + the explicit notice
        for p, q in zip(m.parameters(), fresh(0).parameters()):
            p.data.copy_(q.data)
    m.mark_weights_changed()
    for a, b in zip(fwd(m), before):
        assert torch.equal(a, b)
    m.load_state_dict(other.state_dict())
    fwd(m)
    del opt                                                    # (3) optimizer gone: every forward pass refreshes again
    gc.collect()
    with torch.no_grad():
        for p, q in zip(m.parameters(), fresh(0).parameters()):
            p.data.copy_(q.data)
    for a, b in zip(fwd(m), before):
        assert torch.equal(a, b)
