"""Data-parallel path of the PRODUCT backbone on the MI355X box.

  * one process: dp.GradReducer(pack=True) + deferred reductions + the two-piece backward (what bench.py runs for N > 1)
    reproduce the oracle's local gradients of the tiny PanoSwin model;
  * two processes sharing the one GPU of the box (gloo transport over the device buffers: RCCL refuses two ranks on one
    device; the 8-rank RCCL run is the driver's): after finish() both ranks hold mean_over_ranks(local gradient) of every
    parameter, local gradients captured from the CPU oracle on the two input halves, BatchNorm statistics stay per rank
    (mmdet/apis/train.py:91-99, broadcast_buffers=False).
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist

from _util import TINY, ZERO_GRAD_KEYS, build_filled, loss_weights, model_inputs
from test_dp_gloo import _oracle_local_grads, _run_ranks

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _product_step(m, red, x, tag="dp"):
    """One training-step backward the way bench.py's two-graph step does it: late piece, pack + launch, early piece."""
    from panoswintransformerobjectdetection_amd.dp import BoundaryTap, backward_early, backward_late, split_parameters
    late, early = split_parameters(m)
    tap = BoundaryTap(m.layers[m.LATE_STAGES])
    red.zero_grad()
    outs = m(x)
    ws = [w.to(x.device) for w in loss_weights(outs, tag)]
    l_early = sum((o * w).sum() for o, w in zip(outs[:2], ws[:2]))
    l_late = sum((o * w).sum() for o, w in zip(outs[2:], ws[2:]))
    g_xb = backward_late(l_late, tap.x, late)
    red.pack_grads(late)
    red.launch(red.buckets_within(late))
    backward_early(l_early, tap.x, g_xb, early)
    red.pack_grads(early)
    tap.remove()
    red.finish()


def _check(got, want, what):
    for k, w in want.items():
        if any(z in k for z in ZERO_GRAD_KEYS):
            continue
        tol = 1e-3 * float(w.abs().max()) + 1e-7
        assert torch.allclose(got[k], w, rtol=2e-3, atol=tol), (what, k, float((got[k] - w).abs().max()), tol)


def test_pack_mode_two_piece_backward_matches_oracle_local_gradients():
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer, ops
    from panoswintransformerobjectdetection_amd.dp import GradReducer
    _, want = _oracle_local_grads(0)
    m = build_filled(SimplePanoSwinTransformer, TINY, True, "tiny").to(DEV)
    red = GradReducer(m, bucket_mb=0.05, pack=True)
    assert len(red.buckets_within(red.groups[0])) >= 1 and len(red.buckets) >= 3
    x = model_inputs((1, 3, 64, 128), "dp:rank0").to(DEV)
    prev = ops.set_deferred_reductions(True)
    try:
        for _ in range(2):
            _product_step(m, red, x)
    finally:
        ops.set_deferred_reductions(prev)
    got = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    assert all(p.grad.data_ptr() == p._grad_slot.data_ptr() for p in m.parameters())
    _check(got, want, "single process")


def _gpu_worker(rank, world, port, q):
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer, ops
    from panoswintransformerobjectdetection_amd.dp import GradReducer, init_distributed
    init_distributed(backend="gloo")
    torch.cuda.set_device(0)
    m = build_filled(SimplePanoSwinTransformer, TINY, True, "tiny").to(DEV)
    red = GradReducer(m, bucket_mb=0.05, pack=True)
    red.broadcast_parameters(m)
    x = model_inputs((1, 3, 64, 128), f"dp:rank{rank}").to(DEV)
    ops.set_deferred_reductions(True)
    for _ in range(2):
        _product_step(m, red, x)
    torch.cuda.synchronize()
    bn = dict(m.named_buffers())["patch_embed.proj.1.running_mean"]
    q.put((rank, {k: p.grad.cpu().numpy().copy() for k, p in m.named_parameters()}, len(red.buckets), bn.cpu().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_hold_the_mean_of_the_oracle_local_gradients():
    res = _run_ranks(_gpu_worker, (), timeout=600)
    _, g0 = _oracle_local_grads(0)
    _, g1 = _oracle_local_grads(1)
    want = {k: (g0[k] + g1[k]) / 2 for k in g0}
    for r in range(2):
        _check({k: torch.from_numpy(v) for k, v in res[r][1].items()}, want, f"rank {r}")
    for k in res[0][1]:                                                   # the ranks agree bit for bit
        assert (res[0][1][k] == res[1][1][k]).all(), k
    assert not torch.allclose(torch.from_numpy(res[0][3]), torch.from_numpy(res[1][3]))   # BN statistics stay per rank
