"""Data-parallel contract on CPU: 2 processes over gloo.  After finish() every rank holds the mean over ranks of the
local gradients for every parameter (including parameters that got no gradient) and BatchNorm buffers stay local
(reference: MMDistributedDataParallel with broadcast_buffers=False, mmdet/apis/train.py:91-99)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 8, 3, padding=1)
        self.bn = nn.BatchNorm2d(8)
        self.fc1 = nn.Linear(8, 16)
        self.fc2 = nn.Linear(16, 4)
        self.unused = nn.Parameter(torch.ones(5))        # never touched by forward

    def forward(self, x):
        y = torch.relu(self.bn(self.conv(x))).mean((2, 3))
        return self.fc2(torch.tanh(self.fc1(y)))


def _inputs(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(4, 3, 8, 8, generator=g)


def _local_grads(rank):
    torch.manual_seed(0)
    m = _Net()
    m(_inputs(rank)).square().mean().backward()
    return {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}, m


def _worker(rank, world, port, q, pack=False, two_piece=False, set_to_none=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from panoswintransformerobjectdetection_amd.dp import GradReducer, init_distributed
    r, _, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0 if rank == 0 else 77)            # different initial weights: broadcast must fix that
    m = _Net()
    red = GradReducer(m, bucket_mb=0.0005, pack=pack)    # tiny buckets -> several collectives
    red.broadcast_parameters(m)
    opt = torch.optim.SGD(m.parameters(), lr=0.0)        # lr 0: the weights stay put, zero_grad() is what is tested
    for _ in range(2):                                   # two steps: hooks / counters must re-arm
        if set_to_none:
            # what mmcv's OptimizerHook does every iteration (torch 2 default set_to_none=True): param.grad stops being a
            # view of the flat buffer; the reducer must move the fresh gradients back in before it reduces
            opt.zero_grad(set_to_none=True)
            assert all(p.grad is None for p in m.parameters())
        else:
            red.zero_grad()
        if two_piece:                                    # backward in two pieces, late buckets reduced in between
            from panoswintransformerobjectdetection_amd.dp import BoundaryTap, backward_early, backward_late, split_parameters
            late, early = split_parameters(m, ("fc1.", "fc2."))
            tap = BoundaryTap(m.fc1)
            loss = m(_inputs(rank)).square().mean()
            g_xb = backward_late(loss, tap.x, late)
            red.pack_grads(late)
            red.launch(red.buckets_within(late))
            backward_early(loss.detach() * 0 + tap.x.sum() * 0, tap.x, g_xb, early)   # no loss term before the cut
            red.pack_grads(early)
            tap.remove()
        else:
            m(_inputs(rank)).square().mean().backward()
            if pack:
                red.pack_grads()
        red.finish()
    q.put((rank, {k: p.grad.numpy().copy() for k, p in m.named_parameters()}, len(red.buckets),
           m.bn.running_mean.numpy().copy()))                      # numpy: pickled by value, no fd passing
    dist.barrier()
    dist.destroy_process_group()


import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(target, args, world=2, timeout=300):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=timeout) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("pack,two_piece,set_to_none", [(False, False, False), (True, False, False), (True, True, False),
                                                        (False, False, True)])
def test_two_rank_gradient_average(pack, two_piece, set_to_none):
    world = 2
    res = _run_ranks(_worker, (pack, two_piece, set_to_none))
    g0, _ = _local_grads(0)
    g1, _ = _local_grads(1)
    assert res[0][2] > 1                                  # more than one bucket was exercised
    for k in g0:
        want = (g0[k] + g1[k]) / 2
        for r in range(world):
            assert torch.allclose(torch.from_numpy(res[r][1][k]), want, rtol=1e-5, atol=1e-7), (k, r)
    assert torch.equal(torch.from_numpy(res[0][1]["unused"]), torch.zeros(5))
    assert not torch.allclose(torch.from_numpy(res[0][3]), torch.from_numpy(res[1][3]))   # BN stats stay per rank


def test_flatten_parameters_single_tensor_adamw_matches_per_parameter_adamw():
    """dp.GradReducer.flatten_parameters: parameters become views of one flat buffer; AdamW on that single tensor
    (with the flat gradient buffer as its .grad) takes the same steps as AdamW over the individual parameters."""
    import copy
    import torch
    from panoswintransformerobjectdetection_amd.dp import GradReducer
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.LayerNorm(5), torch.nn.Linear(5, 3))
    mod = copy.deepcopy(ref)
    red = GradReducer(mod, pack=True)
    flat = red.flatten_parameters()
    for p, q in zip(mod.parameters(), ref.parameters()):
        assert torch.equal(p.data, q.data)
        assert p.data.data_ptr() >= flat.data_ptr() and p.data.data_ptr() < flat.data_ptr() + flat.numel() * 4
    o_ref = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.05)
    o_mod = torch.optim.AdamW([flat], lr=1e-2, weight_decay=0.05)
    x = torch.randn(11, 7)
    for _ in range(3):
        o_ref.zero_grad()
        ref(x).square().mean().backward()
        o_ref.step()
        red.zero_grad()
        mod(x).square().mean().backward()
        red.pack_grads()
        red.finish()
        o_mod.step()
    for p, q in zip(mod.parameters(), ref.parameters()):
        assert torch.allclose(p.data, q.data, rtol=1e-6, atol=1e-7)


# ---- the pin of SURVEY.md section 8a row 17: the tiny PanoSwin model, two input halves, oracle-captured local gradients ----
def _oracle_local_grads(rank, steps=1):
    """Local gradients of the tiny pano-mode model (tests/_util.TINY, deterministic fill) on rank `rank`'s input half,
    from the CPU oracle (pinned against the live reference by tests/test_oracle_vs_reference.py)."""
    import panoswin_oracle as po
    from _util import TINY, build_filled, loss_weights, model_inputs
    m = build_filled(po.SimplePanoSwinTransformerOracle, TINY, True, "tiny")
    x = model_inputs((1, 3, 64, 128), f"dp:rank{rank}")
    outs = m(x)
    sum((o * w).sum() for o, w in zip(outs, loss_weights(outs, "dp"))).backward()
    return m, {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}


def _tiny_worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import panoswin_oracle as po
    from _util import TINY, build_filled, loss_weights, model_inputs
    from panoswintransformerobjectdetection_amd.dp import GradReducer, init_distributed
    init_distributed(backend="gloo")
    m = build_filled(po.SimplePanoSwinTransformerOracle, TINY, True, "tiny")
    # the oracle model has no grad_groups(): name the two groups the product defines (late = stages 2-3 + their norms)
    named = list(m.named_parameters())
    late = [p for n, p in reversed(named) if n.startswith(("layers.2.", "layers.3.", "norm2.", "norm3."))]
    early = [p for n, p in reversed(named) if not n.startswith(("layers.2.", "layers.3.", "norm2.", "norm3."))]
    red = GradReducer(m, bucket_mb=0.05, groups=[late, early])           # hook mode: buckets launched during backward
    red.broadcast_parameters(m)
    x = model_inputs((1, 3, 64, 128), f"dp:rank{rank}")
    for step in range(2):
        m.zero_grad(set_to_none=(step == 1))
        outs = m(x)
        sum((o * w).sum() for o, w in zip(outs, loss_weights(outs, "dp"))).backward()
        red.finish()
    bn = dict(m.named_buffers())["patch_embed.proj.1.running_mean"]
    q.put((rank, {k: p.grad.numpy().copy() for k, p in m.named_parameters()}, len(red.buckets), bn.numpy().copy(),
           len(red.buckets_within(late)), len(red.buckets_within(early))))
    dist.barrier()
    dist.destroy_process_group()


def test_tiny_panoswin_two_ranks_match_mean_of_oracle_local_gradients():
    """after the reducer every rank holds mean_over_ranks(local grad) of EVERY backbone parameter, local
    gradients captured from the oracle on the two input halves; BatchNorm running statistics stay per rank
    (mmdet/apis/train.py:91-99: MMDistributedDataParallel(broadcast_buffers=False))."""
    res = _run_ranks(_tiny_worker, ())
    _, g0 = _oracle_local_grads(0)
    _, g1 = _oracle_local_grads(1)
    assert res[0][2] >= 3 and res[0][4] >= 1 and res[0][5] >= 1 and res[0][4] + res[0][5] == res[0][2]
    assert len(g0) > 100
    from _util import ZERO_GRAD_KEYS
    for k in g0:
        if any(z in k for z in ZERO_GRAD_KEYS):      # true gradient 0 (a bias in front of BatchNorm): rounding noise that
            continue                                 # depends on the thread count of the convolution
        want = (g0[k] + g1[k]) / 2
        tol = 1e-6 + 2e-5 * float(want.abs().max())
        for r in range(2):
            assert torch.allclose(torch.from_numpy(res[r][1][k]), want, rtol=1e-4, atol=tol), (k, r)
    assert not torch.allclose(torch.from_numpy(res[0][3]), torch.from_numpy(res[1][3]))


def test_panoswin_t_late_buckets_hold_90_percent_of_the_gradient_bytes():
    """The overlap claim of DESIGN.md section 6 as an assertion: with the product's grad_groups() the buckets that hold
    only late-group (stages 2-3) gradients cover >= 90 % of all gradient bytes of PanoSwin-T, every bucket is pure, and
    in hook mode no late bucket waits for an early-stage parameter."""
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
    from panoswintransformerobjectdetection_amd.dp import GradReducer, split_parameters
    m = SimplePanoSwinTransformer(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], ape=True)
    names = {id(p): n for n, p in m.named_parameters()}
    for pack in (True, False):
        red = GradReducer(m, bucket_mb=32.0, pack=pack)
        late, early = split_parameters(m)
        assert all(names[id(p)].startswith(("layers.2.", "layers.3.", "norm2.", "norm3.")) for p in late)
        assert not any(names[id(p)].startswith(("layers.2.", "layers.3.", "norm2.", "norm3.")) for p in early)
        lb, eb = red.buckets_within(late), red.buckets_within(early)
        assert sorted(lb + eb) == list(range(len(red.buckets)))            # no bucket straddles the cut
        late_bytes = sum(red.buckets[b][1] - red.buckets[b][0] for b in lb) * 4
        total = red.flat.numel() * 4
        assert late_bytes >= 0.9 * total, (late_bytes, total)
        assert max(lb) < min(eb)                                           # late buckets come first in the flat buffer
        for p in m.parameters():
            p.grad = None


def _loss_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from panoswintransformerobjectdetection_amd.dp import init_distributed, reduce_loss_scalars
    init_distributed(backend="gloo")
    local = {"loss_cls": torch.tensor(1.0 + rank), "loss_bbox": torch.tensor(0.25 * (rank + 1)), "loss_mask": torch.tensor([3.0 - rank]),
             "acc": torch.tensor(90.0 + 4 * rank, dtype=torch.float64)}
    out = reduce_loss_scalars(local)
    q.put((rank, {k: float(v) for k, v in out.items()}, list(out)))
    dist.barrier()
    dist.destroy_process_group()


def test_logged_loss_scalars_are_averaged_with_one_collective():
    """BaseDetector._parse_losses (mmdet/models/detectors/base.py:213-218: one all-reduce per log variable, divided by the world size)
    as one small all-reduce: every rank ends with the mean over ranks of every scalar."""
    res = _run_ranks(_loss_worker, ())
    want = {"acc": 92.0, "loss_bbox": 0.375, "loss_cls": 1.5, "loss_mask": 2.5}
    for rank, got, keys in res:
        assert keys == sorted(want) and got == pytest.approx(want)
    from panoswintransformerobjectdetection_amd.dp import reduce_loss_scalars
    one = reduce_loss_scalars({"b": torch.tensor(2.0), "a": torch.tensor(1.0)})             # no process group: identity
    assert list(one) == ["a", "b"] and float(one["a"]) == 1.0 and float(one["b"]) == 2.0


def _nested_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    from _util import TINY
    from panoswintransformerobjectdetection_amd import SimplePanoSwinTransformer
    from panoswintransformerobjectdetection_amd.dp import GradReducer, init_distributed
    init_distributed(backend="gloo")

    class Wrapper(torch.nn.Module):                  # the backbone nested in a detector-like holder (MiniMaskRCNN's shape)
        def __init__(self):
            super().__init__()
            torch.manual_seed(rank)
            self.backbone = SimplePanoSwinTransformer(**TINY, pano_mode=True)
            self.head = torch.nn.Linear(4, 4)

    w = Wrapper()
    before = w.backbone.__dict__.get("_lowp_epoch", 0)
    red = GradReducer(w)
    red.broadcast_parameters(w)
    same = float(sum(p.double().sum() for p in w.parameters()))
    q.put((rank, before, w.backbone.__dict__.get("_lowp_epoch", 0), same))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_through_a_wrapper_marks_the_nested_backbone_weights_changed():
    """broadcast_parameters writes p.data of a backbone nested in a wrapper; the backbone's bf16 weight shadows are refreshed
    on the next forward pass only if ITS epoch is bumped (mark_weights_changed on every submodule that has it, not on the wrapper)."""
    res = sorted(_run_ranks(_nested_worker, ()))
    assert [r[1] for r in res] == [0, 0] and [r[2] for r in res] == [1, 1]
    assert res[0][3] == res[1][3]                    # and the broadcast itself made the ranks equal
