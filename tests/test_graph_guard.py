"""The one-stream guard of graph.GraphedCallable / GraphedSequence: a capture is refused with a Python exception
when a parameter's AccumulateGrad node runs on a stream other than the capture stream.  CPU part: the guard's bookkeeping with an
injected "current stream" (the engine has no streams on CPU); the GPU part is tests/test_graph_guard_gpu.py."""
import pytest
import torch

from panoswintransformerobjectdetection_amd._lib import PswinError
from panoswintransformerobjectdetection_amd.graph import AccumulateStreamGuard, _as_parameters


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))


def test_guard_passes_when_every_accumulate_node_runs_on_the_capture_stream():
    m = _model()
    guard = AccumulateStreamGuard(_as_parameters([m]), current_stream=lambda: "side")
    m(torch.randn(5, 4)).sum().backward()
    guard.disarm()
    assert len(guard.seen) == 4 and set(guard.seen.values()) == {"side"}
    guard.check("side")


def test_guard_raises_a_python_exception_that_names_the_rule():
    m = _model()
    where = {"s": "default"}
    guard = AccumulateStreamGuard(_as_parameters([m]), current_stream=lambda: where["s"])
    m(torch.randn(5, 4)).sum().backward()
    guard.disarm()
    with pytest.raises(PswinError) as e:
        guard.check("side", names={id(p): n for n, p in m.named_parameters()})
    msg = str(e.value)
    assert "ONE stream" in msg and "AccumulateGrad" in msg and "4 of 4" in msg and "1.bias" in msg        # the first gradient a backward pass finishes


def test_guard_keeps_its_nodes_alive_and_removes_its_hooks():
    m = _model()
    guard = AccumulateStreamGuard(_as_parameters([m]), current_stream=lambda: "side")
    p = m[0].weight
    assert p.view_as(p).grad_fn.next_functions[0][0] is guard.nodes[0]          # the same node is reused while the guard lives
    guard.disarm()
    m(torch.randn(5, 4)).sum().backward()
    assert guard.seen == {}                                                      # disarmed: nothing recorded


def test_parameter_selection():
    m = _model()
    m[1].weight.requires_grad_(False)
    ps = _as_parameters([m, m[0].weight])
    assert len(ps) == 3 and all(p.requires_grad for p in ps)
    with pytest.raises(TypeError):
        _as_parameters([3])
