"""HIP-graph capture of a fixed-shape training step.

An eager PanoSwin-T step is ~1100 kernel launches; on MI355X the launches, not the kernels, then set the step time
(measured: 25 ms of GPU work in a 33-48 ms eager step).  Capturing the step once into a hipGraph and replaying it
removes the per-launch host cost.  Requirements on the captured callable: static shapes, no host synchronisation
(`.item()`, prints of device values), every lazily built table already cached (run it a few times first -- that is
what `warmup` does), optimizers constructed with ``capturable=True``.  The C-ABI kernels need nothing special: they
are launched on torch's current stream, which is the capturing stream during capture.

The one-stream rule (and the guard that enforces it).  autograd's AccumulateGrad node of a parameter is created the first
time the parameter takes part in a recorded forward pass and REMEMBERS the stream that was current then; it stays alive -- and
is reused by later passes -- for as long as anything references the graph it belongs to (a `loss` / output tensor of an earlier
eager step that is still in scope, a DDP-style hook holder).  A backward pass captured on a different stream then runs that node
on ITS stream: the engine makes that stream wait on an event of the capturing stream, which pulls it into the capture, and when it
is the legacy default stream ``hipStreamEndCapture`` does not return an error, it crashes the process (round 3:
gpurun_out/r3a_1.log, r3b_3.log, r3c_2.log, r3c_5.log; with a non-default foreign stream the symptom was garbage gradients from
the second replay on).  torch only warns, once per process ("The AccumulateGrad node's stream does not match ...").
``GraphedCallable`` / ``GraphedSequence`` therefore never reach ``capture_end`` in that state: during the first warm-up pass (eager,
on the capture stream) a pre-hook on every parameter's AccumulateGrad node records the stream the engine runs it on, and a
mismatch raises ``PswinError`` naming the rule BEFORE anything is captured.  The nodes the guard creates itself are created on
the capture stream and kept alive by the object, so a step built, warmed up and captured on one stream stays on it.
"""
import gc

import torch

from ._lib import PswinError


def _as_parameters(parameters):
    """parameters: an iterable of nn.Module / nn.Parameter / tensors, or None = every HIP nn.Parameter alive in the process that
    requires a gradient (the guard runs once per capture; a scan of the collector's objects is cheap next to a capture)."""
    if parameters is None:
        return [o for o in gc.get_objects() if isinstance(o, torch.nn.Parameter) and o.is_cuda and o.requires_grad]
    out = []
    for item in parameters:
        if isinstance(item, torch.nn.Module):
            out += [p for p in item.parameters() if p.requires_grad]
        elif torch.is_tensor(item):
            if item.requires_grad and item.is_leaf:
                out.append(item)
        else:
            raise TypeError("parameters: modules or leaf tensors expected")
    seen, uniq = set(), []
    for p in out:
        if id(p) not in seen:
            seen.add(id(p))
            uniq.append(p)
    return uniq


class AccumulateStreamGuard:
    """Records, for every given parameter, the stream its AccumulateGrad node runs on while `armed`; check(stream) raises if one of
    them ran on another stream.  current_stream: how to read the running stream (injected by the CPU test)."""

    def __init__(self, parameters, current_stream=None):
        self._current = current_stream or torch.cuda.current_stream
        self.params = list(parameters)
        self.seen = {}                        # index of the parameter -> stream its node ran on
        self.nodes, self._handles = [], []
        with torch.enable_grad():
            for i, p in enumerate(self.params):
                node = p.view_as(p).grad_fn.next_functions[0][0]  # the existing node, or a new one bound to the CURRENT stream
                self.nodes.append(node)                            # kept alive: later passes reuse it
                self._handles.append(node.register_prehook(self._make_hook(i)))

    def _make_hook(self, i):
        def hook(grads):
            self.seen[i] = self._current()   # the engine guards onto the node's stream before it calls the node and its pre-hooks
            return None
        return hook

    def disarm(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def check(self, stream, names=None):
        bad = [i for i, s in self.seen.items() if s != stream]
        if not bad:
            return
        label = (names or {}).get(id(self.params[bad[0]]), f"a parameter of shape {tuple(self.params[bad[0]].shape)}")
        # the exception's traceback keeps this object alive for as long as the caller holds it: let go of the nodes now, so that they can
        # die with the graph that owns them and a retry (after `del loss`) finds fresh ones
        self.nodes = []
        raise PswinError(
            f"hipGraph capture refused: the AccumulateGrad nodes of {len(bad)} of {len(self.seen)} parameters (first: {label}) run on "
            f"{self.seen[bad[0]]}, not on the capture stream {stream}.  They were created by an earlier forward pass on that stream and "
            "are kept alive by its autograd graph (a `loss` / output tensor of an eager step that is still referenced, or a hook holder).  "
            "Capturing a backward pass on another stream pulls that stream into the capture: hipStreamEndCapture then crashes the process "
            "(default stream) or the replays return garbage gradients.  The rule: construct, warm up, capture and replay on ONE stream -- "
            "`s = torch.cuda.Stream(); with torch.cuda.stream(s): model = build(); opt = ...; step = GraphedCallable(fn, stream=s)` -- or "
            "drop every reference to the earlier passes' outputs / losses (`del loss, outs`) before constructing the graph.")


def _guarded_warmup(fns, warmup, stream, parameters):
    """`warmup` eager passes over `fns` on `stream`; the first one under the AccumulateGrad stream guard.  Returns the guard (its
    nodes must outlive the captures)."""
    stream.wait_stream(torch.cuda.current_stream())
    guard = None
    with torch.cuda.stream(stream):
        params = _as_parameters(parameters)
        guard = AccumulateStreamGuard(params)
        try:
            for it in range(max(1, warmup)):
                for fn in fns:
                    out = fn()
                del out
                if it == 0:
                    guard.disarm()
                    guard.check(stream)
        finally:
            guard.disarm()
    torch.cuda.current_stream().wait_stream(stream)
    torch.cuda.synchronize()
    return guard


class GraphedCallable:
    def __init__(self, fn, warmup=3, stream=None, parameters=None):
        """Warm `fn` up and capture it ON THE SAME side stream (the one-stream rule of the module docstring; a violation raises
        PswinError before the capture starts).  parameters: the modules / parameters whose gradients `fn` produces (None: every HIP
        nn.Parameter alive in the process)."""
        self.fn = fn
        self.stream = stream or torch.cuda.Stream()
        self._guard = _guarded_warmup([fn], warmup, self.stream, parameters)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out


class GraphedSequence:
    """Consecutive pieces of ONE step (e.g. forward + late-layer backward | early-layer backward) captured into separate
    hipGraphs that share a memory pool, so that tensors produced by an earlier piece (activations, the autograd graph's
    saved tensors) stay valid for the later ones.  Between two replays the host can launch work that must not be
    captured -- the RCCL all-reduce of the gradients the first piece has already finished.  `calls[i]()` replays piece
    i and returns what its function returned at capture time (static tensors).  Same one-stream guard as GraphedCallable."""

    def __init__(self, fns, warmup=2, stream=None, parameters=None):
        self.stream = stream or torch.cuda.Stream()
        self._guard = _guarded_warmup(list(fns), warmup, self.stream, parameters)
        self.graphs, self.outs = [], []
        pool = None
        for fn in fns:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream, pool=pool):
                self.outs.append(fn())
            pool = pool or g.pool()
            self.graphs.append(g)
        self.calls = [self._make(i) for i in range(len(fns))]

    def _make(self, i):
        def call():
            self.graphs[i].replay()
            return self.outs[i]
        return call
