"""HIP-graph capture of a fixed-shape training step.

An eager PanoSwin-T step is ~1100 kernel launches; on MI355X the launches, not the kernels, then set the step time
(measured: 25 ms of GPU work in a 33-48 ms eager step).  Capturing the step once into a hipGraph and replaying it
removes the per-launch host cost.  Requirements on the captured callable: static shapes, no host synchronisation
(`.item()`, prints of device values), every lazily built table already cached (run it a few times first -- that is
what `warmup` does), optimizers constructed with ``capturable=True``.  The C-ABI kernels need nothing special: they
are launched on torch's current stream, which is the capturing stream during capture.
"""
import torch


class GraphedCallable:
    def __init__(self, fn, warmup=3, stream=None):
        """Warm `fn` up and capture it ON THE SAME side stream: autograd's AccumulateGrad nodes remember the stream
        they were created on, and a capture on a different stream would put them on a fork of the graph that is not
        joined before the consumers of the gradients (observed: garbage gradients from the second replay on)."""
        self.fn = fn
        self.stream = stream or torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                out = fn()
            del out
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
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
    i and returns what its function returned at capture time (static tensors)."""

    def __init__(self, fns, warmup=2, stream=None):
        self.stream = stream or torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                for fn in fns:
                    out = fn()
            del out
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
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
