"""hipGraph capture of the eval-mode hot path (BASELINE config 5: "hipGraph-captured 3D hourglass").

The forward is ~175 kernel launches through ctypes; at 544x960 the GPU is the bottleneck, but at small shapes
(256x512, D=64) the step is host-bound.  All launches go to torch's current stream, every buffer comes from torch's
caching allocator and the closed-form context injection has no host sync, so the whole path captures into one
hipGraph (torch.cuda.CUDAGraph) and replays with static input/output buffers."""
from __future__ import annotations

import torch


class GraphedHotPath:
    """Captures `model.hot_path(fL, fR[, cL, cR])` (eval mode, no grad) for fixed shapes; call with new tensors of
    the same shapes to replay.  Returns the dict of outputs (static buffers: clone them if they must outlive the
    next call)."""

    def __init__(self, model, *example_inputs: torch.Tensor, warmup: int = 2):
        assert not model.training, "graph capture is for the eval path (training BN updates buffers)"
        self.model = model
        self.static_in = [t.detach().clone() for t in example_inputs]
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(stream):
            for _ in range(warmup):                      # warms hipFuncSetAttribute / allocator state outside capture
                model.hot_path(*self.static_in)
        torch.cuda.current_stream().wait_stream(stream)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = model.hot_path(*self.static_in)

    def __call__(self, *inputs: torch.Tensor):
        for s, t in zip(self.static_in, inputs):
            assert s.shape == t.shape, "graph was captured for a different shape"
            s.copy_(t)
        self.graph.replay()
        return self.static_out
