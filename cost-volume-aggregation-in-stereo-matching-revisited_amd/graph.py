"""hipGraph capture of the eval-mode hot path (BASELINE config 5: "hipGraph-captured 3D hourglass") and of the whole
training step.

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

    def __init__(self, model, *example_inputs, warmup: int = 2):
        """example_inputs: the arguments of `model.hot_path` -- tensors, or tuples of tensors (the extractor's l2 / l3 /
        l4 segments), or None"""
        assert not model.training, "graph capture is for the eval path (training BN updates buffers)"
        self.model = model
        self.static_in = [_tree(t, lambda u: u.detach().clone()) for t in example_inputs]
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(stream):
            for _ in range(warmup):                      # warms hipFuncSetAttribute / allocator state outside capture
                model.hot_path(*self.static_in)
        torch.cuda.current_stream().wait_stream(stream)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = model.hot_path(*self.static_in)

    def __call__(self, *inputs):
        for s, t in zip(self.static_in, inputs):
            _tree2(s, t, _copy_checked)
        self.graph.replay()
        return self.static_out


def _tree(t, fn):
    if t is None:
        return None
    if isinstance(t, (tuple, list)):
        return tuple(_tree(u, fn) for u in t)
    return fn(t)


def _tree2(s, t, fn):
    if s is None:
        return
    if isinstance(s, tuple):
        for a, b in zip(s, t):
            _tree2(a, b, fn)
    else:
        fn(s, t)


def _copy_checked(s, t):
    assert s.shape == t.shape, "graph was captured for a different shape"
    s.copy_(t)


class GraphedTrainStep:
    """One training step as hipGraph replays: `local_step()` (zero grads, forward, losses, backward, gradient gather --
    ~1000 launches through ctypes and the autograd engine) is captured into one graph and `optimizer_step()` into the
    same graph (single process) or a second one (data parallel: the all-reduce stays an eager RCCL call between the two
    replays).  Requirements, all met by this package: no host synchronisation inside the step (models/loss.py avoids
    boolean indexing), static input tensors, an optimizer built with `capturable=True`, every launch on torch's current
    stream, every buffer from torch's allocator.

    Construct it under the stream the training loop runs on; `__call__()` replays one step and returns the (static)
    value `local_step` returned.

    Capture needs `warmup` REAL steps first (optimizer state, allocator pools, kernel attributes must exist before the
    capture): they update parameters, optimizer moments and BatchNorm running statistics on the static batch.  Pass
    `restore=` (an iterable of tensors: parameters, buffers, optimizer state) to have their values snapshotted before and
    put back after the warm-up, so that the first replay is step 1 of the run; optimizer state that does not exist yet
    (Adam's moments are created by the first step) is zeroed instead via `restore_optimizer=`."""

    def __init__(self, local_step, optimizer_step, all_reduce=None, warmup: int = 3, restore=None,
                 restore_optimizer=None):
        self.all_reduce = all_reduce
        saved = [(t, t.detach().clone()) for t in (restore or [])]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # optimizer state, allocator pools, kernel attributes
            for _ in range(warmup):
                local_step()
                if all_reduce is not None:
                    all_reduce()
                optimizer_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for t, v in saved:
                t.copy_(v)
            if restore_optimizer is not None:
                for st in restore_optimizer.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
        self.graph_a = torch.cuda.CUDAGraph()
        self.graph_b = None
        with torch.cuda.graph(self.graph_a):
            self.static_out = local_step()
            if all_reduce is None:
                optimizer_step()
        if all_reduce is not None:
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool()):
                optimizer_step()

    def __call__(self):
        self.graph_a.replay()
        if self.graph_b is not None:
            # The collective is issued only once graph A has drained.  Issued behind an in-flight replay it has its
            # own stream wait for an event at the tail of a ~1000-node graph: in one process that wait costs ~0.9 ms
            # per step (10.4 vs 9.6 ms at 64x128x32), and in the rehearsal with two gloo ranks time-slicing ONE GPU it
            # stalled ~150 ms per step (188 vs 32 ms) because the other rank's replay holds the queues the copy needs
            # (tools/dp_graph_probe.py).  The host sync costs two launch latencies per step.
            torch.cuda.current_stream().synchronize()
            self.all_reduce()
            self.graph_b.replay()
        return self.static_out
