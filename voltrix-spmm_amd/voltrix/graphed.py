"""Launch-bound use of the operator (small graphs, many calls): ``voltrix.spmm`` captured once in a HIP graph.

After its first call for a (handle, width, dtype) -- JIT tile / schedule sweep, module loads, unit table -- ``voltrix.spmm``
issues only stream-ordered work (cast kernels, zero fill, the SpMM launches of either format, the combine pass), so the whole
call replays from a graph: one ``hipGraphLaunch`` instead of 3-8 launches plus their Python (cora-like, F=32: 25-40 us of
host time per eager call, harness/experiments/call_overhead.py).  No reference counterpart (the reference's operator is one
launch per call, voltrix/spmm/spmm.py:92-114).
"""
import torch

from .spmm import spmm


class GraphedSpMM:
    """``op = GraphedSpMM(blk_offsets, hspa_packed, hind, num_nodes, num_edges, feat_like)``; ``out = op(feat)``.

    ``feat`` must have ``feat_like``'s shape, dtype and device; it is copied into the graph's static input (stream-ordered)
    and the returned tensor is the graph's static output -- valid until the next call (clone it to keep it)."""

    def __init__(self, blk_offsets, hspa_packed, hind, num_nodes: int, num_edges: int, feat_like: torch.Tensor):
        assert feat_like.is_cuda and feat_like.dim() == 2
        self._args = (blk_offsets, hspa_packed, hind, num_nodes, num_edges)
        self.static_input = torch.zeros_like(feat_like, memory_format=torch.contiguous_format)
        spmm(*self._args, self.static_input)   # warm: tuner, format choice, first launches -- nothing of this is captured
        spmm(*self._args, self.static_input)
        torch.cuda.synchronize(feat_like.device)
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=feat_like.device)
        side.wait_stream(torch.cuda.current_stream(feat_like.device))
        with torch.cuda.stream(side):
            with torch.cuda.graph(self.graph, stream=side):
                self.static_output = spmm(*self._args, self.static_input)
        torch.cuda.current_stream(feat_like.device).wait_stream(side)

    def __call__(self, feat: torch.Tensor) -> torch.Tensor:
        assert feat.shape == self.static_input.shape and feat.dtype == self.static_input.dtype
        self.static_input.copy_(feat)
        self.graph.replay()
        return self.static_output
