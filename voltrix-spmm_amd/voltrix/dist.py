"""Row-window sharding across the GPUs of one node + all-gather of the dense operand (RCCL over xGMI).

No reference counterpart -- the reference is single-GPU (SURVEY.md section 2.1 "Parallelism", section 8e).
Output row windows are independent, so the sparse matrix shards by contiguous row-window ranges (balanced by
edge count, boundaries on multiples of 16 rows); every rank keeps global column ids, owns the matching row
slice of the dense operand B and produces the matching row slice of C.  The one exchange step per SpMM is an
all-gather of B.

Layout trick: ``all_gather_into_tensor`` needs equal contributions, and shards differ by up to one window plus
balance slack.  Each rank therefore contributes ``rows_padded = max shard rows`` rows and the *column ids of the
local CSR are remapped once at setup* to ``owner(col) * rows_padded + (col - row_start[owner(col)])``, so the
gathered buffer ``[world * rows_padded, F]`` is used as B directly -- no unpadding copy on the hot path.

Round 3: the operator is built from the rank's OWN shard, wherever it lives (``RowShardedSpMM.from_shard``: device CSR
in, ``csr_preprocess_device``, no host round trip, no full graph on any rank -- what ``bench.py --gpus N`` runs), and the
exchange step has three forms (``mode``):

  ``collective``  one ``all_gather_into_tensor`` (in place on RCCL): whatever algorithm RCCL picks over xGMI
  ``p2p``         the direct schedule written out: one send and one receive per peer, batched into one group
                  (``batch_isend_irecv``), i.e. world - 1 concurrent point-to-point copies per rank -- on a fully
                  connected xGMI node every link carries exactly one shard (~23 ms for 8 x 3.55 GB
                  against ~160 ms if the collective falls back to a ring)
  ``rows``        only the rows of B this rank's shard REFERENCES travel ("gather only referenced rows"):
                  at setup every rank tells every owner which of its rows it needs (one all-to-all of index lists); per step
                  each owner packs the requested rows (one indexed copy) and ONE ``all_to_all_single`` with uneven splits
                  delivers them into a compact buffer [own rows | rows from owner 0 | owner 1 | ...] whose positions the local
                  column ids were remapped to.  papers-like on 8 GPUs: 59 % of every remote shard instead of all of it.
  ``slabs=k``     (with ``collective`` / ``p2p``) B is exchanged and multiplied in k feature slabs: the all-gather of slab j + 1
                  runs on the communication stream while the SpMM multiplies slab j

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

BLK_H = 16


def partition_rows(indptr: torch.Tensor, num_nodes: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous row ranges ``[(row_start, row_end)] * world_size``: starts are multiples of 16, edge counts as
    equal as the window granularity allows (the unit of SpMM work is the gathered row, i.e. ~ an edge).
    Deterministic: every rank computes the same partition from the same ``indptr`` (any device: the search runs where
    ``indptr`` lives, only the world_size - 1 boundaries come to the host)."""
    assert indptr.numel() == num_nodes + 1
    num_windows = (num_nodes + BLK_H - 1) // BLK_H
    dev = indptr.device
    win_start = torch.arange(0, num_windows + 1, dtype=torch.int64, device=dev) * BLK_H
    win_start[-1] = num_nodes
    edges_before = indptr.to(torch.int64)[win_start]  # edges before each window boundary
    total = int(edges_before[-1])
    targets = torch.tensor([total * r // world_size for r in range(1, world_size)], dtype=torch.int64, device=dev)
    found = torch.searchsorted(edges_before, targets, right=False).tolist() if world_size > 1 else []
    bounds = [0]
    for w in found:
        bounds.append(max(bounds[-1], min(int(w), num_windows)))
    bounds.append(num_windows)
    return [(min(bounds[r] * BLK_H, num_nodes), min(bounds[r + 1] * BLK_H, num_nodes)) for r in range(world_size)]


def shard_csr(indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, parts: List[Tuple[int, int]], rank: int):
    """Local CSR of ``rank`` (rows of its range, global column ids), int32, on the input's device."""
    r0, r1 = parts[rank]
    e0, e1 = int(indptr[r0]), int(indptr[r1])
    local_indptr = (indptr[r0:r1 + 1] - indptr[r0]).to(torch.int32).contiguous()
    local_indices = indices[e0:e1].to(torch.int32).contiguous()
    return local_indptr, local_indices


def remap_columns(indices: torch.Tensor, parts: List[Tuple[int, int]], rows_padded: int) -> torch.Tensor:
    """Global column id -> row of the padded all-gather buffer (see module docstring)."""
    starts = torch.tensor([p[0] for p in parts], dtype=torch.int64, device=indices.device)
    cols = indices.to(torch.int64)
    owner = torch.searchsorted(starts, cols, right=True) - 1
    return (owner * rows_padded + (cols - starts[owner])).to(torch.int32)


class RowShardedSpMM:
    """``C_local = A[rows of this rank, :] @ all_gather(B_local)`` for a binary CSR ``A``.

    ``local_preprocess(indptr_i32, indices_i32, n_rows) -> handle`` and ``local_spmm(handle, n_rows, n_edges, feat) -> out``
    default to the HIP path (``voltrix.csr_preprocess_device`` for a shard that lives on the GPU, ``voltrix.csr_preprocess``
    for a host one; ``voltrix.spmm``); the CPU tests inject their own.  ``mode`` / ``slabs``: module docstring.
    """

    def __init__(self, indptr: torch.Tensor, indices: torch.Tensor, num_nodes: int, group=None,
                 device: Optional[torch.device] = None, local_preprocess: Optional[Callable] = None,
                 local_spmm: Optional[Callable] = None, hash_tag: Optional[str] = None, mode: str = "collective",
                 slabs: int = 1, exchange_at_world_1: bool = False):
        """From the FULL CSR, present on every rank (any device): partitions, slices and remaps where the tensors live.
        Convenient for graphs that fit one GPU; at papers100M scale use :meth:`from_shard`."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        parts = partition_rows(indptr, num_nodes, world)
        local_indptr, local_indices = shard_csr(indptr, indices, num_nodes, parts, rank)
        self._setup(local_indptr, local_indices, num_nodes, parts, group, device, local_preprocess, local_spmm, hash_tag,
                    mode, slabs, exchange_at_world_1)

    @classmethod
    def from_shard(cls, local_indptr: torch.Tensor, local_indices: torch.Tensor, num_nodes: int,
                   parts: List[Tuple[int, int]], group=None, device: Optional[torch.device] = None,
                   local_preprocess: Optional[Callable] = None, local_spmm: Optional[Callable] = None,
                   hash_tag: Optional[str] = None, mode: str = "collective", slabs: int = 1,
                   exchange_at_world_1: bool = False) -> "RowShardedSpMM":
        """From this rank's OWN rows only: ``local_indptr`` int32 [rows + 1], ``local_indices`` int32 with GLOBAL column ids,
        ``parts`` the row ranges of all ranks (e.g. ``partition_rows`` on the degree prefix sums, which every rank can compute
        without the graph).  Nothing of the other shards is ever materialised here; a device CSR stays on the device.
        ``exchange_at_world_1``: issue the collective also in a one-rank group (rehearsal of the RCCL calls on one GPU)."""
        self = cls.__new__(cls)
        self._setup(local_indptr, local_indices, num_nodes, parts, group, device, local_preprocess, local_spmm, hash_tag,
                    mode, slabs, exchange_at_world_1)
        return self

    def _setup(self, local_indptr, local_indices, num_nodes, parts, group, device, local_preprocess, local_spmm, hash_tag,
               mode, slabs, exchange_at_world_1=False):
        assert mode in ("collective", "p2p", "rows") and slabs >= 1
        assert not (mode == "rows" and slabs > 1), "the referenced-rows exchange is not slab-pipelined"
        self._exchange_always = bool(exchange_at_world_1) and dist.is_initialized()
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert len(parts) == self.world_size
        self.num_nodes = num_nodes
        self.parts = list(parts)
        self.row_start, self.row_end = self.parts[self.rank]
        self.local_rows = self.row_end - self.row_start
        assert local_indptr.numel() == self.local_rows + 1
        self.rows_padded = max(1, max(p[1] - p[0] for p in self.parts))
        self.device = device
        self.mode, self.slabs = mode, slabs

        local_indptr = local_indptr.to(torch.int32).contiguous()
        local_indices = local_indices.to(torch.int32).contiguous()
        gathered_rows = self.world_size * self.rows_padded if self.world_size > 1 else num_nodes
        if self.world_size > 1 and mode == "rows":
            local_indices, gathered_rows = self._setup_referenced_rows(local_indices)
        elif self.world_size > 1:
            local_indices = remap_columns(local_indices, self.parts, self.rows_padded)
        self.compact_rows = gathered_rows
        self.local_edges = int(local_indices.numel())
        if local_preprocess is None:
            from .spmm import csr_preprocess, csr_preprocess_device  # HIP path; raises if the extension is missing

            def local_preprocess(ip, ix, n_rows):   # remapped ids index the all-gather buffer
                if ip.is_cuda:
                    return csr_preprocess_device(ip, ix, n_rows, num_cols=gathered_rows)
                return csr_preprocess(ip, ix, n_rows, num_cols=gathered_rows)
        if local_spmm is None:
            from .spmm import spmm as _spmm

            def local_spmm(handle, n_rows, n_edges, feat):
                return _spmm(handle[0], handle[1], handle[2], n_rows, n_edges, feat)
        self._local_spmm = local_spmm
        self.handle = local_preprocess(local_indptr, local_indices, self.local_rows)
        if hash_tag is not None and hasattr(self.handle[1], "data_ptr"):
            try:
                self.handle[1].hash_tag = f"{hash_tag}_r{self.rank}of{self.world_size}"
            except AttributeError:
                pass
        self._buffers = {}      # (key, shape, dtype) -> gather buffer
        self._comm_stream = None

    # ---- referenced rows only ---------------------------------------------------------------------------------------
    def _setup_referenced_rows(self, local_indices: torch.Tensor):
        """Which rows of every owner does this shard reference?  Exchanges the request lists once and remaps the local column
        ids to the compact buffer [own rows | requested rows of owner 0 | owner 1 | ...] (self excluded).  Returns the
        remapped ids and the buffer's row count."""
        dev = local_indices.device
        starts = torch.tensor([p[0] for p in self.parts] + [self.num_nodes], dtype=torch.int64, device=dev)
        need = torch.unique(local_indices.to(torch.int64))                        # sorted global ids
        owner = torch.searchsorted(starts[:-1].contiguous(), need, right=True) - 1
        remote = owner != self.rank
        need_remote, owner_remote = need[remote], owner[remote]
        want_counts = torch.bincount(owner_remote, minlength=self.world_size)     # rows wanted from every owner (0 for self)
        # tell every owner how many, then which (ids local to the owner); the owners answer nothing: from now on they know
        give_counts = torch.empty_like(want_counts)
        dist.all_to_all_single(give_counts, want_counts, group=self.group)
        self._want = [int(v) for v in want_counts.tolist()]
        self._give = [int(v) for v in give_counts.tolist()]
        requests = (need_remote - starts[owner_remote]).contiguous()              # grouped by owner already (need is sorted)
        give_rows = torch.empty(sum(self._give), dtype=torch.int64, device=dev)
        dist.all_to_all_single(give_rows, requests, output_split_sizes=self._give, input_split_sizes=self._want,
                               group=self.group)
        self._give_rows = give_rows                                               # my local rows, grouped by requester
        self._need_remote = need_remote                                           # sorted: (owner, id) order = arrival order
        return self.compact_ids(local_indices), self.local_rows + int(need_remote.numel())

    def compact_ids(self, global_ids: torch.Tensor) -> torch.Tensor:
        """``mode="rows"``: global column ids (of this shard) -> rows of the compact buffer: own rows first, then the requested
        remote rows in (owner, id) order -- the order the all-to-all delivers them in."""
        cols = global_ids.to(torch.int64)
        own = (cols >= self.row_start) & (cols < self.row_end)
        pos_in_need = torch.searchsorted(self._need_remote, cols)
        return torch.where(own, cols - self.row_start, self.local_rows + pos_in_need).to(torch.int32)

    def exchange_rows_into(self, buf: torch.Tensor, feat_local: torch.Tensor) -> torch.Tensor:
        """``mode="rows"``: own rows to the front of ``buf`` [compact_rows, F], the requested remote rows behind them (one
        indexed copy on the sending side + one all-to-all with uneven splits), on the CURRENT stream."""
        assert buf.shape == (self.compact_rows, feat_local.shape[1]) and feat_local.shape[0] == self.local_rows
        buf[: self.local_rows].copy_(feat_local)
        send = feat_local.index_select(0, self._give_rows) if self._give_rows.numel() else feat_local[:0]
        dist.all_to_all_single(buf[self.local_rows:], send.contiguous(), output_split_sizes=self._want,
                               input_split_sizes=self._give, group=self.group)
        return buf

    def exchange_bytes_received(self, num_feats: int, elem_bytes: int) -> int:
        """Bytes this rank receives per step in the configured mode (reporting)."""
        if self.world_size == 1:
            return 0
        if self.mode == "rows":
            return sum(self._want) * num_feats * elem_bytes
        return (self.world_size - 1) * self.rows_padded * num_feats * elem_bytes

    # ---- the exchange step ------------------------------------------------------------------------------------------
    def _buffer(self, key, num_feats, like: torch.Tensor) -> torch.Tensor:
        """Gather buffer, kept and reused.  ``torch.empty``: the padding rows of a shard are never referenced by the remapped
        column ids, every other row is written by the exchange -- and an allocation queues no fill kernel that could race
        with a gather running on another stream (the slab pipeline allocates before it forks, see ``__call__``)."""
        shape = (self.compact_rows if self.mode == "rows" else self.world_size * self.rows_padded, num_feats)
        buf = self._buffers.get(key)
        if buf is None or buf.shape != shape or buf.dtype != like.dtype or buf.device != like.device:
            buf = torch.empty(shape, dtype=like.dtype, device=like.device)
            self._buffers[key] = buf
        return buf

    def own_rows(self, num_feats: int, like: torch.Tensor, key="whole") -> torch.Tensor:
        """This rank's ``[local_rows, F]`` slice of the gather buffer ``key``: a producer that writes B there (instead of into
        a tensor of its own) saves the shard-sized copy at the top of every exchange -- ``gather_into`` recognises the slice
        by its address and sends it as it is."""
        buf = self._buffer(key, num_feats, like)
        if self.mode == "rows" and self.world_size > 1:
            return buf[: self.local_rows]
        return buf[self.rank * self.rows_padded: self.rank * self.rows_padded + self.local_rows]

    def gather_into(self, buf: torch.Tensor, feat_local: torch.Tensor) -> torch.Tensor:
        """All-gather of B on the CURRENT stream: this rank's ``[local_rows, F]`` into its slice of ``buf``
        ``[world * rows_padded, F]``, every other rank's into theirs (padding rows are never referenced by the remapped
        column ids)."""
        assert feat_local.dim() == 2 and feat_local.shape[0] == self.local_rows and buf.shape[1] == feat_local.shape[1]
        if self.mode == "rows" and self.world_size > 1:
            return self.exchange_rows_into(buf, feat_local)
        mine = buf[self.rank * self.rows_padded:(self.rank + 1) * self.rows_padded]
        if mine.data_ptr() != feat_local.data_ptr():
            mine[: self.local_rows].copy_(feat_local)
        if self.world_size == 1 and not self._exchange_always:
            return buf
        if self.mode == "p2p":
            # the direct schedule: world - 1 sends + world - 1 receives in ONE group; peers in rotated order so that at any
            # moment every rank talks to a different one
            ops = []
            for step in range(1, self.world_size):
                to, frm = (self.rank + step) % self.world_size, (self.rank - step) % self.world_size
                ops.append(dist.P2POp(dist.isend, mine, to if self.group is None else dist.get_global_rank(self.group, to),
                                      self.group))
                ops.append(dist.P2POp(dist.irecv, buf[frm * self.rows_padded:(frm + 1) * self.rows_padded],
                                      frm if self.group is None else dist.get_global_rank(self.group, frm), self.group))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            return buf
        send = mine if dist.get_backend(self.group) == "nccl" else mine.clone()  # only NCCL/RCCL defines the in-place form
        dist.all_gather_into_tensor(buf, send, group=self.group)
        return buf

    def gather(self, feat_local: torch.Tensor) -> torch.Tensor:
        """``[local_rows, F]`` per rank -> ``[world * rows_padded, F]`` (a buffer this object keeps and reuses)."""
        if self.world_size == 1 and not self._exchange_always:
            return feat_local.contiguous()
        return self.gather_into(self._buffer("whole", feat_local.shape[1], feat_local), feat_local.contiguous())

    def choose_exchange(self, feat_local: torch.Tensor, modes=("collective", "p2p"), iters: int = 2,
                        clock: Optional[Callable[[], float]] = None) -> dict:
        """Pick the exchange schedule by MEASUREMENT instead of by flag (first scaling run: nobody knows yet whether RCCL's
        all-gather goes direct or rings over xGMI).  Times ``iters`` exchanges per candidate after one warm-up, takes the MAX
        over ranks (one all-reduce per candidate, so every rank sees the same figures and makes the same choice), keeps the
        fastest as ``self.mode`` and returns ``{mode: ms}``.  Only schedules that share this operator's column layout can be
        compared here ("collective" and "p2p": the padded all-gather buffer); the referenced-rows exchange is a different
        operator (its column ids are remapped to a compact buffer) -- build it beside this one and compare whole steps, as
        ``bench.py --gather auto`` does.  ``clock``: seconds, after a device sync (tests inject a fake one)."""
        assert self.mode != "rows", "the referenced-rows operator has one schedule"
        timings = {}
        if self.world_size == 1 and not self._exchange_always:
            return timings
        import time

        def now():
            if clock is not None:
                return clock()
            if feat_local.is_cuda:
                torch.cuda.synchronize(feat_local.device)
            return time.perf_counter()

        buf = self._buffer("whole", feat_local.shape[1], feat_local)
        feat_local = feat_local.contiguous()
        keep = self.mode
        for mode in modes:
            assert mode in ("collective", "p2p")
            self.mode = mode
            self.gather_into(buf, feat_local)             # warm-up: connection set-up, RCCL channel allocation
            if dist.is_initialized():
                dist.barrier(group=self.group)
            t0 = now()
            for _ in range(iters):
                self.gather_into(buf, feat_local)
            t = torch.tensor([(now() - t0) / iters * 1e3], dtype=torch.float64,
                             device=feat_local.device if (dist.is_initialized() and dist.get_backend(self.group) == "nccl")
                             else "cpu")
            if dist.is_initialized():
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            timings[mode] = float(t.item())
        self.mode = min(timings, key=lambda m: (timings[m], modes.index(m))) if timings else keep
        return timings

    def multiply(self, gathered: torch.Tensor) -> torch.Tensor:
        """The local product on an already gathered B."""
        return self._local_spmm(self.handle, self.local_rows, self.local_edges, gathered)

    def _comm(self, device) -> "torch.cuda.Stream":
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=device)
        return self._comm_stream

    def __call__(self, feat_local: torch.Tensor) -> torch.Tensor:
        if self.slabs == 1 or (self.world_size == 1 and not self._exchange_always) or feat_local.shape[1] < 2 * 8:
            return self.multiply(self.gather(feat_local))
        # feature-slab pipeline: slab j + 1 travels while slab j is multiplied
        num_feats = feat_local.shape[1]
        width = -(-num_feats // self.slabs)
        width = -(-width // 8) * 8                       # 16-byte rows for every 16-bit slab
        bounds = [(c, min(num_feats, c + width)) for c in range(0, num_feats, width)]
        pieces = [feat_local[:, a:b].contiguous() for a, b in bounds]
        outs = []
        if not feat_local.is_cuda:                        # CPU groups (tests): same data flow, no streams
            for j, piece in enumerate(pieces):
                outs.append(self.multiply(self.gather_into(self._buffer(("slab", j % 2), piece.shape[1], piece), piece)))
            return torch.cat(outs, dim=1)
        main, comm = torch.cuda.current_stream(), self._comm(feat_local.device)
        gathered_ev, consumed_ev = [None, None], [None, None]
        # both slab buffers exist BEFORE the fork: whatever their allocation queues on `main` is ordered before `ready`, hence
        # before the first gather on `comm` (an allocation inside launch_gather would be unordered against that gather)
        slab_bufs = [self._buffer(("slab", b), pieces[min(b, len(pieces) - 1)].shape[1], pieces[0]) for b in range(2)]
        ready = torch.cuda.Event()
        ready.record(main)                                # the slabs were cut on `main`

        def launch_gather(j):
            b = j % 2
            buf = slab_bufs[b]
            if buf.shape[1] != pieces[j].shape[1]:        # a narrower last slab: a view of the same storage, no allocation
                buf = buf.view(-1)[: buf.shape[0] * pieces[j].shape[1]].view(buf.shape[0], pieces[j].shape[1])
            with torch.cuda.stream(comm):
                comm.wait_event(ready)
                if consumed_ev[b] is not None:
                    comm.wait_event(consumed_ev[b])       # the product that read this buffer two slabs ago
                self.gather_into(buf, pieces[j])
                gathered_ev[b] = torch.cuda.Event()
                gathered_ev[b].record(comm)
            return buf

        bufs = {0: launch_gather(0)}
        for j in range(len(pieces)):
            if j + 1 < len(pieces):
                bufs[j + 1] = launch_gather(j + 1)
            main.wait_event(gathered_ev[j % 2])
            outs.append(self.multiply(bufs.pop(j)))
            consumed_ev[j % 2] = torch.cuda.Event()
            consumed_ev[j % 2].record(main)
        return torch.cat(outs, dim=1)


def predicted_step_ms(world_size: int, shard_bytes: float, local_spmm_ms: float, link_gbs: float = 153.0,
                      ring_efficiency: float = 0.8) -> dict:
    """What one step (all-gather of B + local SpMM) should take on a fully connected xGMI node (MI355X guide: 7 links x
    ~153 GB/s per GPU), for reading the first measured scaling curve against (profiles/HISTORY.md section 6).  ``shard_bytes`` = bytes
    one rank contributes.  direct: every rank receives world - 1 shards over world - 1 links in parallel; ring: world - 1
    sequential hops of one shard over ONE link each.  ``overlapped`` = the exchange of step k + 1 hidden behind the product
    of step k (the step is whichever is longer)."""
    w = world_size
    if w <= 1:
        return {"allgather_direct_ms": 0.0, "allgather_ring_ms": 0.0, "step_direct_ms": local_spmm_ms,
                "step_ring_ms": local_spmm_ms, "step_direct_overlapped_ms": local_spmm_ms,
                "step_ring_overlapped_ms": local_spmm_ms}
    direct = shard_bytes / (link_gbs * 1e9) * 1e3
    ring = (w - 1) * shard_bytes / (link_gbs * ring_efficiency * 1e9) * 1e3
    return {"allgather_direct_ms": direct, "allgather_ring_ms": ring,
            "step_direct_ms": direct + local_spmm_ms, "step_ring_ms": ring + local_spmm_ms,
            "step_direct_overlapped_ms": max(direct, local_spmm_ms), "step_ring_overlapped_ms": max(ring, local_spmm_ms)}
