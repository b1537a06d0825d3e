"""Data parallelism over rows: one process per GPU, RCCL (``backend="nccl"`` on ROCm) over xGMI.

The reference has no distributed code; the only contract is "same math as one big
batch" (``loss = -mean_b``, suave/model.py:1617).  Each rank runs ``suave_elbo_backward`` on its row
shard with ``global_rows`` = the global batch size, so rank gradients are already scaled by
1/global_rows and their SUM is the single-batch gradient -- also for ragged last batches, which the
reference allows (no drop-last, model.py:1475-1476).  Parameters and Adam state are replicated and stay
bit-identical because every rank applies the same reduced gradient.

Generation / encode / impute shard by rows with no collective: the counter-based generators are keyed
by the global row index (``row_offset``), so outputs do not depend on the number of GPUs.

``DataParallel`` is what ``SUAVE.fit`` and ``bench.py`` drive: the gradient all-reduce is cut in two buckets
in the order the backward finishes them.  ``suave_elbo_backward_overlap`` records an event as soon as the
decoder + likelihood heads + prior bucket is final -- before the latent / encoder backward is launched -- and
the first collective starts behind that event on RCCL's stream while the encoder backward still runs on the
compute stream; the encoder bucket follows on the compute stream's tail.
"""

from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block shard [begin, end) of ``n_rows`` rows; sizes differ by at most one row."""

    base, rem = divmod(int(n_rows), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def active(group=None) -> bool:
    """torch.distributed is initialised with more than one rank."""

    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def generation_span(n_rows: int, row_offset: Optional[int], group=None) -> Tuple[int, int]:
    """Which rows a generation / inference call of the product computes here, as (first global row, count).

    ``row_offset`` given: the caller shards by itself -- its ``n_rows`` rows are global rows ``row_offset ..``.
    Otherwise, with more than one rank, the request is the same on every rank and rank r takes
    ``shard_rows(n_rows, r, world)`` of it; on one rank: everything.  Either way every generator is keyed by the
    GLOBAL row index, so the union of the ranks' outputs is the one-GPU output whatever the number of GPUs."""

    if row_offset is not None:
        return int(row_offset), int(n_rows)
    if not active(group):
        return 0, int(n_rows)
    b, e = shard_rows(n_rows, dist.get_rank(group), dist.get_world_size(group))
    return b, e - b


def sharded_epoch_batches(perm: torch.Tensor, batch: int, row_base: int, n_local: int):
    """The global batches of one epoch over a ROW-SHARDED table, as seen by the rank that holds rows [row_base,
    row_base + n_local): yields ``(start, stop, rows)`` -- the batch is ``perm[start:stop]`` (a permutation of the
    global row ids, the same on every rank) and ``rows`` are its entries held here, order kept, as int32.

    The selection is made once per epoch (one compaction of the permutation and one per-batch count vector: two host
    syncs an epoch) so that no step waits for the host."""

    n = int(perm.numel())
    batch = max(int(batch), 1)
    p32 = perm.to(torch.int32)
    sel = (p32 >= int(row_base)) & (p32 < int(row_base) + int(n_local))
    local = p32[sel]
    n_batches = (n + batch - 1) // batch
    pad = torch.zeros(n_batches * batch, dtype=torch.int32, device=perm.device)
    pad[:n] = sel.to(torch.int32)
    ends = pad.view(n_batches, batch).sum(dim=1).cumsum(0).cpu().tolist()
    begin = 0
    for i in range(n_batches):
        yield i * batch, min((i + 1) * batch, n), local[begin:int(ends[i])]
        begin = int(ends[i])


def all_gather_objects(obj, group=None) -> list:
    """One Python object per rank, in rank order (small host data: sizes, label arrays, frames on request)."""

    if not active(group):
        return [obj]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


def all_gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate the ranks' tensors along dim 0 in rank order (row counts may differ)."""

    if not active(group):
        return t
    world = dist.get_world_size(group)
    counts = [int(c) for c in all_gather_objects(int(t.shape[0]), group)]
    staged = dist.get_backend(group) != "nccl" and t.is_cuda  # gloo rehearsal: through host memory
    src = t.cpu() if staged else t
    width = max(counts)
    pad = src.new_zeros((width,) + tuple(src.shape[1:]))
    pad[: src.shape[0]] = src
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.cat([p[:c] for p, c in zip(parts, counts)])
    return out.to(t.device) if staged else out


def all_reduce_host(values, op: str = "sum", group=None):
    """float64 numpy array summed / maxed over ranks (normalisation statistics of a sharded table)."""

    import numpy as np

    arr = np.asarray(values, dtype=np.float64)
    if not active(group):
        return arr
    t = torch.from_numpy(arr.copy())
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX, group=group)
    return t.cpu().numpy()


def global_batch_rows(local_rows: int, group=None) -> int:
    """Sum of the ranks' batch rows (ragged last batches differ between ranks)."""

    if not dist.is_available() or not dist.is_initialized():
        return int(local_rows)
    t = torch.tensor([int(local_rows)], dtype=torch.int64)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, group=group)
    return int(t.item())


def gradient_buckets(layout, max_bucket_floats: int = 8 << 20) -> List[Tuple[int, int]]:
    """Flat-gradient ranges in the order backward finishes them: decoder (heads, y_projection, backbone) +
    prior first, encoder last.  Small models (S64: 0.93 MB) collapse into a single bucket -- one latency hit
    on the per-link-bound xGMI ring; wide models are cut at ``max_bucket_floats``."""

    enc_b, enc_e = layout.group_ranges["encoder"]
    rest_b, rest_e = layout.group_ranges["decoder"][0], layout.group_ranges["prior"][1]
    if layout.n_params <= max_bucket_floats:
        return [(0, layout.n_params)]
    out: List[Tuple[int, int]] = []
    for b, e in ((rest_b, rest_e), (enc_b, enc_e)):
        cur = b
        while cur < e:
            nxt = min(e, cur + max_bucket_floats)
            out.append((cur, nxt))
            cur = nxt
    return out


def allreduce_gradients(flat_grads: torch.Tensor, buckets: Optional[Sequence[Tuple[int, int]]] = None, group=None,
                        async_op: bool = False):
    """SUM-all-reduce the flat gradient buffer (bucket by bucket).  Returns the work handles when
    ``async_op`` so the caller can overlap the collectives with the remaining backward launches."""

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return []
    handles = []
    if flat_grads.is_cuda and dist.get_backend(group) == "gloo":
        # CPU rehearsal of the multi-rank path (tests, one-GPU boxes): gloo reduces host memory
        for b, e in (buckets or [(0, flat_grads.numel())]):
            host = flat_grads[b:e].cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat_grads[b:e].copy_(host)
        return handles
    for b, e in (buckets or [(0, flat_grads.numel())]):
        h = dist.all_reduce(flat_grads[b:e], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            handles.append(h)
    return handles


def allreduce_sums(sums: torch.Tensor, group=None) -> torch.Tensor:
    """Epoch metrics (sum recon, sum cat_kl, sum gauss_kl, n_rows as float64) so that early-stopping /
    best-state decisions are identical on every rank (model.py:1508-1517, 2544-2558)."""

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


class DataParallel:
    """Row-sharded data parallelism of one fit: rank / world, the two gradient buckets and the overlap plumbing.

    Every rank holds the same packed table and the same parameters and walks the same global batches (same
    permutation: same seed, same generator); of each global batch it computes the rows ``shard_rows`` gives it,
    with ``global_rows`` = the global batch size.  The kernels key their noise by (seed, step, table row), so the
    reduced gradient equals the one-GPU gradient of the whole batch up to fp32 summation order.
    """

    def __init__(self, layout, device, group=None) -> None:
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = torch.device(device)
        self.host_staged = dist.get_backend(group) != "nccl"  # gloo rehearsal: reduce through host memory
        self.early = (layout.group_ranges["decoder"][0], layout.group_ranges["prior"][1])  # final before the encoder backward
        self.late = tuple(layout.group_ranges["encoder"])
        if not (self.late[1] == self.early[0] or self.early[1] == self.late[0]):
            raise ValueError("gradient groups are expected to be two adjacent ranges")
        self.ready: Optional[torch.cuda.Event] = None
        self.side: Optional[torch.cuda.Stream] = None
        if not self.host_staged:
            self.side = torch.cuda.Stream(self.device)
            self.ready = torch.cuda.Event()
            self.ready.record(torch.cuda.current_stream(self.device))  # materialise the hipEvent_t the library records

    @classmethod
    def from_environment(cls, layout, device, group=None) -> Optional["DataParallel"]:
        """None unless torch.distributed is initialised with more than one rank (SUAVE_DP_SINGLE_RANK=1 keeps the
        plumbing on for a one-rank group: the rehearsal of the RCCL stream / event path on a one-GPU box)."""

        if not dist.is_available() or not dist.is_initialized():
            return None
        if dist.get_world_size(group) == 1 and os.environ.get("SUAVE_DP_SINGLE_RANK", "0") != "1":
            return None
        return cls(layout, device, group)

    # -- sharding ---------------------------------------------------------------
    def span(self, n_rows: int) -> Tuple[int, int]:
        return shard_rows(n_rows, self.rank, self.world)

    def shard(self, rows: torch.Tensor) -> torch.Tensor:
        """This rank's rows of a global batch (a tensor of table row ids)."""

        b, e = self.span(int(rows.shape[0]))
        return rows[b:e]

    # -- collectives ------------------------------------------------------------
    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        """In-place SUM over ranks; every rank ends with the same bits."""

        if self.host_staged and t.is_cuda:
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def reduce_gradients(self, flat_grads: torch.Tensor) -> None:
        """SUM the flat gradient over ranks.  With RCCL, and ``self.ready`` handed to ``Engine.elbo`` for this
        step, the decoder-side bucket is reduced while the encoder backward is still running; the compute stream
        is made to wait for both collectives before it returns to the caller (who launches Adam next)."""

        if self.host_staged:
            for b, e in (self.early, self.late):
                self.all_reduce(flat_grads[b:e])
            return
        self.side.wait_event(self.ready)
        with torch.cuda.stream(self.side):
            first = dist.all_reduce(flat_grads[self.early[0]:self.early[1]], op=dist.ReduceOp.SUM, group=self.group,
                                    async_op=True)
        second = dist.all_reduce(flat_grads[self.late[0]:self.late[1]], op=dist.ReduceOp.SUM, group=self.group,
                                 async_op=True)
        first.wait()   # both: the current (compute) stream waits, the host does not
        second.wait()

    def reduce_module_gradients(self, module: torch.nn.Module) -> None:
        """SUM the gradients of a small torch module (the classifier head) as one flat buffer."""

        ps = [p for p in module.parameters() if p.requires_grad]
        if not ps:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps])
        self.all_reduce(flat)
        off = 0
        for p in ps:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(flat[off:off + n].view_as(p))
            off += n

    def broadcast_module(self, module: torch.nn.Module, src: int = 0) -> None:
        for t in list(module.parameters()) + list(module.buffers()):
            if self.host_staged and t.is_cuda:
                host = t.detach().cpu()
                dist.broadcast(host, src=src, group=self.group)
                t.data.copy_(host)
            else:
                dist.broadcast(t.data, src=src, group=self.group)

    def check_replicated(self, *tensors: torch.Tensor) -> None:
        """Every rank must have been given the same table.  Compared over ranks: the row count and, per tensor, a
        64-bit order-sensitive hash of its bits (sum of word_i * (2654435761 i + 1) modulo 2^64, one device reduction
        each) -- tables with the same rows in another order, or with equal column sums, do not pass."""

        sig = [sum(int(t.shape[0]) for t in tensors)]
        for t in tensors:
            if t.numel() == 0:
                sig.append(0)
                continue
            flat = t.contiguous().view(-1)
            words = flat.view(torch.int32) if flat.element_size() == 4 else flat.to(torch.int32)
            pos = torch.arange(words.numel(), device=words.device, dtype=torch.int64)
            sig.append(int(((pos * 2654435761 + 1) * words.to(torch.int64)).sum().item()))  # int64 wraps: modulo 2^64
        sig_t = torch.tensor(sig, dtype=torch.int64)
        lo, hi = sig_t.clone(), sig_t.clone()
        if not self.host_staged:
            lo, hi = lo.to(self.device), hi.to(self.device)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not torch.equal(lo.cpu(), hi.cpu()):
            raise ValueError("data-parallel fit needs the SAME X / y on every rank (rows are sharded inside fit; pass "
                             "sharded=True when every rank holds its own block of rows); the ranks' tables differ")
