"""Data parallelism over rows: one process per GPU, RCCL (``backend="nccl"`` on ROCm) over xGMI.

The reference has no distributed code; the only contract is "same math as one big
batch" (``loss = -mean_b``, suave/model.py:1617).  Each rank runs ``suave_elbo_backward`` on its row
shard with ``global_rows`` = the global batch size, so rank gradients are already scaled by
1/global_rows and their SUM is the single-batch gradient -- also for ragged last batches, which the
reference allows (no drop-last, model.py:1475-1476).  Parameters and Adam state are replicated and stay
bit-identical because every rank applies the same reduced gradient.

Generation / encode / impute shard by rows with no collective: the counter-based generators are keyed
by the global row index (``row_offset``), so outputs do not depend on the number of GPUs.
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block shard [begin, end) of ``n_rows`` rows; sizes differ by at most one row."""

    base, rem = divmod(int(n_rows), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


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
