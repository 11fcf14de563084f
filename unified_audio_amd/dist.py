"""Multi-GPU driver: utterances (and UniSE 5 s segments) are independent, so the hot path shards by clips with NO
collective on the data path; the only exchange steps are the trivial scatter of clips from rank 0 and the gather of
codes / waveforms back.  The reference does rank-strided sharding with no communication at all
(QuarkAudio-UniSE/dataloader/data_module.py:364); this module adds the optional single-collector convenience.

One process per GPU (`torchrun`), `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" in the
CPU tests.  Transfers are direct rank0 <-> rank_i (scatter / gather), never an all-reduce: nothing is reduced.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced block partition of n items: the first n % world ranks get one extra item."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_counts(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def _meta_from_root(t: Optional[torch.Tensor], src: int, group=None):
    meta = [None]
    if dist.get_rank(group) == src:
        meta = [(tuple(t.shape), t.dtype)]
    dist.broadcast_object_list(meta, src=src, group=group)
    return meta[0]


def scatter_clips(clips: Optional[torch.Tensor], device: torch.device, src: int = 0, group=None) -> torch.Tensor:
    """rank `src` holds clips [N, ...]; every rank receives its contiguous block [n_r, ...] on `device`."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    shape, dtype = _meta_from_root(clips, src, group)
    counts = shard_counts(shape[0], world)
    width = max(counts)
    out = torch.empty((width,) + tuple(shape[1:]), dtype=dtype, device=device)
    chunks = None
    if rank == src:
        chunks = []
        for r in range(world):
            a, b = shard_range(shape[0], r, world)
            c = torch.zeros_like(out)
            c[: b - a] = clips[a:b].to(device)
            chunks.append(c)
    dist.scatter(out, chunks, src=src, group=group)
    return out[: counts[rank]]


def gather_ragged(local: torch.Tensor, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Concatenate per-rank results [n_r, ...] (n_r may differ, trailing dims equal) on rank `dst`, in rank order."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    counts = [int(s.item()) for s in sizes]
    width = max(counts)
    padded = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def run_sharded(fn: Callable[..., Sequence[torch.Tensor]], inputs: Sequence[Optional[torch.Tensor]], device: torch.device,
                src: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """scatter every input from rank `src`, run `fn(*local_inputs)` (the per-GPU hot path, e.g. tokenize+detokenize),
    gather every output back to `src`.  Ranks whose shard is empty skip `fn`."""
    local = [scatter_clips(t, device, src, group) for t in inputs]
    if local[0].shape[0] > 0:
        outs = list(fn(*local))
        template = [(tuple(o.shape[1:]), o.dtype) for o in outs]
    else:
        outs, template = None, None
    # ranks with an empty shard need the output signature to build their (empty) contribution
    sigs = [None] * dist.get_world_size(group)
    dist.all_gather_object(sigs, template, group=group)
    sig = next(s for s in sigs if s is not None)
    if outs is None:
        outs = [torch.empty((0,) + shp, dtype=dt, device=device) for shp, dt in sig]
    gathered = [gather_ragged(o, src, group) for o in outs]
    return gathered if dist.get_rank(group) == src else None
