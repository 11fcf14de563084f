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


def gather_ragged(local: torch.Tensor, dst: int = 0, group=None, pad_value=0) -> Optional[torch.Tensor]:
    """Concatenate per-rank results [n_r, ...] on rank `dst`, in rank order.  n_r may differ; trailing dimensions may differ too
    (H-Codec 1.5: the number of groups G of a batch is data dependent, so every rank's codes are [n_r, nq, G_r]) - they are
    right-padded with `pad_value` to the largest extent over the ranks (for length-injected codes pass -codebook_size: a group of
    length 0, exactly what a batch's own shorter clips carry, codec_adaptive.py:68-80)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    shp = torch.tensor(list(local.shape), dtype=torch.int64, device=local.device)
    shapes = [torch.zeros_like(shp) for _ in range(world)]
    dist.all_gather(shapes, shp, group=group)
    shapes = [[int(v) for v in t.tolist()] for t in shapes]
    counts = [t[0] for t in shapes]
    # a rank with an empty shard has no trailing extents of its own (run_sharded builds them from another rank's signature)
    full = tuple(max(t[i] for t in shapes) for i in range(local.dim()))
    padded = torch.full(full, pad_value, dtype=local.dtype, device=local.device)
    padded[tuple(slice(0, n) for n in local.shape)] = local
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def run_sharded(fn: Callable[..., Sequence[torch.Tensor]], inputs: Sequence[Optional[torch.Tensor]], device: torch.device,
                src: int = 0, group=None, pad_values: Optional[Sequence] = None) -> Optional[List[torch.Tensor]]:
    """scatter every input from rank `src`, run `fn(*local_inputs)` (the per-GPU hot path, e.g. tokenize+detokenize),
    gather every output back to `src`.  Ranks whose shard is empty skip `fn`.  `pad_values[i]` fills output i where a rank's
    trailing dimensions are shorter than another's (gather_ragged)."""
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
    pads = list(pad_values) if pad_values is not None else [0] * len(outs)
    gathered = [gather_ragged(o, src, group, pv) for o, pv in zip(outs, pads)]
    return gathered if dist.get_rank(group) == src else None
