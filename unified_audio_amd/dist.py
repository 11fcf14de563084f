"""Multi-GPU driver: utterances (and UniSE 5 s segments) are independent, so the hot path shards by clips with NO
collective on the data path; the only exchange steps are the trivial scatter of clips from rank 0 and the gather of
codes / waveforms back.  The reference does rank-strided sharding with no communication at all
(QuarkAudio-UniSE/dataloader/data_module.py:364); this module adds the optional single-collector convenience.

One process per GPU (`torchrun`), `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" in the
CPU tests.  Transfers are direct rank0 <-> rank_i (scatter / gather), never an all-reduce: nothing is reduced.
`src` / `dst` and every rank in here are ranks OF THE GROUP passed in (group=None: the world); they are translated to global ranks
where torch.distributed wants those (`_peer`), so the drivers work unchanged inside a sub-group (e.g. one group per node).
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
    dist.broadcast_object_list(meta, src=_peer(group, src), group=group)
    return meta[0]


def _peer(group, group_rank: int) -> int:
    """P2POp / send / recv address a peer by its GLOBAL rank; the partitions here are computed in group-local ranks."""
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def _exchange(ops) -> None:
    """Run a list of point-to-point operations as ONE group (RCCL: ncclGroupStart / End, so rank 0's transfers to its peers ride
    their own xGMI links concurrently instead of one after the other) and wait for all of them."""
    if not ops:
        return
    for w in dist.batch_isend_irecv(ops):
        w.wait()


def scatter_clips(clips: Optional[torch.Tensor], device: torch.device, src: int = 0, group=None) -> torch.Tensor:
    """rank `src` holds clips [N, ...]; every rank receives its contiguous block [n_r, ...] on `device`.

    Packed point-to-point form: rank `src` sends rank r exactly the rows [a_r, b_r) of `clips` - a contiguous view,
    no padding to the widest block, no per-rank staging copy of the batch - and keeps its own block as a view (moved to `device`
    only if `clips` lives elsewhere).  All sends of a call are posted as one group."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    shape, dtype = _meta_from_root(clips, src, group)
    a, b = shard_range(shape[0], rank, world)
    if rank == src:
        clips = clips.contiguous()
        ops = []
        for r in range(world):
            ra, rb = shard_range(shape[0], r, world)
            if r != src and rb > ra:
                ops.append(dist.P2POp(dist.isend, clips[ra:rb].to(device), _peer(group, r), group))
        _exchange(ops)
        return clips[a:b].to(device)
    out = torch.empty((b - a,) + tuple(shape[1:]), dtype=dtype, device=device)
    if b > a:
        _exchange([dist.P2POp(dist.irecv, out, _peer(group, src), group)])
    return out


def gather_ragged(local: torch.Tensor, dst: int = 0, group=None, pad_value=0) -> Optional[torch.Tensor]:
    """Concatenate per-rank results [n_r, ...] on rank `dst`, in rank order.  n_r may differ; trailing dimensions may differ too
    (H-Codec 1.5: the number of groups G of a batch is data dependent, so every rank's codes are [n_r, nq, G_r]) - they are
    right-padded with `pad_value` to the largest extent over the ranks (for length-injected codes pass -codebook_size: a group of
    length 0, exactly what a batch's own shorter clips carry, codec_adaptive.py:68-80).

    Packed point-to-point form: every rank sends its result in ITS OWN shape (no padding on the wire); `dst` receives a block
    straight into its rows of the result when the trailing extents agree, through a staging tensor of the sender's shape otherwise."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    shp = torch.tensor(list(local.shape), dtype=torch.int64, device=local.device)
    shapes = [torch.zeros_like(shp) for _ in range(world)]
    dist.all_gather(shapes, shp, group=group)
    shapes = [[int(v) for v in t.tolist()] for t in shapes]
    if rank != dst:
        if local.numel() > 0:
            _exchange([dist.P2POp(dist.isend, local.contiguous(), _peer(group, dst), group)])
        return None
    # a rank with an empty shard has no trailing extents of its own (run_sharded builds them from another rank's signature)
    trail = tuple(max(t[i] for t in shapes) for i in range(1, local.dim()))
    total = sum(t[0] for t in shapes)
    ragged = any(tuple(t[1:]) != trail for t in shapes if t[0] > 0)
    result = (torch.full if ragged else torch.empty)((total,) + trail, *((pad_value,) if ragged else ()), dtype=local.dtype, device=local.device)
    ops, staged, at = [], [], 0
    for r, t in enumerate(shapes):
        n = t[0]
        rows = result[at:at + n]
        at += n
        if n == 0 or (r != dst and min(t) == 0):
            continue
        if r == dst:
            rows[tuple(slice(None) if i == 0 else slice(0, e) for i, e in enumerate(local.shape))] = local
        elif tuple(t[1:]) == trail:
            ops.append(dist.P2POp(dist.irecv, rows, _peer(group, r), group))
        else:
            buf = torch.empty(t, dtype=local.dtype, device=local.device)
            ops.append(dist.P2POp(dist.irecv, buf, _peer(group, r), group))
            staged.append((rows, buf))
    _exchange(ops)
    for rows, buf in staged:
        rows[tuple(slice(None) if i == 0 else slice(0, e) for i, e in enumerate(buf.shape))] = buf
    return result


def _tick(device: torch.device) -> float:
    import time

    if device.type == "cuda":
        torch.cuda.synchronize(device)
    return time.perf_counter()


def run_sharded(fn: Callable[..., Sequence[torch.Tensor]], inputs: Sequence[Optional[torch.Tensor]], device: torch.device,
                src: int = 0, group=None, pad_values: Optional[Sequence] = None, timings: Optional[dict] = None) -> Optional[List[torch.Tensor]]:
    """scatter every input from rank `src`, run `fn(*local_inputs)` (the per-GPU hot path, e.g. tokenize+detokenize),
    gather every output back to `src` (`timings`: a dict that receives this rank's scatter_s / compute_s / gather_s).  Ranks whose shard is empty skip `fn`.  `pad_values[i]` fills output i where a rank's
    trailing dimensions are shorter than another's (gather_ragged).  If `fn` raises on any rank, EVERY rank raises ShardError."""
    t0 = _tick(device) if timings is not None else 0.0
    local = [scatter_clips(t, device, src, group) for t in inputs]
    t1 = _tick(device) if timings is not None else 0.0
    outs, template, err = None, None, None
    try:
        if local[0].shape[0] > 0:
            outs = list(fn(*local))
            template = [(tuple(o.shape[1:]), o.dtype) for o in outs]
    except Exception as e:  # noqa: BLE001  (reported to every rank below: nobody may be left waiting in the gather)
        err = f"{type(e).__name__}: {e}"
    t2 = _tick(device) if timings is not None else 0.0
    _raise_if_any_failed(err, group)
    # ranks with an empty shard need the output signature to build their (empty) contribution
    sigs = [None] * dist.get_world_size(group)
    dist.all_gather_object(sigs, template, group=group)
    sig = next((s for s in sigs if s is not None), None)
    if sig is None:  # no rank had an item: there is no output signature to build the (empty) results from
        raise ValueError("run_sharded: the inputs hold no items")
    if outs is None:
        outs = [torch.empty((0,) + shp, dtype=dt, device=device) for shp, dt in sig]
    pads = list(pad_values) if pad_values is not None else [0] * len(outs)
    gathered = [gather_ragged(o, src, group, pv) for o, pv in zip(outs, pads)]
    if timings is not None:  # this rank's wall time of the three phases: the exchange steps next to - never inside - the compute
        t3 = _tick(device)
        timings.update(scatter_s=t1 - t0, compute_s=t2 - t1, gather_s=t3 - t2)
    return gathered if dist.get_rank(group) == src else None


# ---------------------------------------------------------------------------------------------------------------------
# Ragged utterance lists ("sorted by length to balance").  The reference feeds one utterance per step and shards the
# file list rank-strided (QuarkAudio-UniSE/dataloader/data_module.py:340,364), which leaves the ranks with whatever total length
# the stride happens to give them; here the list is partitioned by LENGTH so that every rank gets about the same number of samples.

class ShardError(RuntimeError):
    """Raised on EVERY rank when the hot path failed on some rank (so that no rank is left waiting in a collective)."""


def balanced_partition(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time partition: utterances sorted by length (longest first, ties by index) are dealt one by one to the
    rank with the smallest total so far (ties: lowest rank).  Returns the utterance indices of every rank, longest first -
    deterministic, every index exactly once, max load <= (4/3 - 1/(3 world)) x optimal (Graham's bound)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads, parts = [0] * world, [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += int(lengths[i])
    return parts


def _raise_if_any_failed(err: Optional[str], group=None) -> None:
    errs = [None] * dist.get_world_size(group)
    dist.all_gather_object(errs, err, group=group)
    bad = [(r, e) for r, e in enumerate(errs) if e is not None]
    if bad:
        raise ShardError("; ".join(f"rank {r}: {e}" for r, e in bad))


def _pack(tensors: Sequence[torch.Tensor], device: torch.device, dtype: torch.dtype) -> torch.Tensor:
    if not tensors:
        return torch.empty(0, dtype=dtype, device=device)
    return torch.cat([t.reshape(-1).to(device=device, dtype=dtype) for t in tensors])


def run_sharded_ragged(fn: Callable[[List[torch.Tensor]], Sequence[torch.Tensor]], utterances: Optional[Sequence[torch.Tensor]],
                       device: torch.device, src: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """rank `src` holds `utterances` (1-D tensors of different lengths, one dtype); they are partitioned by length
    (balanced_partition), every rank receives its share as ONE packed point-to-point transfer, runs
    `fn(list of its utterances) -> one 1-D tensor per utterance` (the per-GPU hot path; results may be of any length), and the results
    travel back the same way.  Returns, on `src`, the results in the ORIGINAL order (None elsewhere).  A rank with no utterance
    skips `fn`.  If `fn` raises anywhere, every rank raises ShardError instead of blocking in the gather."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    plan = [None]
    if rank == src:
        lengths = [int(u.numel()) for u in utterances]
        plan = [(balanced_partition(lengths, world), lengths, utterances[0].dtype if utterances else torch.float32)]
    dist.broadcast_object_list(plan, src=_peer(group, src), group=group)
    parts, lengths, dtype = plan[0]
    mine = parts[rank]
    # ---- scatter: one packed tensor per destination rank
    if rank == src:
        _exchange([dist.P2POp(dist.isend, _pack([utterances[i] for i in parts[r]], device, dtype), _peer(group, r), group)
                   for r in range(world) if r != src and parts[r]])
        local = [utterances[i].reshape(-1).to(device) for i in mine]
    else:
        packed = torch.empty(sum(lengths[i] for i in mine), dtype=dtype, device=device)
        if mine:  # batched on BOTH ends: an unbatched recv against the root's batched isend would, under a lazily
            # initialised ProcessGroupNCCL, open a 2-rank communicator the root never joins
            _exchange([dist.P2POp(dist.irecv, packed, _peer(group, src), group)])
        local, at = [], 0
        for i in mine:
            local.append(packed[at:at + lengths[i]])
            at += lengths[i]
    # ---- the hot path; a failure is reported to everybody before anyone enters the gather
    outs, err = [], None
    try:
        if local:
            outs = [o.reshape(-1) for o in fn(local)]
            if len(outs) != len(local):
                raise ValueError(f"fn returned {len(outs)} results for {len(local)} utterances")
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    _raise_if_any_failed(err, group)
    # ---- gather: lengths first (objects), then one packed transfer per rank
    meta = [None] * world
    dist.all_gather_object(meta, ([int(o.numel()) for o in outs], outs[0].dtype if outs else None), group=group)
    if rank != src:
        if outs:
            _exchange([dist.P2POp(dist.isend, _pack(outs, device, outs[0].dtype), _peer(group, src), group)])
        return None
    result: List[Optional[torch.Tensor]] = [None] * len(lengths)
    bufs = {r: torch.empty(sum(meta[r][0]), dtype=meta[r][1], device=device) for r in range(world) if r != src and meta[r][0]}
    _exchange([dist.P2POp(dist.irecv, buf, _peer(group, r), group) for r, buf in bufs.items()])  # every peer's packed results as one group
    for r in range(world):
        lens, _ = meta[r]
        if not lens:
            continue
        if r == src:
            pieces = outs
        else:
            pieces, at = [], 0
            for n in lens:
                pieces.append(bufs[r][at:at + n])
                at += n
        for i, p in zip(parts[r], pieces):
            result[i] = p
    return result
