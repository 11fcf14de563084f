"""The N > 1 path on CPU: world_size-2 (and 3, with an empty shard) gloo processes exercise shard / scatter / gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unified_audio_amd import dist as qd


def test_shard_range_partitions_exactly():
    for n in (0, 1, 5, 32, 64, 127):
        for w in (1, 2, 3, 8):
            spans = [qd.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == qd.shard_counts(n, w)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(target, world, args_of_rank, timeout=180):
    """Start one process per rank, wait, and NEVER leave a child behind: a rank that hangs (rendezvous, a collective nobody else
    entered) is killed by PID and the test fails, instead of pytest blocking at exit on a non-daemon child."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args_of_rank) + (q,), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    import time

    deadline = time.monotonic() + timeout
    for p in procs:
        p.join(max(0.0, deadline - time.monotonic()))
    hung = [i for i, p in enumerate(procs) if p.exitcode is None]
    for p in procs:
        if p.exitcode is None:
            p.kill()
            p.join(10)
    assert not hung, f"ranks {hung} did not finish within {timeout} s"
    codes = [p.exitcode for p in procs]
    assert codes == [0] * world, f"rank exit codes {codes}"
    return q


def _drain(q, n):
    out = []
    for _ in range(n):
        assert not q.empty(), f"only {len(out)} of {n} ranks reported"
        out.append(q.get())
    return out


def _worker(rank, world, port, n_clips, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        clips = torch.arange(n_clips * 6, dtype=torch.float32).view(n_clips, 6) if rank == 0 else None
        feats = torch.arange(n_clips * 2 * 3, dtype=torch.float32).view(n_clips, 2, 3) + 1000 if rank == 0 else None
        local = qd.scatter_clips(clips, dev)
        a, b = qd.shard_range(n_clips, rank, world)
        assert local.shape == (b - a, 6) and (b == a or float(local[0, 0]) == a * 6)

        def fake_hot_path(wav, feat):  # stands in for tokenize + detokenize: per-clip, no cross-clip dependence
            codes = (wav[:, :4] * 2).to(torch.int64).view(-1, 2, 2)
            return codes, wav * 0.5 + feat.sum(dim=(1, 2))[:, None]

        out = qd.run_sharded(fake_hot_path, [clips, feats], dev)
        if rank == 0:
            codes, wav = out
            ref_codes, ref_wav = fake_hot_path(clips, feats)
            assert torch.equal(codes, ref_codes) and torch.equal(wav, ref_wav)
            q.put("ok")
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_clips", [(2, 5), (2, 4), (3, 2)])
def test_scatter_run_gather_gloo(world, n_clips):
    q = _run_ranks(_worker, world, (n_clips,))
    assert not q.empty() and q.get() == "ok"


def _worker_ragged_tail(rank, world, port, q):
    """H-Codec 1.5's shape problem: every rank's codes are [n_r, nq, G_r] with a data-dependent G_r."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        n_clips, K = 5, 1024
        clips = torch.arange(n_clips * 4, dtype=torch.float32).view(n_clips, 4) if rank == 0 else None

        def fake_adaptive(wav):  # G grows with the rank's first clip index: ranks disagree on the trailing extent
            G = 2 + int(wav[0, 0].item()) // 4
            codes = (wav[:, :1].to(torch.int64) + torch.arange(G)).view(-1, 1, G).expand(-1, 3, G).contiguous()
            return codes, wav * 2

        out = qd.run_sharded(fake_adaptive, [clips], dev, pad_values=[-K, 0.0])
        if rank == 0:
            codes, wav = out
            assert torch.equal(wav, clips * 2)
            g0, g1 = 2, 2 + 3  # rank 0 holds clips 0..2, rank 1 clips 3..4 (first clip index 3 -> wav[0, 0] = 12)
            assert codes.shape == (n_clips, 3, g1)
            assert torch.equal(codes[:3, :, :g0], (clips[:3, :1].to(torch.int64) + torch.arange(g0)).view(-1, 1, g0).expand(-1, 3, g0))
            assert bool((codes[:3, :, g0:] == -K).all())  # padded groups: length 0 in the length-injected format
            assert torch.equal(codes[3:], (clips[3:, :1].to(torch.int64) + torch.arange(g1)).view(-1, 1, g1).expand(-1, 3, g1))
            q.put("ok")
    finally:
        dist.destroy_process_group()


def test_gather_pads_data_dependent_trailing_dims_gloo():
    q = _run_ranks(_worker_ragged_tail, 2, ())
    assert not q.empty() and q.get() == "ok"


def _worker_subgroup(rank, world, port, q):
    """The drivers inside a SUB-group (e.g. one group per node): global ranks 1 and 2 of a world of 3 form the group, its rank 0 (global 1)
    is the root.  `src` / `dst` are group ranks; torch.distributed's point-to-point calls and object broadcasts want global ranks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grp = dist.new_group(ranks=[1, 2])  # every rank of the world calls new_group
        if rank == 0:
            q.put("idle")
            return
        dev = torch.device("cpu")
        root = dist.get_rank(grp) == 0
        clips = torch.arange(5 * 6, dtype=torch.float32).view(5, 6) if root else None

        def hot(wav):
            return (wav[:, :2] * 3).to(torch.int64), wav + 1.0

        out = qd.run_sharded(hot, [clips], dev, group=grp)
        utts = [torch.full((n,), float(i)) for i, n in enumerate((9, 2, 4, 7))] if root else None
        rag = qd.run_sharded_ragged(lambda us: [u * 2 for u in us], utts, dev, group=grp)
        if root:
            ok = all(torch.equal(a, b) for a, b in zip(out, hot(clips))) and all(torch.equal(a, b * 2) for a, b in zip(rag, utts))
            q.put("ok" if ok else "mismatch")
        else:
            assert out is None and rag is None
            q.put("peer")
    finally:
        dist.destroy_process_group()


def test_drivers_work_inside_a_subgroup_gloo():
    q = _run_ranks(_worker_subgroup, 3, ())
    assert sorted(_drain(q, 3)) == ["idle", "ok", "peer"]


# --------------------------------------------------------------------------------------- ragged lists, failure propagation

def test_balanced_partition_properties():
    import random

    rnd = random.Random(3)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 64):
            lengths = [rnd.randint(16000, 16000 * 60) for _ in range(n)]
            parts = qd.balanced_partition(lengths, world)
            assert sorted(i for p in parts for i in p) == list(range(n))            # every utterance exactly once
            loads = [sum(lengths[i] for i in p) for p in parts]
            if n >= world:
                # LPT (Graham): makespan <= 4/3 OPT, and OPT >= max(mean load, longest utterance)
                assert max(loads) <= (4 / 3) * max(sum(lengths) / world, max(lengths)) + 1
            assert parts == qd.balanced_partition(lengths, world)                   # deterministic
            for p in parts:
                assert [lengths[i] for i in p] == sorted((lengths[i] for i in p), reverse=True)  # longest first inside a rank
    # the block split of the same list is far less even: 2 long + 6 short utterances over 2 ranks
    lengths = [60, 60, 1, 1, 1, 1, 1, 1]
    lpt = [sum(lengths[i] for i in p) for p in qd.balanced_partition(lengths, 2)]
    assert max(lpt) == 63 and sum(lengths[:4]) == 122


def _ragged_worker(rank, world, port, lengths, fail_rank, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        utts = [torch.arange(n, dtype=torch.float32) + 1000 * i for i, n in enumerate(lengths)] if rank == 0 else None

        def hot_path(local):  # per-utterance, output length differs from the input's (like codes / padded waveforms)
            if rank == fail_rank:
                raise ValueError("device fault on this rank")
            return [torch.cat([u * 2, u[:1]]) for u in local]

        try:
            out = qd.run_sharded_ragged(hot_path, utts, dev)
        except qd.ShardError as e:
            q.put(("error", rank, str(e)))
            return
        if rank == 0:
            ok = len(out) == len(lengths) and all(torch.equal(o, torch.cat([u * 2, u[:1]])) for o, u in zip(out, utts))
            q.put(("ok" if ok else "mismatch", rank, ""))
        else:
            assert out is None
            q.put(("none", rank, ""))
    finally:
        dist.destroy_process_group()


def _run_ragged(world, lengths, fail_rank=-1):
    q = _run_ranks(_ragged_worker, world, (lengths, fail_rank))
    return sorted(_drain(q, world))


@pytest.mark.parametrize("world,lengths", [(2, [50, 7, 33, 12, 90]), (3, [5, 400, 17, 17, 230, 1, 64]), (3, [9, 4])])
def test_run_sharded_ragged_gloo(world, lengths):
    """Utterances of different lengths, partitioned by length, one packed transfer per rank each way, results back in the original
    order (incl. a rank with nothing to do)."""
    res = _run_ragged(world, lengths)
    assert [r[0] for r in res] == ["none"] * (world - 1) + ["ok"], res


def test_a_failing_rank_raises_everywhere_instead_of_hanging_gloo():
    """The hot path throws on rank 1 of 3: every rank must raise ShardError (naming the rank) rather than wait in the gather."""
    res = _run_ragged(3, [10, 20, 30, 40, 50, 60], fail_rank=1)
    assert [r[0] for r in res] == ["error"] * 3, res
    assert all("rank 1: ValueError: device fault" in r[2] for r in res)


def _fail_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clips = torch.ones(4, 6) if rank == 0 else None

        def hot_path(w):
            if rank == 1:
                raise RuntimeError("out of memory")
            return (w * 2,)

        try:
            qd.run_sharded(hot_path, [clips], torch.device("cpu"))
            q.put("no error")
        except qd.ShardError as e:
            q.put("ShardError" if "rank 1: RuntimeError: out of memory" in str(e) else str(e))
    finally:
        dist.destroy_process_group()


def test_run_sharded_propagates_a_rank_failure_gloo():
    q = _run_ranks(_fail_worker, 2, ())
    assert sorted(_drain(q, 2)) == ["ShardError", "ShardError"]
