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
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() == "ok"
