"""Multi-GPU readiness on real devices: two ranks, one GPU each, torch.distributed backend "nccl" (= RCCL over xGMI on ROCm),
run_sharded(tokenize + detokenize) on the real codec: the gathered codes and waveforms must equal the single-GPU result bit for
bit (clips are independent; the only exchange is the scatter of clips and the gather of results).  Skips on a 1-GPU box (the
driver's 8-GPU scaling run is where N > 1 executes); the same code path runs under gloo on CPU in tests/test_dist_cpu.py."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist

    import unified_audio_amd as qa
    from oracle import hcodec_ref as R
    from oracle import synth
    from tests.util import MINI
    from unified_audio_amd import dist as qd

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        sd = synth.hcodec10_state_dict(9, R.HCodecSpec(**MINI))
        tok = qa.HCodecTokenizer(state_dict=sd, device=dev, spec=qa.HCodecSpec(**MINI))
        n = 5  # uneven: 3 + 2 clips
        wav = synth.synth_wav(1, n, 16 * 12).to(dev) if rank == 0 else None
        feats = synth.synth_feat(2, n, 24, 64).transpose(1, 2).contiguous().to(dev) if rank == 0 else None

        def hot_path(w, f):
            ac, sc = tok.tokenize(w, feats=f)
            return ac, sc, tok.detokenize(ac, sc)

        out = qd.run_sharded(hot_path, [wav, feats], dev)
        if rank == 0:
            ref = hot_path(wav, feats)
            ok = all(torch.equal(a, b) for a, b in zip(out, ref))
            q.put("ok" if ok else "mismatch")
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_run_sharded_on_two_gpus_equals_single_gpu(qa_lib, gpu_device):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the round-end GPU box has one; the driver's scaling run covers N > 1)")
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    hung = [i for i, p in enumerate(procs) if p.exitcode is None]
    for p in procs:  # never leave a rank behind (a hung RCCL rendezvous would block pytest at exit)
        if p.exitcode is None:
            p.kill()
            p.join(10)
    assert not hung, f"ranks {hung} hung"
    assert [p.exitcode for p in procs] == [0, 0]
    assert not q.empty() and q.get() == "ok"
