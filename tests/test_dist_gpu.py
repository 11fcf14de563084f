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


def _worker_world1(port, q):
    """One rank, backend "nccl": RCCL really initialises on the device and every collective run_sharded / run_sharded_ragged /
    bench.py's N > 1 legs issue (object broadcast / all-gather, int64 all_gather of shapes, all_reduce MAX, barrier) executes on it -
    with one rank there is no peer, so what this catches is misuse of the backend (host tensors handed to RCCL, a missing device,
    object collectives without a current device), which the gloo tests cannot see."""
    import torch.distributed as dist

    import unified_audio_amd as qa
    from oracle import hcodec_ref as R
    from oracle import synth
    from tests.util import MINI
    from unified_audio_amd import dist as qd

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sd = synth.hcodec10_state_dict(9, R.HCodecSpec(**MINI))
        tok = qa.HCodecTokenizer(state_dict=sd, device=dev, spec=qa.HCodecSpec(**MINI))
        wav = synth.synth_wav(1, 3, 16 * 12).to(dev)
        feats = synth.synth_feat(2, 3, 24, 64).transpose(1, 2).contiguous().to(dev)

        def hot_path(w, f):
            ac, sc = tok.tokenize(w, feats=f)
            return ac, sc, tok.detokenize(ac, sc)

        tm = {}
        out = qd.run_sharded(hot_path, [wav, feats], dev, timings=tm)
        ref = hot_path(wav, feats)
        ok = all(torch.equal(a, b) for a, b in zip(out, ref)) and set(tm) == {"scatter_s", "compute_s", "gather_s"}
        utts = [torch.full((n,), float(i), device=dev) for i, n in enumerate((7, 3, 5))]
        rag = qd.run_sharded_ragged(lambda us: [u * 2 for u in us], utts, dev)
        ok = ok and all(torch.equal(a, b * 2) for a, b in zip(rag, utts))
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)  # bench.py's max-over-ranks reduction and its fence
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize(dev)
        seen = [None]
        dist.all_gather_object(seen, {"uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", ""))})
        q.put("ok" if ok and float(t.item()) == 1.5 and seen[0] is not None else "mismatch")
    finally:
        dist.destroy_process_group()


def test_rccl_backend_initialises_and_runs_the_sharded_path_with_one_rank(qa_lib, gpu_device):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_worker_world1, args=(_free_port(), q), daemon=True)
    p.start()
    p.join(300)
    if p.exitcode is None:
        p.kill()
        p.join(10)
        pytest.fail("the single-rank RCCL worker hung")
    assert p.exitcode == 0
    assert not q.empty() and q.get() == "ok"


def test_plain_bench_gpus_2_on_a_one_gpu_box_meets_then_says_what_is_missing(qa_lib, gpu_device):
    """VERDICT r03 item 3, on the hardware: `python bench.py --gpus 2` as a plain command self-launches two ranks; on a box with one
    MI355X they rendezvous, rank 0 prints one JSON line naming the missing device, and every rank exits non-zero - no hang, no traceback
    in place of a reason.  (On a box with >= 2 GPUs the command would simply run: not asserted here, the driver's scaling run does that.)"""
    import json
    import subprocess
    import sys

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has >= 2 GPUs: `bench.py --gpus 2` would run the real thing")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300, env=env, cwd=root)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode != 0
    assert len(lines) == 1 and lines[0]["rendezvous"] == "ok" and lines[0]["n_gpus"] == 2 and lines[0]["devices_visible"] == 1, p.stdout + p.stderr[-1500:]
