"""bench.py's N > 1 machinery without a GPU: the plain-command self-launch, the rendezvous, the ranks_seen
all-gather, the scatter -> hot path -> gather exchange leg (unified_audio_amd.dist.run_sharded, packed point-to-point transfers) and the
one JSON line from rank 0 - on gloo, world 2, with bench.py's own stub hot path (--bootstrap-selftest).  The 8-GPU scaling run is the
driver's to launch; what can break before the first kernel is covered here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, *argv], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def _check_selftest_line(line, world):
    assert line["selftest"] is True and line["value"] is None and line["n_gpus"] == world
    assert line["ranks_seen"]["distinct_devices"] == world
    assert sorted(r["rank"] for r in line["ranks_seen"]["ranks"]) == list(range(world))
    ex = line["exchange"]
    assert ex["gathered_equals_unsharded_run"] is True
    assert ex["gathered_shapes"] == [[3 * world, 8], [3 * world, 64]]
    assert ex["exchange_ms"] == pytest.approx(ex["scatter_ms"] + ex["gather_ms"])
    assert line["max_over_ranks"] == float(world)
    # the two multi-GPU BASELINE configurations have their own objects in an N > 1 line (here through the stub hot paths
    # of the real signatures): whole-job value over all ranks, weak scaling, and an exchange block whose rank-0 rows equal rank 0's own run
    for key, unit, n_out in (("configs3_tse", "tokens/sec", 2), ("configs4_hcodec20", "audio-seconds/sec", 3)):
        leg = line[key]
        assert leg["unit"] == unit and leg["n_gpus"] == world and leg["scaling"] == "weak" and leg["value"] > 0 and leg["ms_per_step"] > 0
        assert leg["config"]["baseline_config"].startswith("configs[")
        lx = leg["exchange"]
        assert "error" not in lx, lx
        assert lx["rank0_block_matches_local_run"] is True
        assert len(lx["gathered_shapes"]) == n_out and all(sh[0] == 3 * world for sh in lx["gathered_shapes"])
        assert lx["exchange_ms"] == pytest.approx(lx["scatter_ms"] + lx["gather_ms"])


def test_plain_command_self_launches_one_rank_per_gpu():
    """`python bench.py --gpus 2` with no launcher around it re-executes itself under torch.distributed.run (it used to SystemExit)."""
    p, lines = _run([BENCH, "--gpus", "2", "--bootstrap-selftest"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout  # ONE line, from rank 0
    assert lines[0]["launched_by"] == "torch.distributed.run"
    _check_selftest_line(lines[0], 2)


def test_driver_form_torchrun_world_3():
    """the driver's own N > 1 command line; world 3 makes the block partition of the 9 clips uneven-free but the gather three-sided."""
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    p, lines = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1", "--master-port", str(port),
                     BENCH, "--gpus", "3", "--steps", "1", "--warmup", "0", "--bootstrap-selftest"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    _check_selftest_line(lines[0], 3)


def test_single_rank_selftest_needs_no_process_group():
    p, lines = _run([BENCH, "--bootstrap-selftest"])
    assert p.returncode == 0 and lines[0]["n_gpus"] == 1 and "ranks_seen" not in lines[0]


@pytest.mark.skipif(__import__("torch").cuda.is_available() and __import__("torch").cuda.device_count() >= 2, reason="needs a node with fewer than 2 GPUs")
def test_more_ranks_than_devices_fails_after_the_rendezvous_with_a_clear_message():
    """The real (non-selftest) path with N = 2 on a box that has 0 or 1 GPU: the ranks meet, rank 0 says what is missing, every rank
    leaves with a non-zero status - nobody hangs in a collective, no traceback instead of a reason."""
    p, lines = _run([BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert p.returncode != 0
    assert len(lines) == 1 and lines[0]["rendezvous"] == "ok" and lines[0]["n_gpus"] == 2 and lines[0]["devices_visible"] < 2
    assert "exposes only" in lines[0]["error"]


def test_a_leg_that_fails_on_one_rank_ends_on_all_ranks_together():
    """N > 1 robustness: the multi-GPU legs build their own models; if that fails on ONE rank (out of memory, say) the others must not be
    left in the leg's barrier.  Every fallible phase is followed by an all-gather of error strings (bench.all_ranks_ok): here rank 1's setup of
    the configs[4] leg dies - the run still ends in time with ONE line, that leg an error object naming the rank, the other leg intact."""
    p, lines = _run([BENCH, "--gpus", "2", "--bootstrap-selftest"], env_extra={"QA_SELFTEST_FAIL_RANK": "1"}, timeout=180)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    leg = lines[0]["configs4_hcodec20"]
    assert "error" in leg and "rank 1" in leg["error"] and "injected failure" in leg["error"], leg
    assert lines[0]["configs3_tse"]["exchange"]["rank0_block_matches_local_run"] is True
