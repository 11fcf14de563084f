#!/usr/bin/env python
"""bench.py - headline benchmark of the MI355X-native QuarkAudio hot path (contract: see the task brief / DESIGN.md).

A "step" is one pass of the hot path - Codec.encode followed by Codec.decode - over one batch of synthetic clips that
is already resident in HBM (wav [B, T] and SSL features [B, N50, 768] as the tokenizer hands them over), on every rank.
`value` = audio-seconds processed by all ranks / max-over-ranks wall time of K steps.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--seconds 10]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 16000
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD at 2.4 GHz
CFG_NAMES = ["conv_gemm_kernel<128,32,4,1>", "conv_gemm_kernel<128,64,2,2>", "conv_gemm_kernel<128,128,2,2>", "conv_gemm_kernel<64,128,1,4>",
             "conv_gemm_kernel<64,64,2,2>"]
NPROF = 4 * len(CFG_NAMES)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU (BASELINE metric: b=32)")
    ap.add_argument("--seconds", type=float, default=10.0, help="clip length (BASELINE configs[1]: 10 s)")
    ap.add_argument("--model", choices=["1.5", "1.0", "2.0"], default="1.5",
                    help="H-Codec version (BASELINE configs[1] is 1.5; 2.0 = configs[4]'s per-GPU share: use --batch 16 --seconds 30)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=2, help="clips in the bounded CPU-baseline sample")
    ap.add_argument("--lean", action="store_true", help="profiling runs: only warm-up + timed region (no isolated / PCIe / LM / CPU passes)")
    ap.add_argument("--no-ssl", action="store_true", help="skip the secondary SSL front-end measurement")
    ap.add_argument("--no-lm", action="store_true", help="skip the secondary UniSE AR-LM tokens/sec measurement")
    ap.add_argument("--lm-batch", type=int, default=16, help="UniSE segments per GPU (BASELINE configs[2]: batch=16)")
    ap.add_argument("--no-extras", action="store_true", help="skip the widened secondaries (H-Codec 2.0 share of configs[4], TSE share of "
                                                               "configs[3], second grouping point, LM CPU baseline)")
    ap.add_argument("--no-multi-configs", action="store_true", help="N > 1: skip the configs[3] (TSE, 8 segments per GPU) and configs[4] (H-Codec 2.0, "
                                                                     "16 x 30 s per GPU) legs (`configs3_tse` / `configs4_hcodec20` objects of the line)")
    ap.add_argument("--cpu-baseline-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--bootstrap-selftest", action="store_true",
                    help="tests/test_bench_bootstrap_cpu.py: run ONLY the rank bootstrap (self-launch, rendezvous), the ranks_seen all-gather and the "
                         "scatter -> hot path -> gather exchange leg, on the gloo backend with host tensors and a STUB hot path (no codec, no "
                         "GPU, no throughput: the line carries \"selftest\": true and value null)")
    return ap.parse_args()


def _cpu_baseline_worker(clips, seconds, reps, threads, model="1.5"):
    """Runs in a child process (so a stuck host BLAS thread pool can never hang the bench): prints one JSON line."""
    import faulthandler

    faulthandler.dump_traceback_later(90, exit=True)
    torch.set_num_threads(threads)
    from oracle import hcodec_ref as R
    from oracle import synth

    from oracle import hcodec15_ref as R15

    spec = R.SPEC_15 if model == "1.5" else R.SPEC_10
    sd = synth.hcodec10_state_dict(1234, spec)
    T = int(round(seconds * SR / spec.enc_hop)) * spec.enc_hop
    wav = synth.synth_wav(101, clips, T)
    feat = synth.synth_feat(102, clips, T // 320, spec.sem_in)
    best = float("inf")
    with torch.no_grad():
        for i in range(reps + 1):  # first pass = warm-up
            t0 = time.perf_counter()
            if spec.adaptive:
                codes = R15.encode(sd, wav.unsqueeze(1), feat, spec)
                R15.decode(sd, codes["acoustic_codes"], codes["semantic_codes"], spec)
            else:
                ac, sc = R.encode(sd, wav.unsqueeze(1), feat, spec)
                R.decode(sd, ac, sc, spec)
            dt = time.perf_counter() - t0
            if i:
                best = min(best, dt)
    print(json.dumps({"value": clips * T / SR / best, "unit": "audio-seconds/sec", "cores": torch.get_num_threads(),
                      "kind": "port",
                      "sample": f"oracle/hcodec{'15' if spec.adaptive else ''}_ref.py (PyTorch-CPU restatement of the reference op sequence, H-Codec {model}) encode+decode of "
                                f"{clips} x {T / SR:.1f} s clips, best of {reps} after 1 warm-up"}))


def cpu_baseline(clips, seconds, reps=2, model="1.5"):
    """The oracle timed on this host's cores: kind = "port".  Bounded sample, hard timeout."""
    import subprocess

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, 32))  # the op sizes of a 2-clip sample stop scaling well before 32 threads
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", f"{clips},{seconds},{reps},{threads},{model}"],
                           capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001 - a failed baseline must not lose the GPU measurement
        return {"value": None, "unit": "audio-seconds/sec", "cores": threads, "kind": "port",
                "sample": f"FAILED: {type(e).__name__}: {str(e)[:300]}"}


def _ssl_state_dict(spec, seed=21):
    from unified_audio_amd.synth import ssl_state_dict  # seeded HF-layout weights; lives with the other generators

    return ssl_state_dict(spec, seed)


def ssl_bench(dev, model, B, seconds, reps=3):
    """Secondary: the SSL feature extraction in front of Codec.encode - XLSR-53 architecture for H-Codec 1.5
    (audio_tokenizer.py:47), HuBERT-base for 1.0 - on B clips of `seconds` at 16 kHz, waveform resident in HBM."""
    import unified_audio_amd as qa

    spec, name = {"1.5": (qa.SPEC_XLSR53, "wav2vec2-large-xlsr-53"), "unise": (qa.SPEC_WAVLM_BASE_PLUS, "wavlm-base-plus")}.get(
        model, (qa.SPEC_HUBERT_BASE, "hubert_base"))
    sd = _ssl_state_dict(spec)
    n_params = sum(v.numel() for v in sd.values())
    fx = qa.SSLFeatureExtractor(spec, device=dev).load_state_dict(sd)
    del sd
    T = int(seconds * 16000)
    g = torch.Generator().manual_seed(5)
    wav = (torch.randn(B, T, generator=g) * 0.1).to(dev)
    feats = fx(wav)
    torch.cuda.synchronize(dev)
    assert torch.isfinite(feats).all()
    t0 = time.perf_counter()
    for _ in range(reps):
        feats = fx(wav)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    # contraction FLOPs of one pass (conv stack from layer 1, projections, positional conv, encoder GEMMs + attention)
    L = T + 2 * spec.pad
    fl, cin = 0.0, 1
    for i, (c, k, st) in enumerate(zip(spec.conv_dim, spec.conv_kernel, spec.conv_stride)):
        L = (L - k) // st + 1
        fl += 2.0 * L * c * cin * k
        cin = c
    d, inter = spec.hidden_size, spec.intermediate_size
    fl += 2.0 * L * d * cin + 2.0 * L * d * (d // spec.num_conv_pos_embedding_groups) * spec.num_conv_pos_embeddings
    fl += spec.num_hidden_layers * (2.0 * L * d * (4 * d + 2 * inter) + 4.0 * L * L * d)
    return {"metric": "audio-seconds/sec SSL feature extraction (front-end of Codec.encode)", "value": B * seconds / dt, "unit": "audio-seconds/sec",
            "ms_per_pass": 1e3 * dt, "tflops": B * fl / dt / 1e12,
            "config": {"workload": f"{name} architecture ({n_params / 1e6:.0f} M parameters, seeded random weights), {B} clips x {seconds:.0f} s @16 kHz, "
                                   f"{L} frames x {d} per clip, hidden-state average" + (" + |x|^0.3 compression" if spec.compress_exponent > 0 else ""),
                       "dtype": "f32"}}


HBM_PEAK_TBS = 8.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured float4 copy)


def _lm_cpu_worker(batch, n_mix, steps, threads):
    """Child process: oracle/llm_ref.py generate (greedy) on the host cores - the CPU baseline of the LM metric."""
    import faulthandler

    faulthandler.dump_traceback_later(150, exit=True)
    torch.set_num_threads(threads)
    from oracle import llm_ref as L

    sd = L.lm_state_dict(4321)
    mix = L.synth_feats(50, batch, n_mix)
    t0 = time.perf_counter()
    L.generate(sd, "se", None, mix, steps, 32, L.SPEC_UNISE)
    dt = time.perf_counter() - t0
    print(json.dumps({"value": batch * (33 + steps) / dt, "unit": "tokens/sec", "cores": torch.get_num_threads(), "kind": "port",
                      "sample": f"oracle/llm_ref.py (PyTorch-CPU restatement of LLM_SFT.generate, pinned to the reference's own class) greedy "
                                f"generate of {batch} segments, prompt {n_mix + 2}, 33 + {steps} steps (one pass, prefill included)"}))


def lm_cpu_baseline(batch=16, n_mix=250, steps=60):
    import subprocess

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, 32))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", f"lm,{batch},{n_mix},{steps},{threads}"],
                           capture_output=True, text=True, timeout=200, env=env, cwd=ROOT)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "tokens/sec", "cores": threads, "kind": "port", "sample": f"FAILED: {type(e).__name__}: {str(e)[:300]}"}


def lm_bench(dev, rank, world, dist, batch, reps=2, task="se", n_enroll=0):
    """Secondary metric of BASELINE.json: UniSE AR tokens/sec = B * (33 + N) generated tokens / wall time of generate()
    (prefill included), SE prompt of 252 embeddings (5 s segment) - or TSE prompt of 503 with an enrollment - 283 greedy steps,
    features resident in HBM.  The decode loop streams every weight and the whole KV cache once per step: `roofline` prices the
    step against HBM."""
    import unified_audio_amd as qa
    from unified_audio_amd import synth

    sd = synth.lm_state_dict(4321)
    lm = qa.LLM_SFT(device=dev).load_state_dict(sd)
    mix = synth.synth_feats(50 + rank, batch, 250).to(dev)
    enr = synth.synth_feats(90 + rank, batch, n_enroll).to(dev) if n_enroll else None
    mel = torch.zeros(batch, 250, 80)
    best = float("inf")
    for i in range(reps + 1):
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        lm.generate(task, mel if enr is not None else None, enr, mel, mix, do_sample=False)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if i:
            best = min(best, dt)
    # prefill reported separately: a generate of 1 global + 1 semantic token is the same prefill + 2 decode steps
    short = float("inf")
    mel1 = torch.zeros(batch, 1, 80)
    for _ in range(2):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        lm.generate(task, mel1 if enr is not None else None, enr, mel1, mix, global_length=1, do_sample=False)
        torch.cuda.synchronize(dev)
        short = min(short, time.perf_counter() - t0)
    ms_decode = 1e3 * (best - short) / (283 - 2)
    ms_prefill = 1e3 * short - 2 * ms_decode
    # the prefill's dense GEMMs (adapter + 12 layers of QKV / O / SwiGLU over B x prompt rows) against the fp32 matrix peak - north_star's
    # "MFMA utilisation for the LM GEMMs": per-launch HIP events inside the library around a generate of 1 + 1 tokens
    import ctypes as C

    lib = qa.load_library()
    lib.qa_profile_begin()  # only the prefill runs on conv_gemm (the decode steps are GEMV launches): 1 + 1 tokens keep every output non-empty
    lm.generate(task, mel1 if enr is not None else None, enr, mel1, mix, global_length=1, do_sample=False)
    torch.cuda.synchronize(dev)
    prof = (C.c_double * NPROF)()
    lib.qa_profile_end(prof, NPROF)
    g_flop = sum(prof[4 * i] for i in range(len(CFG_NAMES)))
    g_ms = sum(prof[4 * i + 1] for i in range(len(CFG_NAMES)))
    g_n = int(sum(prof[4 * i + 2] for i in range(len(CFG_NAMES))))
    prefill = {"ms": ms_prefill, "rows": batch * (2 + 250 + (1 + n_enroll if n_enroll else 0)),
               "roofline": {"bound": "mfma", "achieved": g_flop / (g_ms * 1e-3) / 1e12 if g_ms else None, "peak": MFMA_F32_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": g_flop / (g_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS if g_ms else None,
                            "gemm_launches": g_n, "gemm_ms": g_ms, "gemm_flop": g_flop,
                            "whole_prefill_tflops": g_flop / (ms_prefill * 1e-3) / 1e12 if ms_prefill > 0 else None,
                            "note": "all conv_gemm launches of the prefill (HIP events on the launch stream, qa_profile_begin/end); "
                                    "whole_prefill_tflops divides the same FLOPs by the prefill wall time (attention, norms, RoPE / cache append included)"}}
    # algorithmic HBM bytes of one decode step: 12 layers x (qkv 3 d^2 + o d^2 + gate/up 2 d I + down d I) fp32 weights + the active
    # output_head slice, and K + V of every cached position of every sequence (fp32); averaged over the 283 steps
    d, inter, layers, prompt = 512, 2048, 12, 2 + 250 + (1 + n_enroll if n_enroll else 0)
    w_body = layers * (4 * d * d + 3 * d * inter) * 4
    w_head = (33 * 4096 + 250 * 8192) / 283 * d * 4
    kv = sum(layers * batch * (prompt + s + 1) * d * 2 * 4 for s in range(283)) / 283
    step_bytes = w_body + w_head + kv
    ms_step = 1e3 * best / 283
    return {"metric": "UniSE AR tokens/sec (greedy generate, prefill included)", "value": world * batch * 283 / best,
            "unit": "tokens/sec", "ms_per_generate": 1e3 * best, "ms_per_decode_step_incl_prefill": ms_step,
            "ms_prefill": ms_prefill, "ms_per_decode_step": ms_decode, "prefill": prefill,
            "chains": max(1, -(-batch // 64)), "row_groups_per_launch": max(1, -(-min(batch, 64) // 32)),
            "roofline": {"bound": "hbm", "achieved": step_bytes / (ms_step * 1e-3) / 1e12, "peak": HBM_PEAK_TBS, "unit": "TB/s",
                         "frac": step_bytes / (ms_step * 1e-3) / 1e12 / HBM_PEAK_TBS,
                         "bytes_per_step": {"weights": w_body + w_head, "kv_cache_mean": kv},
                         "note": "whole decode step (62 dependent launches: 5 per layer + head + pick; up to 64 sequences ride in ONE chain of such steps), prefill time included in the "
                                 "denominator; the step is bound by the latency of its dependent launches (3.1 us floor each, 5-10 us measured), not by bytes (DESIGN.md section 11)"},
            "config": {"workload": f"LLM_SFT.generate {task.upper()} task, {batch} segments x 5 s per GPU, prompt {prompt}, 33 global + 250 semantic steps",
                       "dtype": "f32"}}


def bicodec_bench(dev, batch, reps=3):
    """Last stage of Model.test_step (model.py:193): BiCodec.detokenize of `batch` 5 s segments (250 semantic + 32 global tokens each),
    published Spark-TTS BiCodec shapes, seeded weights, tokens resident in HBM."""
    import unified_audio_amd as qa
    from unified_audio_amd import synth

    spec = synth.BiCodecShapes()
    sd = synth.bicodec_state_dict(77, spec)
    n_params = sum(v.numel() for v in sd.values())
    m = qa.BiCodec(device=dev).load_state_dict(sd)
    del sd
    sem, glob = synth.bicodec_tokens(78, batch, 250, spec)
    sem, glob = sem.to(dev), glob.to(dev)
    wav = m.detokenize(sem, glob)
    torch.cuda.synchronize(dev)
    assert torch.isfinite(wav).all()
    t0 = time.perf_counter()
    for _ in range(reps):
        wav = m.detokenize(sem, glob)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    # contraction FLOPs per 5 s segment: wave generator (ConvTranspose1d + 3 residual units of k7 + k1 per block) + prenet
    fl, ch, T = 0.0, 1536, 250
    fl += 2.0 * T * 1024 * 1536 * 7
    for k, s in zip((16, 11, 8, 4), (8, 5, 4, 2)):
        co = ch // 2
        fl += 2.0 * T * ch * co * k            # every input frame meets every tap once
        T *= s
        fl += 3 * 2.0 * T * co * co * 8        # k7 + k1
        ch = co
    fl += 2.0 * T * ch * 7
    fl += 250 * (2.0 * 1024 * 384 * 2 + 3 * 2.0 * 384 * 384 * 7 + 16 * 2.0 * 2 * 384 * 2048)
    return {"metric": "audio-seconds/sec BiCodec.detokenize", "value": batch * 5.0 / dt, "unit": "audio-seconds/sec", "ms_per_pass": 1e3 * dt,
            "tflops": batch * fl / dt / 1e12,
            "config": {"workload": f"BiCodec detokenizer ({n_params / 1e6:.0f} M parameters on the decode side, seeded random weights), {batch} segments x "
                                   f"250 semantic + 32 global tokens -> {batch} x 80 000 samples @16 kHz", "dtype": "f32"}}


def unise_pipeline_bench(dev, batches=6, seg_per_batch=16, lm_graph=False):
    """BASELINE configs[2] END TO END over consecutive batches: Model.test_step 'se' (WavLM -> LLM_SFT.generate -> BiCodec.detokenize) on
    `batches` x `seg_per_batch` independent 5 s segments through UniSE.enhance (one stage after the other, one micro-batch at a
    time) and through UniSE.enhance_pipelined (the three stages of consecutive micro-batches on three streams).  Published shapes of
    all three models, seeded weights, waveforms resident in HBM; the two drivers' outputs are compared bit for bit."""
    import unified_audio_amd as qa
    from unified_audio_amd import synth
    from unified_audio_amd import unise as U

    fx = qa.SSLFeatureExtractor(qa.SPEC_WAVLM_BASE_PLUS, device=dev).load_state_dict(_ssl_state_dict(qa.SPEC_WAVLM_BASE_PLUS))
    lm = qa.LLM_SFT(device=dev).load_state_dict(synth.lm_state_dict(4321))
    bic = qa.BiCodec(device=dev).load_state_dict(synth.bicodec_state_dict(77, synth.BiCodecShapes()))
    drv = U.UniSE(lm, fx, tokenizer=qa.BiCodecTokenizer(model=bic))
    n = batches * seg_per_batch
    srcs = [synth.synth_wav(300 + i, 1, U.SEG_LEN).to(dev) for i in range(n)]  # n one-segment utterances
    seq_t = pipe_t = float("inf")
    for it in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        want = []
        for b in range(batches):  # the sequential driver, one micro-batch per call (what end_to_end_b16 sums, measured in one piece)
            want += drv.enhance("se", srcs[b * seg_per_batch:(b + 1) * seg_per_batch])
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        got = drv.enhance_pipelined("se", srcs, segments_per_batch=seg_per_batch, lm_graph=lm_graph)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        if it:
            seq_t, pipe_t = min(seq_t, t1 - t0), min(pipe_t, t2 - t1)
    same = all(torch.equal(a, b) for a, b in zip(got, want))
    return {"value": n * 5.0 / pipe_t, "unit": "audio-seconds/sec", "ms_per_batch": 1e3 * pipe_t / batches,
            "sequential": {"value": n * 5.0 / seq_t, "ms_per_batch": 1e3 * seq_t / batches}, "batches": batches, "segments_per_batch": seg_per_batch,
            "bit_identical_to_sequential": bool(same),
            "config": {"workload": f"UniSE 'se' end to end, {batches} consecutive batches of {seg_per_batch} x 5 s segments, WavLM | AR-LM | BiCodec of "
                                   "consecutive batches on three streams (UniSE.enhance_pipelined)", "dtype": "f32"}}


def unise_micro_batch_bench(dev, n_seg=256, sizes=(64, 128, 256)):
    """throughput beyond the 64-segment micro-batch.  The SAME 256 one-segment utterances through UniSE.enhance ('se':
    WavLM -> LLM_SFT.generate -> BiCodec.detokenize) with max_segments = 64 / 128 / 256, i.e. 4 / 2 / 1 passes whose LM runs 1 / 2 / 4
    chains of 64 sequences (csrc/lm.cpp: chains replay one captured step per token on internal streams).  Results are identical for every
    value (each stage is batch-invariant); the default of unified_audio_amd.UniSE follows the fastest."""
    import unified_audio_amd as qa
    from unified_audio_amd import synth
    from unified_audio_amd import unise as U

    fx = qa.SSLFeatureExtractor(qa.SPEC_WAVLM_BASE_PLUS, device=dev).load_state_dict(_ssl_state_dict(qa.SPEC_WAVLM_BASE_PLUS))
    lm = qa.LLM_SFT(device=dev).load_state_dict(synth.lm_state_dict(4321))
    bic = qa.BiCodec(device=dev).load_state_dict(synth.bicodec_state_dict(77, synth.BiCodecShapes()))
    base = [synth.synth_wav(300 + i, 1, U.SEG_LEN).to(dev) for i in range(16)]
    srcs = [base[i % 16] * (1.0 - 0.002 * (i // 16)) for i in range(n_seg)]  # 256 distinct utterances from 16 generated ones
    out, ref = {}, None
    for m in sizes:
        drv = U.UniSE(lm, fx, tokenizer=qa.BiCodecTokenizer(model=bic), max_segments=m)
        best, lm_ms = float("inf"), None
        for it in range(2):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            est = drv.enhance("se", srcs)
            torch.cuda.synchronize(dev)
            best = min(best, time.perf_counter() - t0)
        chk = float(sum(float(e.double().abs().sum()) for e in est))
        ref = chk if ref is None else ref
        # the LM alone at this micro-batch (features precomputed): tokens/s of one generate call
        feats = fx(torch.cat(srcs[:m], dim=0))
        mel = torch.zeros(m, 250, 80)
        lm.generate("se", None, None, mel, feats, do_sample=False)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        lm.generate("se", None, None, mel, feats, do_sample=False)
        torch.cuda.synchronize(dev)
        lm_ms = 1e3 * (time.perf_counter() - t0)
        out[f"max_segments_{m}"] = {"value": n_seg * 5.0 / best, "unit": "audio-seconds/sec", "ms_per_256_segments": 1e3 * best,
                                    "lm_generate_ms": lm_ms, "lm_tokens_per_sec": m * 283 / (lm_ms * 1e-3), "lm_chains": max(1, -(-m // 64)),
                                    "same_output_as_first": bool(chk == ref)}
        del feats
    out["config"] = {"workload": f"UniSE 'se' end to end (WavLM | AR-LM | BiCodec), {n_seg} x 5 s segments per call of UniSE.enhance, micro-batch swept", "dtype": "f32"}
    return out


def rvq_bench(dev, lib, n_vec, Q, K=1024, D=512, reps=5):
    """The RVQ search by itself (qa_rvq_search = ResidualVQ.forward) at the shape of one stream of BASELINE configs[4]:
    n_vec residual vectors x Q stages against K x D codebooks (codebooks scaled 0.5^q per stage like the synthetic checkpoints).
    FLOPs = the distance products 2 K D per vector-stage; bytes = codebooks once + vectors in / residual update per stage."""
    from unified_audio_amd import _lib

    g = torch.Generator().manual_seed(2024)
    cb = torch.stack([torch.randn(K, D, generator=g) * 0.5 ** q for q in range(Q)]).to(dev)
    x = torch.randn(n_vec, D, generator=g).to(dev)
    idx = torch.empty(n_vec, Q, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run():
        _lib.check(lib.qa_rvq_search(x.data_ptr(), n_vec, cb.data_ptr(), Q, K, D, idx.data_ptr(), None, C.c_void_p(stream)))

    run()
    torch.cuda.synchronize(dev)
    assert int(idx.min()) >= 0 and int(idx.max()) < K
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    flop = 2.0 * n_vec * Q * K * D
    return {"metric": "RVQ search vectors/sec (qa_rvq_search)", "value": n_vec / dt, "unit": "vectors/sec", "ms_per_call": 1e3 * dt,
            "tflops": flop / dt / 1e12, "frac_of_f32_mfma_peak": flop / dt / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "config": {"workload": f"{n_vec} vectors x {Q} stages x {K} codes x {D} dims (one of the two code streams), vectors resident in HBM",
                       "dtype": "f32 distances, int64 indices"}}


def gemm_calibration(dev, lib):
    """CALIBRATION, never a product path (tools/gemm_vs_library.py): conv_gemm's LINEAR launches against the vendor library's fp32
    GEMM (torch.matmul -> hipBLASLt / Tensile assembly, TF32 off) on four shapes that carry the H-Codec FLOPs, back to back on resident operands.  It says how
    far the hand-written kernel is from the best fp32 GEMM known for this device, which the nominal matrix peak alone does not."""
    from unified_audio_amd import _lib

    torch.backends.cuda.matmul.allow_tf32 = False
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = {}

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3 / reps)
        return best

    for M, N, K in ((16000, 3072, 1024), (16000, 1024, 3072), (9056, 2048, 512), (9056, 512, 2048)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        y, y2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        a = _lib.qa_conv_args()
        a.x, a.w, a.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
        a.B, a.T_in, a.C_in, a.T_out, a.N = 1, M, K, M, N
        a.ldx, a.ldy, a.ldr, a.ldg = K, N, N, N
        a.ksize, a.stride = 1, 1
        t_own = timed(lambda: _lib.check(lib.qa_conv1d_cl(C.byref(a), C.c_void_p(stream))))
        t_lib = timed(lambda: torch.matmul(x, w.t(), out=y2))
        fl = 2.0 * M * N * K
        out[f"{M}x{N}x{K}"] = {"conv_gemm_tflops": fl / t_own / 1e12, "vendor_library_tflops": fl / t_lib / 1e12,
                               "max_abs_diff_over_max_abs": float((y - y2).abs().max() / y2.abs().max())}
        del x, w, y, y2
    out["note"] = "calibration only: the vendor library is not linked or called by the product; fp32, TF32 off; TFLOP/s = 2 M N K / best mean launch time"
    return out


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _free_port() -> int:
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n: int) -> None:
    """`python bench.py --gpus N` as a PLAIN command (no RANK in the environment): re-execute this very command line under
    torch.distributed.run with one rank per GPU - the form the driver uses for N > 1 - instead of refusing.  Never returns."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    log(f"--gpus {n} without a launcher: re-executing as `{' '.join(cmd[1:9])} bench.py ...`")
    os.execv(sys.executable, cmd)


def bootstrap(args):
    """Rank bootstrap: -> (rank, local_rank, world, device, torch.distributed or None).  One process per GPU; RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* from the environment (torch.distributed.run); backend nccl (= RCCL over xGMI), gloo in the self-test."""
    if "RANK" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    selftest = args.bootstrap_selftest
    n_dev = 0 if selftest or not torch.cuda.is_available() else torch.cuda.device_count()
    if not selftest and world == 1 and n_dev == 0:
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if not selftest and world > n_dev:
            # every rank of a one-node launch sees the same devices, so every rank takes this branch: rendezvous on gloo (proves the
            # launch + rendezvous path), say clearly what is missing, leave with a non-zero status - no rank is left waiting
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
            msg = (f"bench.py --gpus {world}: the {world} ranks met (rendezvous ok) but this node exposes only {n_dev} GPU(s); "
                   f"ranks {list(range(n_dev, world))} have no device of their own. Run on a node with >= {world} MI355X or lower --gpus.")
            if rank == 0:
                print(json.dumps({"error": msg, "n_gpus": world, "devices_visible": n_dev, "rendezvous": "ok"}), flush=True)
                log(msg)
            dist.destroy_process_group()
            raise SystemExit(3)
    if selftest:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if dist is not None:
        if selftest:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    return rank, local_rank, world, dev, dist


def collect_ranks_seen(dist, rank, local_rank, world, dev):
    """Which devices the ranks really sit on (all-gather of their UUIDs): "did RCCL see N distinct GPUs" is answerable from the line."""
    if dev.type == "cuda":
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "name": props.name, "uuid": str(getattr(props, "uuid", "")),
                "pci_bus_id": getattr(props, "pci_bus_id", None), "host": os.uname().nodename}
    else:  # the gloo self-test: a rank's "device" is its process
        mine = {"rank": rank, "local_rank": local_rank, "name": "cpu (bootstrap self-test)", "uuid": f"pid-{os.getpid()}", "pci_bus_id": None,
                "host": os.uname().nodename}
    seen = [None] * world
    dist.all_gather_object(seen, mine)
    return {"ranks": seen, "distinct_devices": len({(s["host"], s["uuid"], s["pci_bus_id"]) for s in seen})}


def all_ranks_ok(dist, err, what):
    """Agree on a failure BEFORE anybody enters the next collective: every rank contributes its error string (or None); if any rank failed,
    EVERY rank raises the same RuntimeError - so a leg that dies on one rank (out of memory while building a model, say) ends as an
    `{"error": ...}` object on all of them instead of leaving the others in a barrier."""
    errs = [None] * dist.get_world_size()
    dist.all_gather_object(errs, err)
    bad = [(r, e) for r, e in enumerate(errs) if e is not None]
    if bad:
        raise RuntimeError(f"{what}: " + "; ".join(f"rank {r}: {e}" for r, e in bad))


def _guard(fn):
    """-> (result, None) or (None, 'Type: message')"""
    try:
        return fn(), None
    except Exception as e:  # noqa: BLE001
        return None, f"{type(e).__name__}: {str(e)[:200]}"


def exchange_leg(dist, rank, world, dev, hot_path, make_inputs, n_inputs, pad_values, fence, scatter_bytes):
    """The exchange step north_star names - scatter of the clips from rank 0, the per-GPU hot path, gather of codes + waveforms back
    (unified_audio_amd.dist.run_sharded: packed point-to-point transfers, the code path the gloo tests exercise) - timed NEXT TO the hot
    path, never inside `value`.  -> (exchange dict, gathered results on rank 0)"""
    from unified_audio_amd import dist as qd

    inputs, err = _guard(make_inputs) if rank == 0 else ([None] * n_inputs, None)
    all_ranks_ok(dist, err, "building the inputs on rank 0")
    tm = {}
    fence()
    res = qd.run_sharded(hot_path, inputs, dev, pad_values=pad_values, timings=tm)
    tt = torch.tensor([tm["scatter_s"], tm["compute_s"], tm["gather_s"]], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    exchange = {"scatter_ms": 1e3 * float(tt[0]), "compute_ms": 1e3 * float(tt[1]), "gather_ms": 1e3 * float(tt[2]),
                "exchange_ms": 1e3 * float(tt[0] + tt[2]), "scatter_bytes": scatter_bytes,
                "note": "max over ranks; rank 0 sends every other rank exactly its block of clips + SSL features and receives its codes + "
                        "waveforms (grouped point-to-point send / recv over RCCL, no padding, no collective); reported beside the hot "
                        "path, never inside `value`"}
    return exchange, res


def sharded_config_leg(dist, rank, world, dev, setup, make_local, units_per_rank, unit, pad_values, fence, reps, workload, config_ref):
    """One multi-GPU BASELINE configuration at N > 1: every rank runs `hot_path` on ITS OWN share of the
    configuration (inputs resident, barrier + device fence on both sides, MAX over ranks) -> `value` = the units all ranks processed per
    second; beside it - never inside it - the exchange step of that configuration (rank 0 holds every rank's share, run_sharded scatters
    it, runs the same hot path, gathers the results; rank 0's block is checked against its own local run).  make_local(r) -> rank r's
    input tensors [n_r, ...] (seeded by r, so rank 0 can rebuild everybody's share for the scatter); setup() -> hot_path builds the model
    of the leg.  Every phase that can fail on ONE rank (model construction, warm-up, a timed pass) is followed by all_ranks_ok, so the
    ranks leave a failing leg together."""
    def prepare():
        hp = setup()
        loc = [t.to(dev) for t in make_local(rank)]
        hp(*loc)  # warm-up (workspace growth, graphs)
        return hp, loc

    got, err = _guard(prepare)
    all_ranks_ok(dist, err, "setting the leg up")
    hot_path, local = got
    best, own = float("inf"), None
    for _ in range(reps):
        fence()
        t0 = time.perf_counter()
        def timed_pass():
            out = hot_path(*local)
            if torch.device(dev).type == "cuda":
                torch.cuda.synchronize(dev)  # this rank's own work is complete: its clock stops here, the MAX over ranks below is the job's time
            return out

        own, err = _guard(timed_pass)
        dt = time.perf_counter() - t0  # the error agreement (pickling + a collective) stays outside the clock
        all_ranks_ok(dist, err, "a timed pass")
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        best = min(best, float(t.item()))
    leg = {"value": world * units_per_rank / best, "unit": unit, "ms_per_step": 1e3 * best, "n_gpus": world, "scaling": "weak",
           "timing": "barrier + device fence in front, every rank's clock stops at its own device synchronisation, max over ranks, best of %d" % reps,
           "config": {"workload": workload, "baseline_config": config_ref, "parallelism": f"dp{world} (independent items, no collective)", "dtype": "f32"}}
    try:
        def make_inputs():
            shares = [make_local(r) for r in range(world)]
            return [torch.cat([sh[i] for sh in shares]).to(dev) for i in range(len(shares[0]))]

        n_in = len(local)
        sbytes = world * sum(t.numel() * t.element_size() for t in local)
        exchange, res = exchange_leg(dist, rank, world, dev, hot_path, make_inputs, n_in, pad_values, fence, sbytes)
        if rank == 0:
            n0 = local[0].shape[0]
            exchange["gathered_shapes"] = [list(r_.shape) for r_ in res]
            exchange["rank0_block_matches_local_run"] = bool(all(
                torch.equal(g[:n0][tuple(slice(None) if i == 0 else slice(0, e) for i, e in enumerate(o.shape))], o) for g, o in zip(res, own)))
        leg["exchange"] = exchange
    except Exception as e:  # noqa: BLE001
        leg["exchange"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
    return leg


def bootstrap_selftest(args):
    """--bootstrap-selftest: everything of an N > 1 bench run that is NOT the codec - launch, rendezvous, ranks_seen, the exchange leg,
    the max-over-ranks reduction, one JSON line from rank 0 - on gloo with a stub hot path (a fixed affine map of the clips, so the
    gathered result is checkable).  No throughput is claimed: value is null."""
    rank, local_rank, world, dev, dist = bootstrap(args)
    B, T = 3, 64

    def clips_of(r):
        return torch.arange(B * T, dtype=torch.float32).reshape(B, T) + 1000.0 * r

    def stub(w, f):
        return (w[:, ::8] * 2).to(torch.int64), w * 0.5 + f[:, :1]

    def fence():
        if dist is not None:
            dist.barrier()

    line = {"metric": "bootstrap self-test (no throughput)", "value": None, "selftest": True, "n_gpus": world, "backend": "gloo",
            "launched_by": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "direct"}
    if dist is not None:
        ranks_seen = collect_ranks_seen(dist, rank, local_rank, world, dev)

        def make_inputs():
            return [torch.cat([clips_of(r) for r in range(world)]), torch.cat([clips_of(r)[:, :4] + 0.25 for r in range(world)])]

        exchange, res = exchange_leg(dist, rank, world, dev, stub, make_inputs, 2, None, fence, world * B * (T + 4) * 4)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        line["max_over_ranks"] = float(t.item())
        if rank == 0:
            want = stub(*make_inputs())
            exchange["gathered_shapes"] = [list(r_.shape) for r_ in res]
            exchange["gathered_equals_unsharded_run"] = bool(all(torch.equal(a, b) for a, b in zip(res, want)))
        line["ranks_seen"], line["exchange"] = ranks_seen, exchange
        # the two multi-GPU BASELINE configurations' legs (configs[3] TSE share, configs[4] H-Codec 2.0 share) through the SAME code path
        # bench.py uses at N > 1, with stub hot paths of the real signatures: (mix, enrollment) -> (global ids, semantic ids) and
        # (wav, features) -> (acoustic codes, semantic codes, waveform)
        def tse_stub(mix, enr):
            return (mix[:, :4, 0] * 8 + enr[:, :4, 0]).to(torch.int64), (mix[:, :, 1] * 4).to(torch.int64)

        def codec_stub(w, f):
            return (w[:, ::16] * 4).to(torch.int64)[:, None, :], (f[:, :, 0] * 4).to(torch.int64)[:, None, :], w * 0.5

        fail_rank = int(os.environ.get("QA_SELFTEST_FAIL_RANK", "-1"))  # tests: the H-Codec 2.0 leg's setup dies on this rank only

        def codec_setup():
            if rank == fail_rank:
                raise MemoryError("injected failure of one rank's model construction")
            return codec_stub

        line["configs3_tse"] = sharded_config_leg(
            dist, rank, world, dev, lambda: tse_stub, lambda r: [clips_of(r).reshape(B, 16, 4), clips_of(r).reshape(B, 16, 4) + 0.5], B * 6, "tokens/sec",
            None, fence, 1, "stub hot path (self-test)", "configs[3]")
        try:  # as main() does: a leg that fails - on ANY rank - becomes an error object on rank 0's line, and nobody hangs
            line["configs4_hcodec20"] = sharded_config_leg(
                dist, rank, world, dev, codec_setup, lambda r: [clips_of(r), clips_of(r).reshape(B, 16, 4)], B * T / 48000.0, "audio-seconds/sec",
                None, fence, 1, "stub hot path (self-test)", "configs[4]")
        except Exception as e:  # noqa: BLE001
            line["configs4_hcodec20"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.cpu_baseline_worker:
        parts = args.cpu_baseline_worker.split(",")
        if parts[0] == "lm":
            return _lm_cpu_worker(int(parts[1]), int(parts[2]), int(parts[3]), int(parts[4]))
        c, sec, reps, thr, mdl = parts
        return _cpu_baseline_worker(int(c), float(sec), int(reps), int(thr), mdl)
    if args.bootstrap_selftest:
        return bootstrap_selftest(args)
    rank, local_rank, world, dev, dist = bootstrap(args)

    import unified_audio_amd as qa
    from unified_audio_amd import _lib, synth  # synthetic weights / inputs: data generation only; nothing under oracle/ is
    #                                            imported outside the cpu_baseline worker

    lib = qa.load_library()
    global SR
    log(f"generating seeded H-Codec {args.model} weights ...")
    if args.model == "2.0":
        SR = 48000
        spec = synth.Shapes20()
        sd = synth.hcodec20_state_dict(1234, spec)
        codec = qa.Codec(None, None, None, spec=qa.SPEC_20, device=dev).load_state_dict(sd)
        hop_in, frame_hop = spec.hop, spec.frame_hop
    else:
        spec = qa.SPEC_15 if args.model == "1.5" else qa.SPEC_10
        sd = synth.hcodec10_state_dict(1234, spec)
        codec = qa.Codec(None, None, None, spec=spec, device=dev).load_state_dict(sd)
        hop_in, frame_hop = 320, spec.enc_hop
    adaptive = getattr(spec, "adaptive", False)
    n_params = sum(v.numel() for v in sd.values())

    B = args.batch
    T = int(round(args.seconds * SR / frame_hop)) * frame_hop
    # each rank draws its own shard of clips (weak scaling: B clips per GPU, no data-path collective)
    wav = (synth.synth_wav_fullband(7 + rank, B, T) if args.model == "2.0" else synth.synth_wav(7 + rank, B, T)).to(dev)
    feats = synth.synth_feat(9 + rank, B, T // hop_in, spec.sem_in).transpose(1, 2).contiguous().to(dev)  # [B, N50, C] as the SSL model emits
    groups = [0]

    def step():
        if adaptive:
            codes = codec.encode(wav.unsqueeze(1), feats.transpose(1, 2))
            groups[0] = codes["acoustic_codes"].shape[-1]
            return codec.decode(**codes)
        ac, sc = codec.encode(wav.unsqueeze(1), feats.transpose(1, 2))
        return codec.decode(ac, sc)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    log(f"setup done: B={B} T={T} world={world}")
    for _ in range(args.warmup):
        step()
    fence()
    log("warm-up done")
    _lib.check(lib.qa_profile_begin())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {elapsed:.3f} s for {args.steps} steps")
    prof = (C.c_double * NPROF)()
    _lib.check(lib.qa_profile_end(prof, NPROF))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()

    # N > 1: (a) which devices the ranks really sit on (all-gather of their UUIDs: "did RCCL see N distinct GPUs" is answerable from
    # the line), (b) the exchange step north_star names - scatter of the clips from rank 0, gather of codes + waveforms back over
    # RCCL (unified_audio_amd.dist.run_sharded, the code path the gloo tests exercise) - timed NEXT TO the hot path, never inside
    # `value`: in the timed region above every rank draws its own shard, like the reference's rank-strided file lists
    ranks_seen, exchange = None, None
    if dist is not None:
        ranks_seen = collect_ranks_seen(dist, rank, local_rank, world, dev)
        try:
            K = spec.codebook_size

            def make_inputs():
                return [torch.cat([(synth.synth_wav_fullband(7 + r, B, T) if args.model == "2.0" else synth.synth_wav(7 + r, B, T)) for r in range(world)]).to(dev),
                        torch.cat([synth.synth_feat(9 + r, B, T // hop_in, spec.sem_in).transpose(1, 2).contiguous() for r in range(world)]).to(dev)]

            def hot_path(w, f):
                if adaptive:
                    codes = codec.encode(w.unsqueeze(1), f.transpose(1, 2))
                    return codes["acoustic_codes"], codes["semantic_codes"], codec.decode(**codes)
                a_, s_ = codec.encode(w.unsqueeze(1), f.transpose(1, 2))
                return a_, s_, codec.decode(a_, s_)

            exchange, res = exchange_leg(dist, rank, world, dev, hot_path, make_inputs, 2, [-K, -K, 0.0] if adaptive else None, fence,
                                         world * B * (T + (T // hop_in) * spec.sem_in) * 4)
            if rank == 0:
                own = hot_path(wav, feats)  # rank 0's block of the gathered result must be what rank 0 computes from its own shard
                g0 = res[2][:B]
                exchange["gathered_shapes"] = [list(r_.shape) for r_ in res]
                exchange["rank0_block_matches_local_run"] = bool(torch.equal(g0[..., : own[2].shape[-1]], own[2]))
            del res
        except Exception as e:  # noqa: BLE001
            exchange = {"error": f"{type(e).__name__}: {str(e)[:300]}"}

    # the same kernels with the library's internal stream concurrency switched off (every launch alone on the device):
    # what a kernel achieves by itself, as opposed to while it shares the CUs with the other stream's kernels
    iso = (C.c_double * NPROF)()
    n_hbm = lib.qa_profile_hbm_kinds()
    hbm = (C.c_double * (3 * n_hbm))()
    _lib.check(lib.qa_set_serial(1))
    _lib.check(lib.qa_profile_begin_ex(3))  # + the byte-bound kernels, each with its algorithmic bytes (qa_profile_end_hbm)
    for _ in range(0 if args.lean else 2):
        step()
    torch.cuda.synchronize(dev)
    _lib.check(lib.qa_profile_end_hbm(hbm, 3 * n_hbm))
    _lib.check(lib.qa_profile_end(iso, NPROF))
    _lib.check(lib.qa_set_serial(0))
    hbm_rows = []
    for k in range(n_hbm):
        by, ms, n = hbm[3 * k], hbm[3 * k + 1], hbm[3 * k + 2]
        if n:
            row = {"kernel": lib.qa_profile_hbm_name(k).decode(), "launches_per_step": n / 2, "avg_us": 1e3 * ms / n,
                   "algorithmic_bytes_per_launch": by / n, "achieved_GBps": by / (ms * 1e-3) / 1e9, "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                   "ms_per_step": ms / 2}
            if row["kernel"].startswith("seanet_block") and args.model != "2.0":
                # the one kernel north_star calls HBM-bound is not, in fp32: conv0 (k7, 1 -> 32) + k3 32 -> 16 + 1x1 16 -> 32 + the 32 -> 32 shortcut
                # are 3 296 MACs per sample for 132 bytes moved: 50 FLOP/B against a ridge of 157.3 / 8 = 19.7 FLOP/B, so its floor is the
                # matrix pipe (0.21 ms per launch), not HBM (0.085 ms) - price it against BOTH
                flop = 2.0 * 3296.0 * B * T
                row.update({"flop_per_launch": flop, "tflops": flop / (1e-3 * ms / n) / 1e12, "frac_of_f32_mfma_peak": flop / (1e-3 * ms / n) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                            "arithmetic_intensity_flop_per_byte": flop / (by / n), "ridge_flop_per_byte": MFMA_F32_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBPS,
                            "bound": "mfma (arithmetic intensity above the fp32 ridge: even at the matrix peak this launch would move its bytes at "
                                     "0.40 of the HBM peak)"})
            hbm_rows.append(row)

    # secondary: the same step with host (pageable) tensors in and out, as HCodecTokenizer's __main__ moves them
    # (audio_tokenizer.py:79,84: wav.to(device) ... wav_rec.cpu()); never `value`
    wav_h, feats_h = wav.cpu(), feats.cpu()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(1 if args.lean else 2):
        w_d, f_d = wav_h.to(dev), feats_h.to(dev)
        if adaptive:
            codes = codec.encode(w_d.unsqueeze(1), f_d.transpose(1, 2))
            rec_h = codec.decode(**{k: v.cpu().to(dev) for k, v in codes.items()}).cpu()
        else:
            a_, s_ = codec.encode(w_d.unsqueeze(1), f_d.transpose(1, 2))
            rec_h = codec.decode(a_.cpu().to(dev), s_.cpu().to(dev)).cpu()
    pcie_elapsed = (time.perf_counter() - t1) / (1 if args.lean else 2)
    del rec_h

    lm_line = None
    if not args.no_lm and not args.lean:
        del out
        log("UniSE LM generate ...")
        lm_line = lm_bench(dev, rank, world, dist, args.lm_batch)

    # N > 1: the two BASELINE configurations that ARE multi-GPU - configs[3] (UniSE TSE with enrollment, 64 mixed segments over 8 GPUs = 8
    # per GPU; the enrollment travels with its segment, model.py:209-210) and configs[4] (H-Codec 2.0, 128 x 30 s over 8 GPUs = 16 per
    # GPU) - every rank its own share, max over ranks, each with its own exchange block
    multi = {}
    if dist is not None and not args.lean and not args.no_extras and not args.no_multi_configs:
        if not args.no_lm:
            try:
                log("configs[3]: TSE share, 8 segments per GPU ...")
                keep3 = {}

                def tse_setup():
                    lm3 = keep3["lm"] = qa.LLM_SFT(device=dev).load_state_dict(synth.lm_state_dict(4321))
                    mel3 = torch.zeros(1, 250, 80)

                    def tse_path(mix, enr):
                        n_ = mix.shape[0]
                        return lm3.generate("tse", mel3.expand(n_, -1, -1), enr, mel3.expand(n_, -1, -1), mix, do_sample=False)

                    return tse_path

                multi["configs3_tse"] = sharded_config_leg(
                    dist, rank, world, dev, tse_setup, lambda r: [synth.synth_feats(50 + r, 8, 250), synth.synth_feats(90 + r, 8, 250)], 8 * 283,
                    "tokens/sec", None, fence, 2, "LLM_SFT.generate TSE task (greedy, prefill included), 8 segments x 5 s per GPU, prompt 503 "
                    "(mixture 250 + enrollment 250 + 3), 33 global + 250 semantic steps; enrollment features scattered with their segment",
                    "configs[3]: UniSE TSE, batch 64 over 8 GPUs")
                keep3.clear()
            except Exception as e:  # noqa: BLE001
                multi["configs3_tse"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        try:
            log("configs[4]: H-Codec 2.0 share, 16 x 30 s per GPU ...")
            spec20m = synth.Shapes20()
            B20m, T20m = 16, 30 * 48000 // spec20m.frame_hop * spec20m.frame_hop
            keep4 = {}

            def h20_setup():
                codec20m = keep4["codec"] = qa.Codec(None, None, None, spec=qa.SPEC_20, device=dev).load_state_dict(synth.hcodec20_state_dict(1234, spec20m))

                def h20_path(w, f):
                    a_, s_ = codec20m.encode(w, f)
                    return a_, s_, codec20m.decode(a_, s_)

                return h20_path

            multi["configs4_hcodec20"] = sharded_config_leg(
                dist, rank, world, dev, h20_setup,
                lambda r: [synth.synth_wav_fullband(17 + r, B20m, T20m), synth.synth_feat(19 + r, B20m, T20m // spec20m.hop, 768)],
                B20m * T20m / 48000.0, "audio-seconds/sec", None, fence, 2,
                f"H-Codec 2.0 Codec.encode+Codec.decode (1.17 G parameters), 16 clips x {T20m / 48000:.0f} s @48 kHz per GPU, 16 + 16 codebooks, SSL features precomputed",
                "configs[4]: H-Codec 2.0, 128 x 30 s over 8 GPUs")
            keep4.clear()
        except Exception as e:  # noqa: BLE001
            multi["configs4_hcodec20"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}

    extras = {}
    if rank == 0 and world == 1 and not args.lean and not args.no_extras and args.model == "1.5":
        try:
            # (1) the headline at a second grouping point (G ~ N25/8 ... N25/2): the per-call threshold of
            #     Codec.encode (codec_adaptive.py:150-158) raised until about half the frames open a group
            log("second grouping point ...")
            lo_t, hi_t, best_t, best_g = spec.threshold, 1.0, None, None
            for _ in range(7):
                mid = 0.5 * (lo_t + hi_t)
                g_ = codec.encode(wav.unsqueeze(1), feats.transpose(1, 2), threshold=mid)["acoustic_codes"].shape[-1]
                if best_g is None or abs(g_ - 125) < abs(best_g - 125):
                    best_t, best_g = mid, g_
                lo_t, hi_t = (mid, hi_t) if g_ < 125 else (lo_t, mid)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            for _ in range(3):
                codes = codec.encode(wav.unsqueeze(1), feats.transpose(1, 2), threshold=best_t)
                codec.decode(**codes)
            torch.cuda.synchronize(dev)
            dt2 = (time.perf_counter() - t2) / 3
            extras["hcodec15_second_grouping_point"] = {"value": B * T / SR / dt2, "unit": "audio-seconds/sec", "ms_per_step": 1e3 * dt2,
                                                        "groups_per_clip": int(best_g), "threshold": best_t,
                                                        "note": "same batch, similarity threshold raised so that G ~ N25/2 = 125 (aggregator "
                                                                "sequences 250 + G, RVQ over 32 x G vectors)"}
        except Exception as e:  # noqa: BLE001
            extras["hcodec15_second_grouping_point"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
    if rank == 0 and world == 1 and not args.lean and not args.no_extras and not args.no_lm:
        try:
            log("UniSE LM TSE share of configs[3] (8 segments per GPU, prompt 503) ...")
            extras["unise_lm_tse_b8"] = lm_bench(dev, rank, world, None, 8, reps=1, task="tse", n_enroll=250)
        except Exception as e:  # noqa: BLE001
            extras["unise_lm_tse_b8"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        try:
            # BASELINE configs[3] on ONE GPU is 64 segments: one chain, two row groups of 32 per launch (csrc/lm.cpp, r05; above 64: chains)
            log("UniSE LM at 64 segments per GPU (one chain, two row groups per launch): SE and TSE ...")
            extras["unise_lm_b64"] = lm_bench(dev, rank, world, None, 64, reps=1)
        except Exception as e:  # noqa: BLE001
            extras["unise_lm_b64"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        try:  # its own try: a failure of the 64-segment front-end / detokenizer must not overwrite the LM result above
            # configs[2] end to end at 64 segments (the driver's default micro-batch until r06: UniSE(max_segments=64); now 128, see unise_micro_batch): the LM's decode step
            # costs about the same for 16 and for 64 sequences, so the three stages are timed at 64 segments x 5 s as well
            fe64, bd64 = ssl_bench(dev, "unise", 64, 5.0, reps=2), bicodec_bench(dev, 64, reps=2)
            st64 = {"wavlm_ms": fe64["ms_per_pass"], "lm_generate_ms": extras["unise_lm_b64"]["ms_per_generate"], "bicodec_detokenize_ms": bd64["ms_per_pass"]}
            tot64 = sum(st64.values()) * 1e-3
            extras["unise_end_to_end_b64"] = {"value": 64 * 5.0 / tot64, "unit": "audio-seconds/sec", "ms": 1e3 * tot64, "stages": st64,
                                              "tokens_per_sec": 64 * 283 / tot64,
                                              "note": "Model.test_step 'se' on 64 x 5 s segments (the micro-batch unified_audio_amd.UniSE feeds by default), stages timed "
                                                      "back to back; outputs do not depend on the micro-batch (every stage is batch-invariant)"}
        except Exception as e:  # noqa: BLE001
            extras["unise_end_to_end_b64"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        try:
            extras["unise_lm_tse_b64"] = lm_bench(dev, rank, world, None, 64, reps=1, task="tse", n_enroll=250)
        except Exception as e:  # noqa: BLE001
            extras["unise_lm_tse_b64"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        try:
            log("UniSE micro-batch sweep (256 segments end to end at max_segments 64 / 128 / 256) ...")
            extras["unise_micro_batch"] = unise_micro_batch_bench(dev)
        except Exception as e:  # noqa: BLE001
            extras["unise_micro_batch"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}

    if rank == 0 and world == 1 and not args.lean and not args.no_extras and args.model == "1.5":
        try:
            # (3) H-Codec 2.0, the per-GPU share of BASELINE configs[4]: 16 clips x 30 s @ 48 kHz (1.17 B parameters)
            log("H-Codec 2.0 16 x 30 s (per-GPU share of configs[4]) ...")
            spec20 = synth.Shapes20()
            sd20 = synth.hcodec20_state_dict(1234, spec20)
            n20 = sum(v.numel() for v in sd20.values())
            codec20 = qa.Codec(None, None, None, spec=qa.SPEC_20, device=dev).load_state_dict(sd20)
            del sd20
            B20, T20 = 16, 30 * 48000 // spec20.frame_hop * spec20.frame_hop
            wav20 = synth.synth_wav_fullband(17, B20, T20).to(dev)
            feats20 = synth.synth_feat(19, B20, T20 // spec20.hop, 768).to(dev)
            for i in range(4):
                if i == 1:
                    torch.cuda.synchronize(dev)
                    t3 = time.perf_counter()
                a20, s20 = codec20.encode(wav20, feats20)
                rec20 = codec20.decode(a20, s20)
            torch.cuda.synchronize(dev)
            dt3 = (time.perf_counter() - t3) / 3
            assert torch.isfinite(rec20).all()
            extras["hcodec20_16x30s"] = {"value": B20 * T20 / 48000 / dt3, "unit": "audio-seconds/sec", "ms_per_step": 1e3 * dt3,
                                         "config": {"workload": f"H-Codec 2.0 Codec.encode+Codec.decode ({n20 / 1e6:.0f} M parameters), 16 clips x "
                                                                f"{T20 / 48000:.0f} s @48 kHz, 16 + 16 codebooks, SSL features precomputed", "dtype": "f32"}}
            del codec20, wav20, feats20, rec20
        except Exception as e:  # noqa: BLE001
            extras["hcodec20_16x30s"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}

    if rank == 0 and world == 1 and not args.lean and not args.no_extras and args.model == "1.5":
        try:
            # (4) H-Codec 1.0 at the metric's own batch (32 x 10 s @16 kHz): the non-adaptive SEANet / ConvNeXt codec of configs[0], at size
            log("H-Codec 1.0 32 x 10 s ...")
            sd10 = synth.hcodec10_state_dict(1234, qa.SPEC_10)
            codec10 = qa.Codec(None, None, None, spec=qa.SPEC_10, device=dev).load_state_dict(sd10)
            n10 = sum(v.numel() for v in sd10.values())
            del sd10
            feats10 = synth.synth_feat(9 + rank, B, T // hop_in, qa.SPEC_10.sem_in).to(dev)
            for i in range(6):
                if i == 1:
                    torch.cuda.synchronize(dev)
                    t4 = time.perf_counter()
                a10, s10 = codec10.encode(wav.unsqueeze(1), feats10)
                rec10 = codec10.decode(a10, s10)
            torch.cuda.synchronize(dev)
            dt4 = (time.perf_counter() - t4) / 5
            assert torch.isfinite(rec10).all()
            extras["hcodec10_32x10s"] = {"value": B * T / SR / dt4, "unit": "audio-seconds/sec", "ms_per_step": 1e3 * dt4,
                                         "config": {"workload": f"H-Codec 1.0 Codec.encode+Codec.decode ({n10 / 1e6:.0f} M parameters), {B} clips x "
                                                                f"{T / SR:.0f} s @16 kHz, 4 + 4 codebooks, SSL features precomputed", "dtype": "f32"}}
            del codec10, feats10, rec10
        except Exception as e:  # noqa: BLE001
            extras["hcodec10_32x10s"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}

    if rank == 0 and world == 1 and not args.lean and not args.no_extras:
        try:
            # (5) the RVQ search alone at configs[4]'s shape: 16 stages x 1024 x 512; 6 000 vectors = the per-GPU share
            #     (16 clips x 375 frames), 48 000 = the whole configuration's 128 clips on one GPU
            log("RVQ search at the configs[4] shape ...")
            extras["rvq_search_6000x16"] = rvq_bench(dev, lib, 6000, 16)
            extras["rvq_search_48000x16"] = rvq_bench(dev, lib, 48000, 16, reps=3)
        except Exception as e:  # noqa: BLE001
            extras["rvq_search_6000x16"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        try:
            log("GEMM calibration against the vendor library ...")
            extras["gemm_calibration"] = gemm_calibration(dev, lib)
        except Exception as e:  # noqa: BLE001
            extras["gemm_calibration"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}

    ssl_line = None
    if world == 1 and not args.lean and not args.no_ssl and args.model != "2.0":
        log("SSL front-end ...")
        try:
            ssl_line = ssl_bench(dev, args.model, B, args.seconds)
            if lm_line is not None:  # UniSE's own front-end (Model.extract_semantic_features): WavLM on the LM's 16 x 5 s segments
                lm_line["ssl_frontend"] = ssl_bench(dev, "unise", args.lm_batch, 5.0)
                lm_line["bicodec_detokenize"] = bicodec_bench(dev, args.lm_batch)
        except Exception as e:  # secondary: never take the headline line down with it
            ssl_line = {"error": f"{type(e).__name__}: {str(e)[:300]}"}

    if rank == 0:
        audio_s = world * B * T / SR * args.steps
        cfgs = []
        for i, name in enumerate(CFG_NAMES):
            fl, ms, n, by = prof[4 * i], prof[4 * i + 1], prof[4 * i + 2], prof[4 * i + 3]
            if n:
                cfgs.append({"kernel": name, "launches_per_step": n / args.steps, "avg_us": 1e3 * ms / n,
                             "tflops": fl / (ms * 1e-3) / 1e12, "share_of_step_time": ms * 1e-3 / elapsed,
                             "algorithmic_bytes_per_launch": by / n, "flop_per_launch": fl / n})
        dom = max(cfgs, key=lambda c: c["share_of_step_time"])
        # HBM bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950
        # correction, + WRITE_SIZE) over this same command, summarised under profiles/ (PMC cannot be sampled from inside the run)
        traffic, traffic_src, mfma_busy = None, None, None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic_hcodec15.json")
        if args.model == "1.5" and B == 32 and abs(args.seconds - 10.0) < 1e-6 and os.path.exists(pmc_file):
            import hashlib

            pmc = json.load(open(pmc_file))
            src_hash = hashlib.sha256(open(os.path.join(ROOT, "unified_audio_amd", "csrc", "conv_gemm.hip"), "rb").read()).hexdigest()[:16]
            want = dom["kernel"].replace("conv_gemm_kernel<", "").rstrip(">").replace(",", ", ")
            if pmc.get("_kernel_source_sha256_16") != src_hash:  # counters of another build of the kernel: do not quote them
                traffic_src = (f"stale: {pmc.get('_source', 'profiles/pmc_traffic_hcodec15.json')} was collected on conv_gemm.hip "
                               f"{pmc.get('_kernel_source_sha256_16')}, this build is {src_hash}")
            else:
                for k, v in pmc.items():
                    if isinstance(v, dict) and f"conv_gemm_kernel<{want}> (all" in k:
                        traffic = v["hbm_bytes_per_launch"]
                        traffic_src = pmc.get("_source", "profiles/") + " (2*FETCH_SIZE + WRITE_SIZE, per launch; rocprofv3 --pmc passes of this command)"
                        mfma_busy = v["mfma_busy_frac"]
        gemm_ms = sum(prof[4 * i + 1] for i in range(len(CFG_NAMES)))
        gemm_flop = sum(prof[4 * i] for i in range(len(CFG_NAMES)))
        line = {
            "metric": "audio-seconds/sec H-Codec encode+decode @16kHz b=32",
            "value": audio_s / elapsed,
            "unit": "audio-seconds/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": f"synthetic (seeded band-limited noise + tones; seeded random weights of the H-Codec {args.model} architecture)",
            "config": {"workload": f"H-Codec {args.model} Codec.encode+Codec.decode ({n_params / 1e6:.0f} M parameters"
                                   + (f", 32-layer aggregators + bottleneck, threshold {spec.threshold}, {groups[0]} groups per clip" if adaptive else "")
                                   + f"), {B} clips x {T / SR:.0f} s @{SR // 1000} kHz per GPU, SSL features precomputed, inputs resident in HBM",
                       "clips_per_gpu": B, "clip_seconds": T / SR, "parallelism": f"dp{world} (independent clips, no collective)"},
            "roofline": {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"], "peak": MFMA_F32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": dom["tflops"] / MFMA_F32_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                         "traffic_algorithmic": dom["algorithmic_bytes_per_launch"], "flop_per_launch": dom["flop_per_launch"],
                         "mfma_busy_frac_pmc": mfma_busy,
                         "avg_launch_us": dom["avg_us"], "launches_per_step": dom["launches_per_step"],
                         "gemm_launch_time_over_step_time_overlapping_streams": gemm_ms * 1e-3 / elapsed,
                         "whole_step": {"tflops": gemm_flop / elapsed / 1e12, "frac": gemm_flop / elapsed / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                        "gemm_tflop_per_step": gemm_flop / args.steps / 1e12,
                                        "note": "ALL conv_gemm FLOPs of the timed region / its wall time (recurrences, attention, norms, host gaps in the "
                                                "denominator): the whole step against the nominal fp32 matrix peak"},
                         "all_gemm_configs": cfgs,
                         "note": "live HIP events in the timed region; kernels on the library's concurrent internal streams share the "
                                 "CUs, so their durations overlap (shares - and gemm_launch_time_over_step_time_overlapping_streams, the SUM of launch durations over wall time - can pass 1). `isolated` = the same kernel with "
                                 "qa_set_serial(1), alone on the device (what rocprofv3 under QA_SERIAL=1 reports)"},
        }
        if hbm_rows:
            # the byte-bound side of the path (north_star: "rocprof HBM GB/s"): every elementwise / norm / gather kernel alone on the device
            # (qa_set_serial), ALGORITHMIC bytes (each input and output element once) / HIP-event duration against the 8 TB/s HBM3E peak.
            # The entry's headline is the kernel with the most time per step among them; `all` lists every kind.
            top = max(hbm_rows, key=lambda r_: r_["ms_per_step"])
            line["roofline_hbm"] = {"bound": "hbm", "kernel": top["kernel"], "achieved": top["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                    "frac": top["frac"], "traffic": None, "avg_launch_us": top["avg_us"], "launches_per_step": top["launches_per_step"],
                                    "share_of_step_time": top["ms_per_step"] / (1e3 * elapsed / args.steps),
                                    "all": sorted(hbm_rows, key=lambda r_: -r_["ms_per_step"]),
                                    "byte_bound_ms_per_step": sum(r_["ms_per_step"] for r_ in hbm_rows),
                                    "note": "isolated pass (qa_set_serial(1)); bytes are algorithmic, so a two-pass kernel (GroupNorm) or one that is matrix-pipe-bound in fp32 "
                                            "(the fused SEANet front: three small contractions per sample, 50 FLOP/B, priced against both rooflines in its row) shows as a low fraction"}
        di = CFG_NAMES.index(dom["kernel"])
        if iso[4 * di + 2]:
            tf = iso[4 * di] / (iso[4 * di + 1] * 1e-3) / 1e12
            line["roofline"]["isolated"] = {"achieved": tf, "frac": tf / MFMA_F32_PEAK_TFLOPS,
                                            "avg_launch_us": 1e3 * iso[4 * di + 1] / iso[4 * di + 2]}
        # the library's run-time switches as this process ran them (csrc/knobs.h): values that differ from the defaults come from the environment
        kn = _lib.knobs()
        line["knobs"] = {"non_default": {k: v[0] for k, v in kn.items() if v[0] != v[1]},
                         "recurrence": {k: kn[k][0] for k in ("QA_LSTM_XCD", "QA_LSTM_PERSISTENT", "QA_LSTM_TEAM") if k in kn},
                         "note": "QA_LSTM_XCD = 1: the d = 512 / 768 LSTMs run as the XCD-local in-launch recurrence (DESIGN.md section 3)"}
        line["pcie_inclusive"] = {"value": B * T / SR / pcie_elapsed, "unit": "audio-seconds/sec", "ms_per_step": 1e3 * pcie_elapsed,
                                  "note": "rank-0 only: wav+features H2D from pageable memory, codes D2H+H2D, waveform D2H included"}
        if lm_line is not None:
            if world == 1 and not args.no_cpu_baseline and not args.no_extras:
                log("LM cpu baseline ...")
                lm_line["cpu_baseline"] = lm_cpu_baseline()
            fe = lm_line.get("ssl_frontend")
            bd = lm_line.get("bicodec_detokenize")
            if fe and "ms_per_pass" in fe:  # BASELINE configs[2]: WavLM features -> AR-LM token generation -> codec decode, B = 16
                parts = {"wavlm_ms": fe["ms_per_pass"], "lm_generate_ms": lm_line["ms_per_generate"],
                         "bicodec_detokenize_ms": bd["ms_per_pass"] if bd and "ms_per_pass" in bd else None}
                tot = sum(v for v in parts.values() if v is not None) * 1e-3
                lm_line["end_to_end_b16"] = {"value": args.lm_batch * 5.0 / tot, "unit": "audio-seconds/sec", "ms": 1e3 * tot, "stages": parts,
                                             "tokens_per_sec": args.lm_batch * 283 / tot,
                                             "note": "Model.test_step 'se' on 16 x 5 s segments: WavLM front-end + LLM_SFT.generate + BiCodec.detokenize "
                                                     "(the codec decode the reference's test.py uses, model.py:193), stages timed back to back"}
            if world == 1 and not args.no_extras and not args.lean:
                try:
                    log("UniSE end to end, pipelined over 6 batches of 16 segments ...")
                    lm_line["end_to_end_pipelined_b16"] = unise_pipeline_bench(dev, 6, args.lm_batch)
                except Exception as e:  # noqa: BLE001
                    lm_line["end_to_end_pipelined_b16"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            line["unise_lm"] = lm_line
        if ranks_seen is not None:
            line["ranks_seen"] = ranks_seen
        if exchange is not None:
            line["exchange"] = exchange
        for k_, v_ in multi.items():  # configs3_tse / configs4_hcodec20 (N > 1 only), beside `value`
            line[k_] = v_
        if extras:
            line["extras"] = extras
        if ssl_line is not None:
            line["ssl_frontend"] = ssl_line
        if args.model == "2.0":
            line["metric"] = "audio-seconds/sec H-Codec 2.0 encode+decode @48kHz (BASELINE configs[4], per-GPU share)"
        if not args.no_cpu_baseline and not args.lean and args.model != "2.0":
            # rank 0 only, AFTER every timed region (at N > 1 the other ranks have finished their GPU work and are leaving)
            log("cpu baseline ...")
            line["cpu_baseline"] = cpu_baseline(args.cpu_clips, args.seconds, model=args.model)
            if line["cpu_baseline"]["value"]:
                line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
