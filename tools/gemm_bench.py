#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM kernel on the shapes H-Codec 1.0 (B=32 x 10 s) actually launches.
usage: [QA_GEMM_CFG=0|1|2] [QA_GEMM_XCD=0|1] python tools/gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import conv1d_cl  # noqa: E402
from unified_audio_amd import load_library  # noqa: E402

SHAPES = [  # name, M(rows), N, Cin, k, stride
    ("small.rvq_dist", 1056, 1024, 512, 1, 1), ("small.lm_qkv", 4032, 1536, 512, 1, 1), ("small.lm_o", 4032, 512, 512, 1, 1),
    ("small.lm_gateup", 4032, 4096, 512, 1, 1), ("small.lm_down", 4032, 512, 2048, 1, 1), ("small.agg_2k", 2112, 512, 512, 1, 1),
    ("small.ssl_ffn", 3984, 3072, 768, 1, 1), ("small.bicodec", 4000, 384, 2048, 1, 1),
    ("cal.4096^3", 4096, 4096, 4096, 1, 1), ("cal.8192x4096x4096", 8192, 4096, 4096, 1, 1), ("cal.16384x1024x2048", 16384, 1024, 2048, 1, 1),
    ("mimi.in_proj", 9056, 1536, 512, 1, 1), ("mimi.out_proj", 9056, 512, 512, 1, 1), ("mimi.lin1", 9056, 2048, 512, 1, 1),
    ("mimi.lin2", 9056, 512, 2048, 1, 1), ("bt.in_proj", 8000, 3072, 1024, 1, 1), ("bt.lin1", 8000, 2048, 1024, 1, 1),
    ("bt.lin2", 8000, 1024, 2048, 1, 1),
    ("agg15.qkv", 9056, 1536, 512, 1, 1), ("agg15.half", 4528, 512, 512, 1, 1), ("wavlm.ffn1", 4000, 3072, 768, 1, 1), ("wavlm.ffn2", 4000, 768, 3072, 1, 1),
    ("h20.pw1", 24000, 4608, 1536, 1, 1), ("h20.pw2", 24000, 1536, 4608, 1, 1),
    ("convnext.pw1", 16000, 2304, 768, 1, 1), ("convnext.pw2", 16000, 768, 2304, 1, 1),
    ("dec.k3", 16000, 768, 768, 3, 1), ("dec.lstm_ih", 16000, 3072, 768, 1, 1), ("dec.qkv", 16000, 2304, 768, 1, 1),
    ("dec.w2", 16000, 768, 3072, 1, 1), ("enc.lstm_ih", 16000, 2048, 512, 1, 1), ("enc.o", 16000, 512, 512, 1, 1),
    ("sem.k3", 16000, 768, 768, 3, 1), ("istft.basis", 16000, 1280, 1312, 1, 1), ("enc.down3", 16000, 512, 256, 16, 8),
    ("enc.down2", 128000, 256, 128, 10, 5), ("enc.down1", 640000, 128, 64, 8, 4), ("enc.down0", 2560000, 64, 32, 4, 2),
    ("enc.res0.k3", 5120000, 32, 32, 3, 1), ("enc.res0.sc", 5120000, 32, 32, 1, 1), ("enc.res1.pw", 2560000, 64, 32, 1, 1),
]


def bench_shape(lib, dev, M, N, C, k, s, reps=5):
    T = M * s  # one batch item; zero padding, enough frames for M outputs
    g = torch.Generator(device=dev).manual_seed(M + N + C)  # the same operands for every configuration of a sweep
    # ksize-1 / stride-1 layers get exactly M frames (T_in == T_out): that is what makes a launch LINEAR (conv_gemm.hip), as every nn.Linear of the
    # model graphs is - with T + k frames the r06 256 x 128 tile silently fell back to 128 x 128 (first session of profiles/r06_gemm_256_ab.txt)
    x = torch.randn(1, T if (k == 1 and s == 1) else T + k, C, device=dev, generator=g)
    w = torch.randn(N, k, C, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    for _ in range(2):
        y = conv1d_cl(lib, x, w, b, stride=s, T_out=M)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        conv1d_cl(lib, x, w, b, stride=s, T_out=M)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, y


def main():
    lib = load_library()
    dev = torch.device("cuda:0")
    only = os.environ.get("QA_BENCH_ONLY")
    shapes = [sh for sh in SHAPES if not only or any(sh[0].startswith(o) for o in only.split(","))]
    sweep = os.environ.get("QA_BENCH_CFGS")  # e.g. "-1,1,2,3,4": one column per forced tile configuration (qa_set_knob), -1 = cost model
    if sweep:
        cfgs = [int(c) for c in sweep.split(",")]
        print("TFLOP/s per forced QA_GEMM_CFG (-1 = the cost model's choice); `same` = outputs bit-identical to the first column's")
        print(f"{'shape':16s} {'M':>8s} {'N':>5s} {'K':>5s} " + " ".join(f"{'cfg' + str(c):>9s}" for c in cfgs) + "  same  best")
        for name, M, N, C, k, s in shapes:
            cols, ref, same = [], None, True
            for c in cfgs:
                if c == 0 and N > 64:  # 128x32 on wide layers: never chosen, slow to run
                    cols.append(float("nan"))
                    continue
                lib.qa_set_knob(b"QA_GEMM_CFG", c)
                ms, y = bench_shape(lib, dev, M, N, C, k, s, reps=3)
                cols.append(2.0 * M * N * C * k / ms / 1e9)
                if ref is None:
                    ref = y.clone()
                else:
                    same = same and bool(torch.equal(ref, y))
            lib.qa_set_knob(b"QA_GEMM_CFG", -1)
            best = max(range(len(cfgs)), key=lambda i: -1.0 if cols[i] != cols[i] else cols[i])
            print(f"{name:16s} {M:8d} {N:5d} {C * k:5d} " + " ".join(f"{v:9.1f}" for v in cols) + f"  {str(same):5s} cfg{cfgs[best]}", flush=True)
        return
    tot_t = tot_f = 0.0
    for name, M, N, C, k, s in shapes:
        ms, _ = bench_shape(lib, dev, M, N, C, k, s)
        fl = 2.0 * M * N * C * k
        tot_t += ms
        tot_f += fl
        print(f"{name:14s} M={M:8d} N={N:5d} K={C * k:5d}  {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
    print(f"TOTAL {tot_t:.3f} ms  {tot_f / tot_t / 1e9:.1f} TFLOP/s  cfg={os.environ.get('QA_GEMM_CFG', 'auto')} xcd={os.environ.get('QA_GEMM_XCD', '1')}")


if __name__ == "__main__":
    main()
