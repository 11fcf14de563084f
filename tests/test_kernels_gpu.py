"""Kernel-level parity through the C-ABI (qa_conv1d_cl, qa_rvq_search, qa_rvq_lookup) against CPU restatements."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hcodec_ref as R
from oracle import rvq_c
from tests.util import act_ref, conv1d_cl, dwconv_cl, rel_err, rownorm

pytestmark = pytest.mark.gpu

CONV_CASES = [
    # B, T, Cin, N, k, stride, mode, prologue, act, post, gamma, res, gate
    dict(B=1, T=300, Cin=64, N=70, k=1),                                     # plain linear, ragged M and N
    dict(B=2, T=50, Cin=64, N=96, k=3, mode="zero"),                          # "same" conv
    dict(B=3, T=37, Cin=32, N=64, k=8, stride=4, mode="reflect"),             # SConv1d with extra right padding
    dict(B=2, T=2, Cin=32, N=32, k=16, stride=8, mode="reflect"),             # input shorter than the reflect pad
    dict(B=2, T=5, Cin=32, N=32, k=10, stride=5, mode="reflect"),             # odd stride: asymmetric padding
    dict(B=2, T=40, Cin=32, N=16, k=3, mode="reflect", prologue=1, act=1),    # SEANet residual branch, N < tile
    dict(B=2, T=64, Cin=32, N=32, k=1, res=True, post=1),                     # shortcut + block, ELU after the add
    dict(B=1, T=200, Cin=128, N=384, k=1, act=2),                             # GELU epilogue
    dict(B=1, T=200, Cin=384, N=128, k=1, gamma=True, res=True),              # ConvNeXt pwconv2
    dict(B=1, T=130, Cin=128, N=256, k=1, gate=True),                         # SwiGLU product
    dict(B=2, T=100, Cin=64, N=64, k=4, stride=2, mode="zero1"),              # semantic strided conv (pad 1,1)
    dict(B=4, T=250, Cin=256, N=300, k=3, mode="zero"),                       # 128x128 tiles, ragged N
    dict(B=1, T=1, Cin=32, N=32, k=1),                                        # single row
    dict(B=2, T=330, Cin=64, N=160, k=16, stride=8, mode="reflect"),          # ksize > 8: source-frame table rebuilt mid-tile (128x64/128 tiles)
    dict(B=3, T=97, Cin=32, N=136, k=9, stride=4, mode="reflect", res=True),  # ragged M, N % 8 == 0 only, residual, two table windows
    dict(B=1, T=300, Cin=96, N=66, k=3, mode="zero", gate=True, act=3),       # N % 4 != 0: scalar epilogue path
    dict(B=2, T=129, Cin=512, N=512, k=1, gamma=True, res=True, post=1),      # K = 512 (BK = 16 variant), every epilogue stage but the gate
]


def _ref_conv(x, w, bias, case):
    """x [B,T,C], w [N,k,C] -> [B,T_out,N] with torch ops on CPU (the reference's own op sequence)."""
    xc = x.transpose(1, 2)  # [B,C,T]
    wc = w.permute(0, 2, 1).contiguous()  # [N,C,k]
    k, stride = case["k"], case.get("stride", 1)
    if case.get("prologue"):
        xc = F.elu(xc)
    mode = case.get("mode")
    if mode == "reflect":
        y = R.sconv1d(xc, wc, bias, stride)
    elif mode == "zero":
        y = F.conv1d(xc, wc, bias, stride=stride, padding=(k - 1) // 2)
    elif mode == "zero1":
        y = F.conv1d(xc, wc, bias, stride=stride, padding=1)
    else:
        y = F.conv1d(xc, wc, bias, stride=stride)
    return y.transpose(1, 2)


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(i) for i in range(len(CONV_CASES))])
def test_conv1d_cl_matches_torch(qa_lib, gpu_device, case):
    g = torch.Generator().manual_seed(17)
    B, T, Cin, N, k = case["B"], case["T"], case["Cin"], case["N"], case["k"]
    stride = case.get("stride", 1)
    x = torch.randn(B, T, Cin, generator=g)
    w = torch.randn(N, k, Cin, generator=g) / (k * Cin) ** 0.5
    bias = torch.randn(N, generator=g)
    ref = _ref_conv(x, w, bias, case)
    T_out = ref.shape[1]
    gamma = torch.rand(N, generator=g) + 0.5 if case.get("gamma") else None
    res = torch.randn(B, T_out, N, generator=g) if case.get("res") else None
    gate = torch.randn(B, T_out, N, generator=g) if case.get("gate") else None
    if gate is not None:
        ref = F.silu(gate) * ref
    ref = act_ref(ref, case.get("act", 0))
    if gamma is not None:
        ref = ref * gamma
    if res is not None:
        ref = ref + res
    ref = act_ref(ref, case.get("post", 0))

    mode = case.get("mode")
    if mode == "reflect":
        pt = k - stride
        right = pt // 2
        left = pt - right
        extra = T_out * stride - T
        pad, pm = (left, right + extra), 1
    elif mode == "zero":
        pad, pm = ((k - 1) // 2, (k - 1) // 2), 0
    elif mode == "zero1":
        pad, pm = (1, 1), 0
    else:
        pad, pm = (0, 0), 0
    dev = gpu_device
    y = conv1d_cl(qa_lib, x.to(dev), w.to(dev), bias.to(dev), stride=stride, pad=pad, pad_mode=pm,
                  prologue=case.get("prologue", 0), act=case.get("act", 0), post_act=case.get("post", 0),
                  gamma=None if gamma is None else gamma.to(dev), residual=None if res is None else res.to(dev),
                  gate=None if gate is None else gate.to(dev), T_out=T_out)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    err = rel_err(y, ref)
    assert err < 2e-6, f"rel err {err}"  # fp32 tolerance: only the summation order differs


@pytest.mark.parametrize("case", [0, 1, 5, 7, 8, 9, 11, 13, 14, 16])
def test_conv_gemm_tile_configurations_are_bit_identical(qa_lib, gpu_device, knob, case):
    """Every tile configuration of conv_gemm (QA_GEMM_CFG 1 .. 4: 128x64, 128x128 and the 64-row tiles 64x128 / 64x64 of round 4)
    accumulates an output element over k in the same order, so the choice of tile - which the cost model makes from M, i.e. from
    the batch size - never changes a bit: required for `a clip's result does not depend on the batch it rides in`."""
    c = CONV_CASES[case]
    g = torch.Generator().manual_seed(23)
    B, T, Cin, N, k = c["B"], c["T"], c["Cin"], c["N"], c["k"]
    stride = c.get("stride", 1)
    x = torch.randn(B, T, Cin, generator=g).to(gpu_device)
    w = (torch.randn(N, k, Cin, generator=g) / (k * Cin) ** 0.5).to(gpu_device)
    bias = torch.randn(N, generator=g).to(gpu_device)
    T_out = (T - k) // stride + 1  # no padding: geometry is covered by test_conv1d_cl_matches_torch
    gamma = (torch.rand(N, generator=g) + 0.5).to(gpu_device) if c.get("gamma") else None
    res = torch.randn(B, T_out, N, generator=g).to(gpu_device) if c.get("res") else None
    gate = torch.randn(B, T_out, N, generator=g).to(gpu_device) if c.get("gate") else None
    outs = {}
    for cfg in (-1, 1, 2, 3, 4):
        knob("QA_GEMM_CFG", cfg)
        outs[cfg] = conv1d_cl(qa_lib, x, w, bias, stride=stride, prologue=c.get("prologue", 0), act=c.get("act", 0), post_act=c.get("post", 0),
                              gamma=gamma, residual=res, gate=gate, T_out=T_out).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(outs[-1]).all()
    for cfg in (1, 2, 3, 4):
        assert torch.equal(outs[cfg], outs[-1]), f"QA_GEMM_CFG={cfg} differs from the cost model's choice"


@pytest.mark.gpu
@pytest.mark.parametrize("mode,C,bias", [(1, 512, False), (2, 512, True), (2, 768, True), (1, 1024, False), (2, 1024, False), (2, 256, True), (1, 1536, False)])
def test_rownorm_matches_torch(qa_lib, gpu_device, mode, C, bias):
    """qa_rownorm against torch's fp32 LayerNorm (nn.LayerNorm: biased variance) / the RMSNorm of transformer.py:77-96 (x * rsqrt(mean(x^2) + eps) * w),
    rows with a large common offset included (the centred second pass of the LayerNorm form must not lose it)."""
    g = torch.Generator().manual_seed(100 + C + mode)
    rows = 1003
    x = torch.randn(rows, C, generator=g) * (torch.rand(rows, 1, generator=g) * 3 + 0.1) + torch.randn(rows, 1, generator=g) * 5
    w = torch.rand(C, generator=g) + 0.5
    b = torch.randn(C, generator=g) if bias else None
    eps = 1e-5 if mode == 2 else 1e-6
    y = rownorm(qa_lib, x.to(gpu_device), w.to(gpu_device), b.to(gpu_device) if bias else None, eps=eps, mode=mode).cpu()
    xd = x.double()
    ref = (F.layer_norm(xd, (C,), w.double(), b.double() if bias else None, eps) if mode == 2
           else xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + eps) * w.double())
    assert rel_err(y, ref) < 5e-7
    assert (y - ref.float()).abs().max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,C,k,ln,pad", [(2, 500, 1024, 7, True, -1), (3, 77, 768, 7, True, -1), (1, 5, 512, 7, True, -1), (2, 131, 256, 5, False, -1),
                                              (2, 64, 1024, 5, True, 4), (4, 1500, 1024, 7, True, 6), (2, 33, 384, 7, True, -1), (1, 9, 1024, 3, True, -1)])
def test_dwconv_matches_torch_and_strip_is_bit_identical_to_the_row_kernel(qa_lib, gpu_device, knob, B, T, C, k, ln, pad):
    """qa_dwconv_cl (depthwise Conv1d with zero padding + LayerNorm: ConvNeXtBlock's dwconv k7 -> norm, vq/conv.py:200-203; the sub-pixel upsampler's k5)
    against torch's fp32 conv1d(groups = C) + layer_norm, including causal padding, clips shorter than a pass and ragged strip ends; and the r06 row-strip
    kernel (QA_DWCONV_STRIP, the default where C is whole 256-channel chunks and k = 5 / 7) against the one-wave-per-row kernel BIT FOR BIT."""
    g = torch.Generator().manual_seed(B * 1000 + T + C + k)
    x = (torch.randn(B, T, C, generator=g) + torch.randn(B, T, 1, generator=g) * 2).to(gpu_device)
    w = (torch.randn(k, C, generator=g) / k ** 0.5).to(gpu_device)
    bias = torch.randn(C, generator=g).to(gpu_device)
    lw = (torch.rand(C, generator=g) + 0.5).to(gpu_device) if ln else None
    lb = torch.randn(C, generator=g).to(gpu_device) if ln else None
    knob("QA_DWCONV_STRIP", 0)
    rows = dwconv_cl(qa_lib, x, w, bias, lw, lb, pad_left=pad)
    knob("QA_DWCONV_STRIP", 1)
    strip = dwconv_cl(qa_lib, x, w, bias, lw, lb, pad_left=pad)
    torch.cuda.synchronize()
    assert torch.isfinite(strip).all()
    assert torch.equal(strip, rows), f"strip kernel differs from the row kernel (max |d| {float((strip - rows).abs().max()):.3e})"
    pl = k // 2 if pad < 0 else pad
    xd = F.pad(x.double().cpu().transpose(1, 2), (pl, k - 1 - pl))
    ref = F.conv1d(xd, w.double().cpu().t().unsqueeze(1).contiguous(), bias.double().cpu(), groups=C).transpose(1, 2)
    if ln:
        ref = F.layer_norm(ref, (C,), lw.double().cpu(), lb.double().cpu(), 1e-6)
    assert rel_err(strip, ref) < 2e-6


def _rvq_problem(n, Q, K, D, seed):
    rng = np.random.default_rng(seed)
    cb = np.stack([rng.standard_normal((K, D)).astype(np.float32) * (0.6 * 0.5 ** q) for q in range(Q)])
    x = (rng.standard_normal((n, D)) * 0.6).astype(np.float32)
    return x, cb


@pytest.mark.parametrize("n,Q,K,D", [(1000, 4, 1024, 512), (33, 3, 64, 128), (1, 1, 1, 8), (257, 2, 100, 64)])
def test_rvq_search_exact(qa_lib, gpu_device, n, Q, K, D):
    """Indices are integer output: bit-exact against the fp32 oracle except where the two best codes are closer than
    fp32 summation-order noise, and ALWAYS an exact arg-min of the double-precision distance up to that noise."""
    from unified_audio_amd import _lib

    x, cb = _rvq_problem(n, Q, K, D, seed=5)
    xd, cbd = torch.from_numpy(x).to(gpu_device), torch.from_numpy(cb).to(gpu_device)
    idx = torch.full((n, Q), -1, dtype=torch.int64, device=gpu_device)
    qout = torch.empty((n, D), device=gpu_device)
    _lib.check(qa_lib.qa_rvq_search(xd.data_ptr(), n, cbd.data_ptr(), Q, K, D, idx.data_ptr(), qout.data_ptr(), None))
    torch.cuda.synchronize()
    got = idx.cpu().numpy()
    assert got.min() >= 0 and got.max() < K
    excess, best, gap = rvq_c.check_f64(x, cb, got)
    scale = float((x.astype(np.float64) ** 2).sum(1).mean())
    tol = 2e-5 * scale  # fp32 dot products over D terms
    assert excess.max() <= tol, f"non-optimal code chosen: excess {excess.max()} vs tol {tol}"
    # wherever the decision is not a near-tie the index must equal the double-precision arg-min exactly
    clear = gap > tol
    assert (got[clear] == best[clear]).all()
    # and agreement with the free-running fp32 C oracle and the torch restatement is (near-)total
    ref_c = rvq_c.search_f32(x, cb)
    ref_t, _ = R.rvq_search(torch.from_numpy(x), torch.from_numpy(cb))
    frac_c = (got == ref_c).all(axis=1).mean()
    frac_t = (got == ref_t.numpy()).all(axis=1).mean()
    assert frac_c >= 0.999 and frac_t >= 0.999, (frac_c, frac_t)
    # quantized_out = stage-ordered sum of the chosen codes, bit-exact
    assert np.array_equal(qout.cpu().numpy(), rvq_c.lookup_f32(got, cb))


def test_rvq_ties_resolve_to_lowest_index(qa_lib, gpu_device):
    """Engineered exact ties: duplicated code vectors - the first copy must win (torch.max returns the first maximum)."""
    from unified_audio_amd import _lib

    rng = np.random.default_rng(3)
    K, D, n = 96, 64, 64
    cb = rng.standard_normal((1, K, D)).astype(np.float32)
    cb[0, 40:80] = cb[0, 0:40]  # codes 40..79 duplicate 0..39, which sit in other 32-code tiles / other waves
    x = cb[0, rng.integers(0, 40, size=n)] + 0.01 * rng.standard_normal((n, D)).astype(np.float32)
    xd, cbd = torch.from_numpy(x).to(gpu_device), torch.from_numpy(cb).to(gpu_device)
    idx = torch.empty((n, 1), dtype=torch.int64, device=gpu_device)
    _lib.check(qa_lib.qa_rvq_search(xd.data_ptr(), n, cbd.data_ptr(), 1, K, D, idx.data_ptr(), None, None))
    got = idx.cpu().numpy()[:, 0]
    assert (got < 40).all()
    assert np.array_equal(got, rvq_c.search_f32(x, cb)[:, 0])


def test_rvq_lookup_exact(qa_lib, gpu_device):
    from unified_audio_amd import _lib

    x, cb = _rvq_problem(10, 4, 256, 128, seed=9)
    rng = np.random.default_rng(1)
    idx = rng.integers(0, 256, size=(777, 4)).astype(np.int64)
    out = torch.empty((777, 128), device=gpu_device)
    idx_d, cb_d = torch.from_numpy(idx).to(gpu_device), torch.from_numpy(cb).to(gpu_device)
    _lib.check(qa_lib.qa_rvq_lookup(idx_d.data_ptr(), 777, cb_d.data_ptr(), 4, 256, 128, out.data_ptr(), None))
    assert np.array_equal(out.cpu().numpy(), rvq_c.lookup_f32(idx, cb))


def test_rvq_lookup_dropped_codes_contribute_zero(qa_lib, gpu_device):
    """VERDICT r04 item 9 / SURVEY 8c: upstream `get_output_from_indices` treats -1 as a DROPPED code (masked to a zero vector; the
    quantize-dropout convention of vector_quantize_pytorch) - qa_rvq_lookup does the same, bit for bit equal to the stand-in that
    restates upstream's masking, and a row that is dropped at every stage decodes to exactly 0."""
    from oracle import hcodec_ref as R
    from unified_audio_amd import _lib

    _, cb = _rvq_problem(10, 4, 256, 128, seed=9)
    rng = np.random.default_rng(2)
    idx = rng.integers(0, 256, size=(500, 4)).astype(np.int64)
    idx[rng.random((500, 4)) < 0.3] = -1
    idx[7] = -1
    idx[8, 1:] = -1  # the trailing stages dropped: what quantize dropout produces
    out = torch.empty((500, 128), device=gpu_device)
    idx_d, cb_d = torch.from_numpy(idx).to(gpu_device), torch.from_numpy(cb).to(gpu_device)
    _lib.check(qa_lib.qa_rvq_lookup(idx_d.data_ptr(), 500, cb_d.data_ptr(), 4, 256, 128, out.data_ptr(), None))
    got = out.cpu().numpy()
    assert np.array_equal(got, R.rvq_lookup(torch.from_numpy(idx), torch.from_numpy(cb)).numpy())
    assert not got[7].any() and np.array_equal(got[8], cb[0, idx[8, 0]])
    # the range check that goes with it: -1 is legal, -2 and K are not
    bad = torch.zeros(1, dtype=torch.int64, device=gpu_device)
    probe = torch.tensor([-1, 0, 255, -2, 256, -1], dtype=torch.int64, device=gpu_device)
    _lib.check(qa_lib.qa_codes_check_async(probe.data_ptr(), probe.numel(), -1, 256, bad.data_ptr(), None))
    assert int(bad.item()) == 2


@pytest.mark.parametrize("Q", [1, 4])
def test_rvq_reencode_of_quantised_vectors(qa_lib, gpu_device, Q):
    """VERDICT r04 item 9, the exact-hit case (decode -> re-encode): the input IS a sum of code vectors, so at some stage the residual
    equals a code vector up to rounding and its squared distance rounds to +-1e-7 |x|^2 around 0.  Upstream computes -cdist with
    clamp(min = 0).sqrt(), which turns every rounding-negative squared distance into an exact 0 (first index wins among THOSE); the
    reference's in-tree statement (core_vq.py:223-231) and this library take the arg-min of the squared distance itself.  The two can
    only differ when a SECOND code lies within rounding of the residual - a near-tie by the audit's definition.  Checked: the HIP search
    equals the fp64 arg-min of its own residual at every stage (no excess), and with one stage the codes come back exactly."""
    from unified_audio_amd import _lib

    n, K, D = 3000, 1024, 512
    _, cb = _rvq_problem(10, Q, K, D, seed=31)
    rng = np.random.default_rng(5)
    codes = rng.integers(0, K, size=(n, Q)).astype(np.int64)
    cbd = torch.from_numpy(cb).to(gpu_device)
    x = torch.empty((n, D), device=gpu_device)
    cd = torch.from_numpy(codes).to(gpu_device)
    _lib.check(qa_lib.qa_rvq_lookup(cd.data_ptr(), n, cbd.data_ptr(), Q, K, D, x.data_ptr(), None))
    idx = torch.full((n, Q), -7, dtype=torch.int64, device=gpu_device)
    quant = torch.empty((n, D), device=gpu_device)
    _lib.check(qa_lib.qa_rvq_search(x.data_ptr(), n, cbd.data_ptr(), Q, K, D, idx.data_ptr(), quant.data_ptr(), None))
    got, xh = idx.cpu().numpy(), x.cpu().numpy()
    excess, best, gap = rvq_c.check_f64(xh, cb, got)
    tol = 2e-5 * float((xh.astype(np.float64) ** 2).sum(1).mean())
    assert excess.max() <= tol
    assert (got[gap > tol] == best[gap > tol]).all()
    if Q == 1:  # one stage: the residual IS code vector codes[:, 0]; seeded random codebooks hold no duplicate rows
        assert np.array_equal(got, codes)
        assert np.array_equal(quant.cpu().numpy(), xh)  # and the quantised output is that code vector, bit for bit
    # the same decisions under upstream's formulation, evaluated in fp64: clamp + sqrt never changes the winner here (no second code
    # within rounding of an exact hit) - if it ever did, the audit above would already have called that stage a near-tie
    r = xh.astype(np.float64).copy()
    cb64 = cb.astype(np.float64)
    for q in range(Q):
        d2 = (r * r).sum(1)[:, None] - 2.0 * r @ cb64[q].T + (cb64[q] ** 2).sum(1)[None, :]
        up = np.sqrt(np.clip(d2, 0.0, None)).argmin(1)
        sure = gap[:, q] > tol
        assert (up[sure] == got[sure, q]).all()
        r -= cb64[q][got[:, q]]


_RVQ_GOLDEN = sorted(__import__("glob").glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rvq_corevq_*.npz")))


@pytest.mark.parametrize("path", _RVQ_GOLDEN, ids=[os.path.basename(p)[:-4] for p in _RVQ_GOLDEN])
def test_rvq_matches_reference_core_vq_golden(qa_lib, gpu_device, path):
    """HIP search / lookup against the numbers the reference's own vq/core_vq.py:223-231,394-412 produced
    (oracle/gen_golden_rvq.py).  Integer output: equal wherever the decision is not an fp32 near-tie, and the look-up is
    bit-exact."""
    from oracle import gen_golden_rvq as G
    from unified_audio_amd import _lib

    g = np.load(path)
    n, Q, K, D = int(g["n"]), int(g["Q"]), int(g["K"]), int(g["D"])
    x, cb = G.case_inputs(int(g["seed"]), n, Q, K, D)
    want = g["indices"].astype(np.int64)
    xd, cbd = torch.from_numpy(x).to(gpu_device), torch.from_numpy(cb).to(gpu_device)
    idx = torch.full((n, Q), -1, dtype=torch.int64, device=gpu_device)
    _lib.check(qa_lib.qa_rvq_search(xd.data_ptr(), n, cbd.data_ptr(), Q, K, D, idx.data_ptr(), None, None))
    got = idx.cpu().numpy()
    _audit_rvq(x, cb, got, want)
    out = torch.empty((n, D), device=gpu_device)
    wd = torch.from_numpy(want).to(gpu_device)
    _lib.check(qa_lib.qa_rvq_lookup(wd.data_ptr(), n, cbd.data_ptr(), Q, K, D, out.data_ptr(), None))
    assert np.array_equal(out.cpu().numpy()[::7, ::5], g["quant_sample"])


def _audit_rvq(x, cb, got, want):
    """Near-tie protocol for free-running residual search: walk the stages; as long as the previous stages agree the
    residual is the same, so a differing index is only acceptable if the double-precision top-2 gap at that stage is
    below fp32 dot-product noise.  After an accepted near-tie the vector's later stages see another residual and are
    audited against the double-precision arg-min of THEIR residual instead (rvq_c.check_f64 follows `got`)."""
    excess, best, gap = rvq_c.check_f64(x, cb, got)
    tol = 2e-5 * float((x.astype(np.float64) ** 2).sum(1).mean())
    assert excess.max() <= tol
    assert (got[gap > tol] == best[gap > tol]).all()
    diverged = np.zeros(got.shape[0], bool)
    for q in range(got.shape[1]):
        differs = (got[:, q] != want[:, q]) & ~diverged
        assert (gap[differs, q] <= tol).all(), f"stage {q}: index differs from the reference away from a tie"
        diverged |= differs
    assert diverged.mean() <= 0.01


# ------------------------------------------------------------------------------------------------ attention kernel alone
def _attention_alone(lib, qkv, H, hd, causal=0):
    """qa_debug_attention (exported test hook, not in the public header): attention_kernel on a fused [B, N, 3 H hd] buffer."""
    import ctypes as C

    fn = lib.qa_debug_attention
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int,
                   C.c_longlong, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
    B, N, _ = qkv.shape
    d = H * hd
    out = torch.full((B, N, d), float("nan"), device=qkv.device)
    st = fn(qkv.data_ptr(), 3 * d, qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, 3 * d, out.data_ptr(), d, B, N, N, N * 3 * d, H, hd,
            hd ** -0.5, causal, torch.cuda.current_stream().cuda_stream)
    assert st == 0, lib.qa_last_error()
    return out


@pytest.mark.parametrize("B,N,H,hd", [(4, 1500, 24, 64), (16, 1500, 8, 64), (2, 1500, 6, 128), (2, 1500, 4, 96), (3, 283, 8, 64), (2, 1531, 4, 32)])
def test_attention_kernel_long_ragged_sequences_are_exact_and_deterministic(qa_lib, gpu_device, B, N, H, hd):
    """H-Codec 2.0's working point (30 s clips: N = 1500 = 46 x 32 + 28 keys, 24 heads x 64) and other lengths that are not
    multiples of the 32-key / 128-query tiles: against softmax(QK^T / sqrt(hd)) V in double precision, and bit-identical from run to
    run.  The round-2 kernel failed BOTH at (N = 1500, hd = 64) - its predicated K / V prefetch was miscompiled (csrc/attention.hip,
    fetch()) - and no test reached that shape; the at-size parity test of H-Codec 2.0 found it."""
    d = H * hd
    g = torch.Generator().manual_seed(B * 7 + N + hd)
    qkv = torch.randn(B, N, 3 * d, generator=g).to(gpu_device)
    outs = [_attention_alone(qa_lib, qkv, H, hd) for _ in range(3)]
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    q, k, v = (t.reshape(B, N, H, hd).transpose(1, 2).double() for t in qkv.split(d, dim=2))
    ref = (torch.softmax(q @ k.transpose(2, 3) * hd ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(B, N, d)
    err = rel_err(outs[0], ref)
    assert err < 2e-6, err
