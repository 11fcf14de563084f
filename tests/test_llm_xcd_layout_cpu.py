"""CPU check of the per-XCD persistent decode kernel's DATA FLOW (csrc/lm_xcd.hip, QA_LM_XCD - written without GPU time left).

The kernel cannot run here, but everything about it that is index arithmetic can be checked on the CPU: the weight layouts come from
the library itself (qa_debug_lm_xcd_pack = the exact code build_lm runs), and the lane / wave / block arithmetic of lmx_gemv16 /
lmx_gemv128, the row -> (head, rotary pair) maps, the SwiGLU pairing, the residual wiring and the slot-wise arg-max are transcribed
below from the kernel and run on one decode step of a random Llama body - against a plain numpy step that knows nothing of slots.
What this does NOT cover is what only hardware shows: barriers, hand-off visibility, the MFMA operand convention (taken from the
working lm_gemv4_kernel)."""
import ctypes as C

import numpy as np
import pytest

from unified_audio_amd import _lib

D, H, HD, I = 512, 8, 64, 2048


@pytest.fixture(scope="module")
def lib():
    lib = _lib.load_library()
    fn = lib.qa_debug_lm_xcd_pack
    fn.restype = C.c_int
    fn.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    return fn


def pack(fn, kind, w0, w1, K, slot, aux, n_out):
    w0 = np.ascontiguousarray(w0, dtype=np.float32)
    w1 = None if w1 is None else np.ascontiguousarray(w1, dtype=np.float32)
    out = np.empty(n_out, dtype=np.float32)
    st = fn(kind, w0.ctypes.data, None if w1 is None else w1.ctypes.data, K, slot, aux, out.ctypes.data)
    assert st == 0
    return out


def gemv16(packed, x, K):
    """lmx_gemv16: wave w, block (rb = block & 3, kp = block >> 2): A = the 4 lanes' KW weights, B = the sequences' K slice 4 w + kp."""
    KW = K // 32
    P = packed.reshape(8, KW // 4, 64, 4).astype(np.float64)
    out = np.zeros((16, 4))
    for w in range(8):
        for block in range(16):
            rb, kp = block & 3, block >> 2
            A = np.stack([P[w, :, 4 * block + r, :].reshape(KW) for r in range(4)])  # [r][k]
            B = x[:, (4 * w + kp) * KW:(4 * w + kp + 1) * KW]                          # [q][k]
            out[4 * rb:4 * rb + 4] += A @ B.T
    return out  # [row 0..15][q]


def gemv128(packed, x):
    """lmx_gemv128: wave -> (group g = w & 1, K slice ks = w >> 1 of 128), lane = row of the group."""
    P = packed.reshape(2, 4, 32, 64, 4).astype(np.float64)
    out = np.zeros((128, 4))
    for g in range(2):
        for ks in range(4):
            for block in range(16):
                A = np.stack([P[g, ks, :, 4 * block + r, :].reshape(128) for r in range(4)])
                out[64 * g + 4 * block:64 * g + 4 * block + 4] += A @ x[:, ks * 128:(ks + 1) * 128].T
    return out


def rms(x, eps):
    return x / np.sqrt((x * x).mean(-1, keepdims=True) + eps)


def rope_tables(pos):
    inv = 1.0 / (10000.0 ** (np.arange(0, HD, 2) / HD))
    return np.cos(pos * inv), np.sin(pos * inv)


def rotate(v, c, s):  # rotate-half RoPE over the last axis (HD)
    v1, v2 = v[..., :HD // 2], v[..., HD // 2:]
    return np.concatenate([v1 * c - v2 * s, v2 * c + v1 * s], axis=-1)


def test_one_decode_step_through_the_slot_layouts_equals_a_plain_llama_step(lib):
    rng = np.random.default_rng(5)
    nq, pos, eps, lo, width = 3, 6, 1e-5, 3, 4096
    V = lo + width
    W = {n: rng.standard_normal(s).astype(np.float32) * 0.05 for n, s in
         dict(q=(D, D), k=(D, D), v=(D, D), o=(D, D), g=(I, D), u=(I, D), dn=(D, I), head=(V, D)).items()}
    x0 = rng.standard_normal((nq, D)) * 0.5
    kc = rng.standard_normal((nq, pos, D)) * 0.5  # cached (already rotated) keys / values of positions 0 .. pos - 1
    vc = rng.standard_normal((nq, pos, D)) * 0.5
    cs, sn = rope_tables(pos)

    # ---- plain step (LlamaDecoderLayer + final norm + head slice + arg-max); norm gains are 1 (the library folds them into W)
    Wd = {n: w.astype(np.float64) for n, w in W.items()}
    h = rms(x0, eps)
    q = rotate((h @ Wd["q"].T).reshape(nq, H, HD), cs, sn)
    k = rotate((h @ Wd["k"].T).reshape(nq, H, HD), cs, sn)
    v = (h @ Wd["v"].T).reshape(nq, H, HD)
    K_all = np.concatenate([kc.reshape(nq, pos, H, HD), k[:, None]], axis=1)
    V_all = np.concatenate([vc.reshape(nq, pos, H, HD), v[:, None]], axis=1)
    sc = np.einsum("bhd,bthd->bht", q, K_all) / np.sqrt(HD)
    p = np.exp(sc - sc.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    att = np.einsum("bht,bthd->bhd", p, V_all).reshape(nq, D)
    x1 = x0 + att @ Wd["o"].T
    h2 = rms(x1, eps)
    gt, up = h2 @ Wd["g"].T, h2 @ Wd["u"].T
    x2 = x1 + ((gt / (1 + np.exp(-gt))) * up) @ Wd["dn"].T
    logits = rms(x2, eps) @ Wd["head"][lo:lo + width].T
    want_tok = lo + logits.argmax(-1)

    # ---- the same step slot by slot, as lm_xcd_decode_kernel lays it out
    xs = np.zeros((4, D))
    xs[:nq] = x0
    rs = 1.0 / np.sqrt((xs * xs).mean(-1) + eps)
    q_e, k_e, v_e = np.zeros((nq, D)), np.zeros((nq, D)), np.zeros((nq, D))
    for slot in range(32):  # S1
        for part, (name, dst) in enumerate((("q", q_e), ("k", k_e), ("v", v_e))):
            out = gemv16(pack(lib, 0, W[name], None, D, slot, part, 16 * D), xs, D) * rs[None, :]
            for rb in range(4):
                for qq in range(nq):
                    val = out[4 * rb:4 * rb + 4, qq]
                    if part == 2:
                        dst[qq, 16 * slot + 4 * rb:16 * slot + 4 * rb + 4] = val
                    else:
                        for pr in range(2):
                            P = slot * 8 + 2 * rb + pr
                            hh, jj = P >> 5, P & 31
                            v1, v2 = val[2 * pr], val[2 * pr + 1]
                            dst[qq, hh * HD + jj] = v1 * cs[jj] - v2 * sn[jj]
                            dst[qq, hh * HD + jj + 32] = v2 * cs[jj] + v1 * sn[jj]
    assert np.allclose(q_e, q.reshape(nq, D), atol=1e-5) and np.allclose(k_e, k.reshape(nq, D), atol=1e-5)
    assert np.allclose(v_e, v.reshape(nq, D), atol=1e-5)
    xatt = np.zeros((4, D))
    xatt[:nq] = att  # S2 is lm_attn_kernel's arithmetic (tested on the GPU); S3 consumes its merged output
    x1_e = np.zeros((4, D))
    for slot in range(32):  # S3
        out = gemv16(pack(lib, 1, W["o"], None, D, slot, 0, 16 * D), xatt, D)
        x1_e[:nq, 16 * slot:16 * slot + 16] = x0[:, 16 * slot:16 * slot + 16] + out[:, :nq].T
    assert np.allclose(x1_e[:nq], x1, atol=1e-5)
    rs2 = 1.0 / np.sqrt((x1_e * x1_e).mean(-1) + eps)
    act = np.zeros((4, I))
    for slot in range(32):  # S4
        out = gemv128(pack(lib, 2, W["g"], W["u"], D, slot, 0, 128 * D), x1_e)
        g_, u_ = out[:64] * rs2[None, :], out[64:] * rs2[None, :]
        act[:, 64 * slot:64 * slot + 64] = ((g_ / (1 + np.exp(-g_))) * u_).T
    x2_e = np.zeros((4, D))
    for slot in range(32):  # S5
        out = gemv16(pack(lib, 1, W["dn"], None, I, slot, 0, 16 * I), act, I)
        x2_e[:nq, 16 * slot:16 * slot + 16] = x1_e[:nq, 16 * slot:16 * slot + 16] + out[:, :nq].T
    assert np.allclose(x2_e[:nq], x2, atol=1e-5)
    rows_wg = width // 32  # head: per-slot maxima, then the fold over the 32 slots (first maximum wins)
    best = np.full((nq, 32), -np.inf)
    bidx = np.zeros((nq, 32), dtype=np.int64)
    for slot in range(32):
        for ch in range(rows_wg // 128):
            out = gemv128(pack(lib, 3, W["head"], None, D, slot, lo + slot * rows_wg + ch * 128, 128 * D), x2_e)
            for qq in range(nq):
                j = int(out[:, qq].argmax())
                if out[j, qq] > best[qq, slot]:
                    best[qq, slot], bidx[qq, slot] = out[j, qq], slot * rows_wg + ch * 128 + j
    got_tok = np.array([lo + bidx[qq, int(best[qq].argmax())] for qq in range(nq)])
    assert (got_tok == want_tok).all()
