"""Pins the RVQ oracle (oracle/rvq_ref.c, oracle/hcodec_ref.rvq_search / rvq_lookup, and the vector_quantize_pytorch
stand-in the imported reference Codec runs on) to the reference's OWN in-tree statement of the algorithm,
/root/reference/QuarkAudio-HCodec/HCodec-1.0/vq/core_vq.py:223-231,394-412 (`ResidualVectorQuantization.encode/decode`):
live where the reference tree is mounted, and everywhere through the golden vectors that module produced
(oracle/gen_golden_rvq.py -> tests/golden/rvq_corevq_*.npz)."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

from oracle import gen_golden_rvq as G
from oracle import hcodec_ref as R
from oracle import ref_shim, rvq_c

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rvq_corevq_*.npz")))


def _stub_rvq(cb):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(G.__file__)), "stubs"))
    try:
        import vector_quantize_pytorch as vqp
    finally:
        sys.path.pop(0)
    Q, K, D = cb.shape
    m = vqp.ResidualVQ(dim=D, codebook_size=K, num_quantizers=Q).eval()
    for q, layer in enumerate(m.layers):
        layer._codebook.embed.copy_(torch.from_numpy(cb[q])[None])
    return m


def test_rvq_golden_fixtures_present():
    assert len(GOLDEN) == len(G.CASES)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_rvq_oracles_reproduce_core_vq_golden(path):
    g = np.load(path)
    x, cb = G.case_inputs(int(g["seed"]), int(g["n"]), int(g["Q"]), int(g["K"]), int(g["D"]))
    want = g["indices"].astype(np.int64)
    # plain-C oracle, torch restatement and the package stand-in: all bit-exact on the integer output
    assert np.array_equal(rvq_c.search_f32(x, cb), want)
    idx_t, quant_t = R.rvq_search(torch.from_numpy(x), torch.from_numpy(cb))
    assert np.array_equal(idx_t.numpy(), want)
    with torch.no_grad():
        quant_s, idx_s, _ = _stub_rvq(cb)(torch.from_numpy(x)[None])
        dec_s = _stub_rvq(cb).get_output_from_indices(torch.from_numpy(want)[None])
    assert np.array_equal(idx_s[0].numpy(), want)
    # decode = stage-ordered sum of look-ups (core_vq.py:406-412): bit-exact (same fp32 additions in the same order)
    assert np.array_equal(rvq_c.lookup_f32(want, cb)[::7, ::5], g["quant_sample"])
    assert np.array_equal(R.rvq_lookup(torch.from_numpy(want), torch.from_numpy(cb)).numpy()[::7, ::5], g["quant_sample"])
    assert np.array_equal(dec_s[0].numpy()[::7, ::5], g["quant_sample"])
    assert np.array_equal(quant_t.numpy()[::7, ::5], g["quant_fwd_sample"])
    assert np.array_equal(quant_s[0].numpy()[::7, ::5], g["quant_fwd_sample"])


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference is only mounted in the build container")
@pytest.mark.parametrize("seed,n,Q,K,D", [(21, 300, 4, 1024, 512), (22, 65, 8, 256, 64), (23, 1, 2, 7, 16)])
def test_rvq_oracles_match_reference_core_vq_live(seed, n, Q, K, D):
    mod = G.load_core_vq()
    x, cb = G.case_inputs(seed, n, Q, K, D)
    idx_ref, quant_ref, quant_fwd = G.reference_rvq(mod, x, cb)
    assert np.array_equal(rvq_c.search_f32(x, cb), idx_ref)
    idx_t, quant_t = R.rvq_search(torch.from_numpy(x), torch.from_numpy(cb))
    assert np.array_equal(idx_t.numpy(), idx_ref)
    assert np.array_equal(rvq_c.lookup_f32(idx_ref, cb), quant_ref)
    assert np.array_equal(quant_t.numpy(), quant_fwd)
    with torch.no_grad():
        quant_s, idx_s, _ = _stub_rvq(cb)(torch.from_numpy(x)[None])
    assert np.array_equal(idx_s[0].numpy(), idx_ref) and np.array_equal(quant_s[0].numpy(), quant_fwd)


@pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference is only mounted in the build container")
def test_golden_files_are_what_the_reference_produces_today():
    mod = G.load_core_vq()
    for path in GOLDEN:
        g = np.load(path)
        x, cb = G.case_inputs(int(g["seed"]), int(g["n"]), int(g["Q"]), int(g["K"]), int(g["D"]))
        idx_ref, quant_ref, _ = G.reference_rvq(mod, x, cb)
        assert np.array_equal(idx_ref, g["indices"].astype(np.int64)) and np.array_equal(quant_ref[::7, ::5], g["quant_sample"])


def test_dropped_codes_are_masked_like_upstream_get_codes_from_indices():
    """SURVEY 8c / VERDICT r04 item 9: upstream `get_output_from_indices` treats index -1 as a DROPPED code - `mask = indices == -1`,
    look-up on the mask-filled indices, `masked_fill(mask, 0.)` - so the stage contributes a zero vector.  The oracle and the stand-in the
    imported reference Codec runs on both restate that; everything that is not -1 is unchanged (bit for bit the plain sum of look-ups).
    Unpinned against the pip package itself (absent, no network): this pins the two restatements to each other and to the definition."""
    rng = np.random.default_rng(4)
    Q, K, D = 4, 32, 16
    cb = rng.standard_normal((Q, K, D)).astype(np.float32)
    idx = torch.from_numpy(rng.integers(0, K, size=(3, 9, Q)).astype(np.int64))
    plain = R.rvq_lookup(idx, torch.from_numpy(cb))
    with torch.no_grad():
        assert torch.equal(_stub_rvq(cb).get_output_from_indices(idx), plain)
    drop = idx.clone()
    drop[0, 0, :] = -1
    drop[1, 2, 1:] = -1
    drop[2, 5, 2] = -1
    got = R.rvq_lookup(drop, torch.from_numpy(cb))
    with torch.no_grad():
        assert torch.equal(_stub_rvq(cb).get_output_from_indices(drop), got)
    assert not got[0, 0].any()
    assert torch.equal(got[1, 2], torch.from_numpy(cb[0, idx[1, 2, 0]]))
    want = torch.from_numpy(cb[0, idx[2, 5, 0]]) + torch.from_numpy(cb[1, idx[2, 5, 1]]) + torch.from_numpy(cb[3, idx[2, 5, 3]])
    assert torch.allclose(got[2, 5], want, atol=1e-6)
    keep = torch.ones(3, 9, dtype=torch.bool)
    keep[0, 0] = keep[1, 2] = keep[2, 5] = False
    assert torch.equal(got[keep], plain[keep])
