"""UniSE driver on the HIP components: batched multi-utterance processing == one utterance at a time (the reference's mode),
and == the hand-written composition of test_step's 'se' / 'tse' branches (QuarkAudio-UniSE/model/model.py:170-222)."""
import pytest
import torch

from oracle import llm_ref as L
from oracle import ssl_ref as S

pytestmark = pytest.mark.gpu


def _components(device):
    import unified_audio_amd as qa

    sspec = S.SSLSpec(conv_dim=(64,) * 7, hidden_size=96, num_hidden_layers=2, num_attention_heads=3, intermediate_size=192,
                      num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2, num_buckets=32, max_bucket_distance=100,
                      compress_exponent=0.0)
    fx = qa.SSLFeatureExtractor(qa.SSLSpec(**{f: getattr(sspec, f) for f in sspec.__dataclass_fields__}), device=device)
    fx.load_state_dict(S.synth_state_dict(4, sspec, "wavlm"))
    lspec = L.LMSpec(hidden=256, n_layers=2, n_heads=4, global_size=64, semantic_size=128, feats_dim=96)
    lm = qa.LLM_SFT(feats_dim=96, llm_base_config=dict(global_size=64, semantic_size=128, hidden_size=256, num_layers=2,
                                                       num_attention_heads=4), device=device)
    lm.load_state_dict(L.lm_state_dict(8, lspec))
    return fx, lm


def test_batched_utterances_equal_one_at_a_time(qa_lib, gpu_device):
    from unified_audio_amd import unise as U

    fx, lm = _components(gpu_device)
    drv = U.UniSE(lm, fx)
    g = torch.Generator().manual_seed(1)
    srcs = [(torch.randn(1, n, generator=g) * 0.1).to(gpu_device) for n in (70000, 170001, 80000)]
    enrs = [(torch.randn(1, 32000, generator=g) * 0.1).to(gpu_device) for _ in srcs]
    for mode, e in (("se", None), ("tse", enrs)):
        batched = drv.enhance_tokens(mode, srcs, e)
        assert [b[0].shape for b in batched] == [(1, 32), (3, 32), (1, 32)] and batched[1][1].shape == (3, 250)
        for i, src in enumerate(srcs):
            one = drv.enhance_tokens(mode, [src], None if e is None else [e[i]])[0]
            assert torch.equal(one[0], batched[i][0]) and torch.equal(one[1], batched[i][1])
        # the reference's own sequence of calls for utterance 1 (model.py:176-190 / 197-219)
        src = srcs[1]
        seg = U.segment(src, normalise=(mode == "se"))
        mix_feats = fx(seg)
        mel = torch.zeros(seg.size(0), U.mel_frames(U.SEG_LEN), 80)
        if mode == "se":
            gi, si = lm.generate("se", None, None, mel, mix_feats, do_sample=False)
        else:
            ef = fx(e[1])
            emel = torch.zeros(seg.size(0), U.mel_frames(e[1].size(-1)), 80)
            gi, si = lm.generate("tse", emel, torch.cat([ef for _ in range(seg.size(0))], dim=0), mel, mix_feats, do_sample=False)
        assert torch.equal(gi, batched[1][0]) and torch.equal(si, batched[1][1])
        assert int(si.max()) < 128 and int(gi.max()) < 64


def test_enhance_end_to_end_against_the_oracle_chain(qa_lib, gpu_device):
    """Model.test_step 'se' and 'ss' on the HIP components (WavLM front-end -> AR-LM -> BiCodec detokenize) against the CPU oracles
    chained the same way (ssl_ref -> llm_ref.generate -> bicodec_ref.detokenize): tokens equal up to audited near-ties, waveform of
    the same tokens within the north_star tolerance, output length = input length."""
    import dataclasses

    import unified_audio_amd as qa
    from oracle import bicodec_ref as BR
    from tests.test_llm_gpu import _audit
    from unified_audio_amd import synth
    from unified_audio_amd import unise as U

    sspec = S.SSLSpec(conv_dim=(64,) * 7, hidden_size=96, num_hidden_layers=2, num_attention_heads=3, intermediate_size=192,
                      num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2, num_buckets=32, max_bucket_distance=100,
                      compress_exponent=0.0)
    ssl_sd = S.synth_state_dict(4, sspec, "wavlm")
    fx = qa.SSLFeatureExtractor(qa.SSLSpec(**{f: getattr(sspec, f) for f in sspec.__dataclass_fields__}), device=gpu_device).load_state_dict(ssl_sd)
    bspec = BR.BiCodecSpec(latent_dim=64, codebook_size=128, codebook_dim=8, spk_latent_dim=32, token_num=32, vocos_dim=32, vocos_inter=64,
                           vocos_layers=2, gen_channels=512, rates=(8, 5, 4, 2), kernel_sizes=(16, 11, 8, 4))
    lspec = L.LMSpec(hidden=256, n_layers=2, n_heads=4, global_size=4096, semantic_size=128, feats_dim=96)
    lm_sd = L.lm_state_dict(8, lspec)
    lm = qa.LLM_SFT(feats_dim=96, llm_base_config=dict(global_size=4096, semantic_size=128, hidden_size=256, num_layers=2,
                                                       num_attention_heads=4), device=gpu_device).load_state_dict(lm_sd)
    bsd = synth.bicodec_state_dict(5, bspec)
    bic = qa.BiCodec(qa.BiCodecSpec(**{f: getattr(bspec, f) for f in bspec.__dataclass_fields__}), device=gpu_device).load_state_dict(bsd)
    drv = U.UniSE(lm, fx, tokenizer=qa.BiCodecTokenizer(model=bic))
    g = torch.Generator().manual_seed(3)
    src = torch.randn(1, 90000, generator=g) * 0.1
    # ---- 'se'
    (gids, sids), = drv.enhance_tokens("se", [src.to(gpu_device)])
    seg = U.segment(src, normalise=True)
    with torch.no_grad():
        feats = S.extract_features(ssl_sd, seg, sspec)
    _audit(lm_sd, lspec, "se", None, feats, 250, 32, gids.cpu(), sids.cpu(), tol=5e-4)  # features differ by ~1e-6 from the oracle's
    est, = drv.enhance("se", [src.to(gpu_device)])
    assert est.shape == (90000,)
    want = BR.detokenize(bsd, sids.cpu(), gids.cpu().unsqueeze(1), bspec).squeeze(1).reshape(-1)[:90000]
    assert float((est.cpu() - want).pow(2).mean().sqrt()) < 1e-3
    # ---- 'ss': two waveforms of the mixture's length, reproducible, and different from each other
    (s1, s2), = drv.enhance("ss", [src.to(gpu_device)])
    (t1, t2), = drv.enhance("ss", [src.to(gpu_device)])
    assert s1.shape == s2.shape == (90000,) and torch.equal(s1, t1) and torch.equal(s2, t2) and not torch.equal(s1, s2)
    assert torch.isfinite(s1).all() and torch.isfinite(s2).all()


def test_pipelined_driver_is_bit_identical_to_the_sequential_one(qa_lib, gpu_device):
    """UniSE.enhance_pipelined (three streams: WavLM of batch k + 1 | LM of batch k | BiCodec of batch k - 1, micro-batches of a few
    segments, LM on replayed step graphs) against UniSE.enhance on the same utterances: identical samples, 'se' and 'tse', incl. a
    ragged last micro-batch and micro-batches that mix utterances."""
    import unified_audio_amd as qa
    from oracle import bicodec_ref as BR
    from unified_audio_amd import synth
    from unified_audio_amd import unise as U

    sspec = S.SSLSpec(conv_dim=(64,) * 7, hidden_size=96, num_hidden_layers=2, num_attention_heads=3, intermediate_size=192,
                      num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2, num_buckets=32, max_bucket_distance=100,
                      compress_exponent=0.0)
    fx = qa.SSLFeatureExtractor(qa.SSLSpec(**{f: getattr(sspec, f) for f in sspec.__dataclass_fields__}), device=gpu_device)
    fx.load_state_dict(S.synth_state_dict(4, sspec, "wavlm"))
    bspec = BR.BiCodecSpec(latent_dim=64, codebook_size=128, codebook_dim=8, spk_latent_dim=32, token_num=32, vocos_dim=32, vocos_inter=64,
                           vocos_layers=2, gen_channels=512, rates=(8, 5, 4, 2), kernel_sizes=(16, 11, 8, 4))
    lspec = L.LMSpec(hidden=256, n_layers=2, n_heads=4, global_size=4096, semantic_size=128, feats_dim=96)
    lm = qa.LLM_SFT(feats_dim=96, llm_base_config=dict(global_size=4096, semantic_size=128, hidden_size=256, num_layers=2,
                                                       num_attention_heads=4), device=gpu_device).load_state_dict(L.lm_state_dict(8, lspec))
    bic = qa.BiCodec(qa.BiCodecSpec(**{f: getattr(bspec, f) for f in bspec.__dataclass_fields__}), device=gpu_device)
    bic.load_state_dict(synth.bicodec_state_dict(5, bspec))
    drv = U.UniSE(lm, fx, tokenizer=qa.BiCodecTokenizer(model=bic))
    g = torch.Generator().manual_seed(3)
    srcs = [(torch.randn(1, n, generator=g) * 0.1).to(gpu_device) for n in (90000, 250001, 80000, 161000)]  # 2 + 4 + 1 + 3 segments
    enrs = [(torch.randn(1, 32000, generator=g) * 0.1).to(gpu_device) for _ in srcs]
    for mode, e in (("se", None), ("tse", enrs)):
        want = drv.enhance(mode, srcs, e)
        for m in (3, 4, 16):
            got = drv.enhance_pipelined(mode, srcs, e, segments_per_batch=m)
            torch.cuda.synchronize()
            assert len(got) == len(want)
            for a, b, src in zip(got, want, srcs):
                assert a.shape == b.shape == (src.size(-1),) and torch.equal(a, b), (mode, m)
