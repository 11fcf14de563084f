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
    lspec = L.LMSpec(hidden=128, n_layers=2, n_heads=2, global_size=64, semantic_size=128, feats_dim=96)
    lm = qa.LLM_SFT(feats_dim=96, llm_base_config=dict(global_size=64, semantic_size=128, hidden_size=128, num_layers=2,
                                                       num_attention_heads=2), device=device)
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
