"""Pins the UniSE driver (unified_audio_amd/unise.py) to the reference's OWN `Model.test_step` (QuarkAudio-UniSE/model/model.py:170-290)
run in this container (oracle/ref_unise_shim.py): both drive the SAME component models - transformers' WavLMModel, the reference's
LLM_SFT, the reference's BiCodec modules, seeded random weights - so every difference would be glue: wrap padding, 5 s segmentation,
peak normalisation, mel frame count, enrollment tiling, the 'ss' mode's enrollment clip / scaling, `est.reshape(-1)[:len]`."""
import pytest
import torch

from oracle import bicodec_ref as BR
from oracle import llm_ref as L
from oracle import ref_bicodec_shim, ref_llm_shim
from oracle import ref_unise_shim as RU
from oracle import ssl_ref as S
from unified_audio_amd import synth
from unified_audio_amd import unise as U

pytestmark = pytest.mark.skipif(not RU.reference_available(), reason="/root/reference is only mounted in the build container")


class _RefLM:
    """The reference's LLM_SFT behind the driver: the driver hands over frame counts (`_Frames`), the reference's generate reads
    `.size()` AND `.device` of a mel tensor (llm_sft.py:108-110) - so a real tensor of that many frames is made here; a wrong frame
    count in the driver would change the number of semantic steps and fail the comparison."""

    def __init__(self, lm):
        self.lm = lm

    def generate(self, task_name, enroll_mel, enroll_feats, mix_mel, mix_feats, do_sample=False):
        mk = lambda f: None if f is None else torch.zeros(f.size(0), f.size(1), 80)  # noqa: E731
        return self.lm.generate(task_name=task_name, enroll_mel=mk(enroll_mel), enroll_feats=enroll_feats, mix_mel=mk(mix_mel),
                                mix_feats=mix_feats, do_sample=do_sample)


@pytest.fixture(scope="module")
def rig():
    from transformers import WavLMModel

    sspec = S.SSLSpec(conv_dim=(32,) * 7, hidden_size=48, num_hidden_layers=2, num_attention_heads=2, intermediate_size=96,
                      num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2, num_buckets=32, max_bucket_distance=100,
                      compress_exponent=0.0)
    wavlm = WavLMModel(S.hf_config(sspec, "wavlm")).eval()
    wavlm.load_state_dict(S.synth_state_dict(4, sspec, "wavlm"), strict=False)
    bspec = BR.BiCodecSpec(latent_dim=32, codebook_size=128, codebook_dim=8, spk_latent_dim=16, token_num=32, vocos_dim=16, vocos_inter=32,
                           vocos_layers=1, gen_channels=64, rates=(8, 5, 4, 2), kernel_sizes=(16, 11, 8, 4))
    detok = ref_bicodec_shim.load_reference_detokenizer(bspec, synth.bicodec_state_dict(5, bspec))
    lspec = L.LMSpec(hidden=64, n_layers=2, n_heads=2, global_size=4096, semantic_size=128, feats_dim=48)
    lm = ref_llm_shim.load_state(ref_llm_shim.load_reference_llm(lspec), L.lm_state_dict(8, lspec))
    ref = RU.load_reference_model(wavlm, lm, detok)
    # the driver around the very same objects: the reference's own extract_semantic_features, LLM_SFT and detokenize
    drv = U.UniSE(_RefLM(lm), ref.extract_semantic_features, detokenize=ref.tokenizer.detokenize)
    return ref, drv


def _utt(seed, n):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, n, generator=g) * 0.1


@pytest.mark.parametrize("n", [90000, 30000, 80000])  # two segments with a wrapped tail; shorter than one segment (wraps 3 x); exact
def test_se_matches_reference_test_step(rig, n):
    ref, drv = rig
    src = _utt(n, n)
    want, = RU.run_test_step(ref, "se", src)
    got, = drv.enhance("se", [src])
    assert got.shape == want.shape == (n,)
    assert torch.equal(got, want)


def test_tse_matches_reference_test_step(rig):
    ref, drv = rig
    src, enroll = _utt(1, 170001), _utt(2, 32000)
    want, = RU.run_test_step(ref, "tse", src, enroll)
    got, = drv.enhance("tse", [src], [enroll])
    assert got.shape == want.shape == (170001,) and torch.equal(got, want)


@pytest.mark.parametrize("n", [90000, 50000])  # longer / shorter than the 5 s the first pass cuts
def test_ss_matches_reference_test_step(rig, n):
    ref, drv = rig
    src = _utt(10 + n, n)
    w1, w2 = RU.run_test_step(ref, "ss", src)
    (g1, g2), = drv.enhance("ss", [src])
    assert g1.shape == w1.shape == (n,) and torch.equal(g1, w1) and torch.equal(g2, w2)


def test_batched_utterances_equal_reference_one_at_a_time(rig):
    """The reference takes one utterance per call (batch_size 1); the driver's multi-utterance batch must give each of them the
    reference's result."""
    ref, drv = rig
    srcs = [_utt(21, 70000), _utt(22, 100000)]
    got = drv.enhance("se", srcs)
    for s, g in zip(srcs, got):
        want, = RU.run_test_step(ref, "se", s)
        assert float((g - want).abs().max()) < 1e-5  # batch composition changes BLAS blocking: not bit-equal, same tokens
