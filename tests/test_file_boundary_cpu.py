"""The FILE side of the drop-in boundary, host logic only (no device): the configuration files the reference's constructors read are
parsed the way the reference parses them.

    BiCodec.load_from_checkpoint      QuarkAudio-UniSE/model/bicodec/bicodec.py:69-115  (config.yaml['audio_tokenizer'] -> module constructors)
    load_config                       QuarkAudio-UniSE/model/bicodec/utils/file.py:116-130 (OmegaConf.load + base_config merge)
    TestDataset                       QuarkAudio-UniSE/dataloader/data_module.py:296-410
    Model.test_step                   QuarkAudio-UniSE/model/model.py:170-290 (batch tuple, save_enhanced file names)
"""
import copy
import math
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import ref_bicodec_shim, ref_shim
from tests import ref_configs as RC

live = pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference is only mounted in the build container")


@live
def test_restated_configs_equal_the_shipped_yaml_files():
    root = os.path.join(ref_shim.REFERENCE_ROOT, "QuarkAudio-HCodec")
    assert yaml.safe_load(open(os.path.join(root, "HCodec-1.5", "conf", "config_adaptive_v3.yaml"))) == RC.HCODEC15_CONFIG
    assert yaml.safe_load(open(os.path.join(root, "HCodec-2.0", "conf", "large_12.5hz_config.yaml"))) == RC.HCODEC20_CONFIG


def test_bicodec_spec_from_the_published_config(qa_lib):
    import unified_audio_amd as qa

    assert qa.BiCodecSpec.from_config(RC.BICODEC_CONFIG["audio_tokenizer"]) == qa.SPEC_BICODEC
    small = qa.BiCodecSpec.from_config(RC.small_bicodec_config()["audio_tokenizer"])
    assert (small.latent_dim, small.codebook_size, small.spk_latent_dim, small.vocos_dim, small.vocos_inter, small.vocos_layers,
            small.gen_channels) == (64, 128, 32, 32, 64, 2, 512)
    assert small.hop == 320 and small.global_size == 4096


def test_bicodec_config_errors_follow_the_reference_constructors(qa_lib):
    """A key a constructor without **kwargs does not take is a TypeError there (Decoder / WaveGenerator / SpeakerEncoder), a missing
    required argument is a TypeError, FactorizedVectorQuantize swallows extras (**kwargs); a value the detokenizer has no kernels for
    is refused BY NAME instead of being ignored."""
    import unified_audio_amd as qa

    base = RC.BICODEC_CONFIG["audio_tokenizer"]

    def cfg(block, **kw):
        c = copy.deepcopy(base)
        for k, v in kw.items():
            if v is KeyError:
                c[block].pop(k)
            else:
                c[block][k] = v
        return c

    with pytest.raises(TypeError, match="unexpected"):
        qa.BiCodecSpec.from_config(cfg("prenet", dilation=3))
    with pytest.raises(TypeError, match="unexpected"):
        qa.BiCodecSpec.from_config(cfg("decoder", upsample_initial=4))
    with pytest.raises(TypeError, match="missing"):
        qa.BiCodecSpec.from_config(cfg("decoder", rates=KeyError))
    with pytest.raises(TypeError, match="missing"):
        qa.BiCodecSpec.from_config(cfg("quantizer", commitment=KeyError))
    assert qa.BiCodecSpec.from_config(cfg("quantizer", some_future_training_flag=1)) == qa.SPEC_BICODEC
    c = copy.deepcopy(base)
    c.pop("speaker_encoder")
    with pytest.raises(KeyError):
        qa.BiCodecSpec.from_config(c)
    for block, kw, word in (("prenet", dict(sample_ratios=[2, 2]), "sample_ratios"), ("prenet", dict(use_tanh_at_final=True), "tanh"),
                            ("speaker_encoder", dict(fsq_num_quantizers=2), "fsq_num_quantizers"), ("decoder", dict(d_out=2), "d_out"),
                            ("quantizer", dict(codebook_dim=1024), "Identity"), ("prenet", dict(condition_dim=None), "condition_dim")):
        with pytest.raises(qa.QuarkAudioError, match=word):
            qa.BiCodecSpec.from_config(cfg(block, **kw))
    with pytest.raises(ValueError, match="agree"):
        qa.BiCodecSpec.from_config(cfg("decoder", input_channel=512))
    # other depths / rates are configuration, not refusals
    s = qa.BiCodecSpec.from_config(cfg("decoder", rates=[8, 5, 4], kernel_sizes=[16, 11, 8], channels=768))
    assert s.hop == 160 and s.gen_channels == 768 and len(s.kernel_sizes) == 3


def test_load_config_merges_base_config_and_refuses_interpolation(qa_lib, tmp_path):
    import unified_audio_amd as qa
    from unified_audio_amd.bicodec import load_config

    base = tmp_path / "base.yaml"
    base.write_text(yaml.safe_dump(dict(RC.BICODEC_CONFIG, sample_rate=16000)))
    over = tmp_path / "config.yaml"
    over.write_text(yaml.safe_dump({"base_config": str(base), "audio_tokenizer": {"prenet": {"vocos_num_layers": 6}}, "ref_segment_duration": 6}))
    cfg = load_config(str(over))
    assert cfg["sample_rate"] == 16000 and cfg["ref_segment_duration"] == 6  # OmegaConf.merge(base, config): mappings merge key by key
    assert cfg["audio_tokenizer"]["prenet"]["vocos_num_layers"] == 6 and cfg["audio_tokenizer"]["prenet"]["vocos_dim"] == 384
    assert qa.BiCodecSpec.from_config(cfg["audio_tokenizer"]).vocos_layers == 6
    bad = tmp_path / "interp.yaml"
    bad.write_text("audio_tokenizer:\n  decoder:\n    input_channel: ${audio_tokenizer.quantizer.input_dim}\n")
    with pytest.raises(qa.QuarkAudioError, match="interpolation"):
        load_config(str(bad))


@live
@pytest.mark.parametrize("which", ["published", "small"])
def test_reference_constructors_accept_the_blocks_and_give_the_tensors_the_spec_promises(qa_lib, which):
    """bicodec.py:80-87 verbatim - `Encoder(**config["encoder"])`, `FactorizedVectorQuantize(**config["quantizer"])`, `Decoder(**config["prenet"])`,
    `Decoder(**config["postnet"])`, `WaveGenerator(**config["decoder"])`, `SpeakerEncoder(**config["speaker_encoder"])` - with the
    reference's OWN classes: every block is accepted, and the detokenizer-side parameters have exactly the names and shapes that
    `synth.bicodec_state_dict(BiCodecSpec.from_config(...))` (= what qa_bicodec_create consumes) has."""
    import unified_audio_amd as qa
    from unified_audio_amd import synth

    config = (RC.BICODEC_CONFIG if which == "published" else RC.small_bicodec_config())["audio_tokenizer"]
    imp = ref_bicodec_shim._import
    mods = torch.nn.ModuleDict(dict(
        encoder=imp("encoder_decoder.feat_encoder").Encoder(**config["encoder"]),
        quantizer=imp("vq.factorized_vector_quantize").FactorizedVectorQuantize(**config["quantizer"]),
        prenet=imp("encoder_decoder.feat_decoder").Decoder(**config["prenet"]),
        postnet=imp("encoder_decoder.feat_decoder").Decoder(**config["postnet"]),
        decoder=imp("encoder_decoder.wave_generator").WaveGenerator(**config["decoder"]),
        speaker_encoder=imp("speaker.speaker_encoder").SpeakerEncoder(**config["speaker_encoder"])))
    ref_shapes = {k: tuple(v.shape) for k, v in mods.state_dict().items() if k.startswith(ref_bicodec_shim.DETOK_PREFIXES)}
    spec = qa.BiCodecSpec.from_config(config)
    ours = {k: tuple(v.shape) for k, v in synth.bicodec_state_dict(3, spec).items()}
    assert set(ours) <= set(ref_shapes), sorted(set(ours) - set(ref_shapes))[:5]
    assert {k: ref_shapes[k] for k in ours} == ours
    # and nothing of the detokenizer is missing from what the library is given (buffers like cluster_size aside)
    missing = [k for k in ref_shapes if k not in ours and not k.endswith(("cluster_size", "embed_avg", "initted"))]
    assert not missing, missing[:5]


def _write(path, x, sr=16000, subtype="FLOAT"):
    from unified_audio_amd import audio_io

    audio_io.write_wav(str(path), x, sr, subtype)


def test_test_dataset_yields_the_reference_batch_tuple(qa_lib, tmp_path):
    """data_module.py:338-384: (mode, enroll [1, 5 s] wrapped or cut and scaled to 0.99, src [1, T], tgt, fs LongTensor([16000]), lengths, [stem])."""
    from unified_audio_amd.unise import TestDataset

    g = torch.Generator().manual_seed(0)
    src_d, enr_d = tmp_path / "mix", tmp_path / "aux"
    src_d.mkdir(), enr_d.mkdir()
    utts = {"a.wav": torch.randn(30000, generator=g) * 0.1, "b.wav": torch.randn(123457, generator=g) * 0.1}
    enrs = {"a.wav": torch.randn(20000, generator=g) * 0.05, "b.wav": torch.randn(100000, generator=g) * 0.05}
    for n, x in utts.items():
        _write(src_d / n, x)
        _write(enr_d / n, enrs[n])
    ds = TestDataset(str(enr_d), str(src_d), str(src_d), "tse", enroll_duration=5.0, device="cpu")
    assert len(ds) == 2
    seen = {}
    for mode, enroll, src, tgt, fs, lengths, names in ds:
        assert mode == "tse" and fs.dtype == torch.int64 and fs.tolist() == [16000] and len(names) == 1
        seen[names[0]] = (enroll, src, tgt, int(lengths[0]))
    assert sorted(seen) == ["a", "b"]
    for stem in ("a", "b"):
        enroll, src, tgt, length = seen[stem]
        x, e = utts[stem + ".wav"].numpy()[None], enrs[stem + ".wav"].numpy()[None]
        assert src.shape == (1, x.shape[1]) and length == x.shape[1] and torch.equal(src, tgt) and np.array_equal(src.numpy(), x)
        n = 80000  # data_module.py:346-352
        want = np.pad(e, [(0, 0), (0, n - e.shape[-1])], mode="wrap") if e.shape[-1] < n else e[..., :n]
        want = want / (np.max(np.abs(want)) + 1e-5) * 0.99
        assert enroll.shape == (1, n) and np.allclose(enroll.numpy(), want, rtol=0, atol=1e-7)
    se = TestDataset(None, str(src_d), str(src_d), "se", device="cpu")
    assert all(b[1] is None for b in se)
    assert [b[6][0] for b in TestDataset(None, str(src_d), str(src_d), "se", device="cpu", rank=1, world_size=2)] == \
        [sorted(utts)[1][:-4]] or len(list(TestDataset(None, str(src_d), str(src_d), "se", device="cpu", rank=1, world_size=2))) == 1
    with pytest.raises(AssertionError):
        TestDataset(None, str(src_d), str(src_d), "se", batch_size=2)


class _FakeSSL:
    def __call__(self, wavs):
        return wavs[:, :6:2].unsqueeze(-1).repeat(1, 1, 4)


class _FakeLM:
    def __init__(self):
        self.calls = []

    def generate(self, task_name, enroll_mel, enroll_feats, mix_mel, mix_feats, do_sample):
        B = mix_feats.shape[0]
        self.calls.append((task_name, B, None if enroll_mel is None else enroll_mel.size(1)))
        key = (mix_feats[:, 0, 0].abs() * 1e4).long() + (0 if enroll_feats is None else (enroll_feats[:, 0, 0].abs() * 1e4).long())
        return torch.zeros(B, 32, dtype=torch.int64), (key[:, None] + torch.arange(mix_mel.size(1))[None]) % 7  # a row's tokens depend on that row only


class _FakeTok:
    def detokenize(self, global_tokens, semantic_tokens):  # [B, 1, 32], [B, N] -> [B, 1, N * 320]
        return (semantic_tokens.float() / 16.0).repeat_interleave(320, dim=1).unsqueeze(1)


def test_model_test_step_takes_the_reference_batch_and_writes_the_reference_file_names(qa_lib, tmp_path):
    """model.py:170-228: `{save_enhanced}/{names[0]}.wav` for 'se' / 'tse', `_s1` / `_s2` for 'ss', nothing when save_enhanced is absent
    or None, an unknown mode falls through silently; `est` has the source's length."""
    from unified_audio_amd import audio_io
    from unified_audio_amd.unise import Model

    out = tmp_path / "enh"
    out.mkdir()
    lm = _FakeLM()
    m = Model({"save_enhanced": str(out)}, device="cpu", semantic_model=_FakeSSL(), tokenizer=_FakeTok(), dnn=lm)
    src = torch.randn(1, 90000) * 0.1
    fs, lengths = torch.tensor([16000]), torch.tensor([90000])
    est = m.test_step(("se", None, src, src, fs, lengths, ["utt1"]), 0)
    assert est.shape == (90000,) and sorted(os.listdir(out)) == ["utt1.wav"]
    back, sr = audio_io.read_wav(str(out / "utt1.wav"))
    assert sr == 16000 and back.shape == (1, 90000) and float((back[0] - est.clamp(-1, 1)).abs().max()) <= 1.0 / 32768 + 1e-7
    enroll = torch.randn(1, 48000) * 0.1
    m.test_step(("tse", enroll, src, src, fs, lengths, ["utt2"]), 1)
    assert lm.calls[-1] == ("tse", 2, 150)
    s1, s2 = m.test_step(("ss", None, src, src, fs, lengths, ["utt3"]), 2)
    assert s1.shape == s2.shape == (90000,)
    assert sorted(os.listdir(out)) == ["utt1.wav", "utt2.wav", "utt3_s1.wav", "utt3_s2.wav"]
    assert m.test_step(("asr", None, src, src, fs, lengths, ["utt4"]), 3) is None and len(os.listdir(out)) == 4
    with pytest.raises(ValueError):
        m.test_step(("se", None, torch.zeros(2, 100), None, fs, lengths, ["x", "y"]), 0)
    with pytest.raises(ValueError):
        m.test_step(("tse", None, src, src, fs, lengths, ["utt5"]), 0)
    # no save_enhanced (test.py:15-17 only sets it when the flag is given): nothing is written
    n = len(os.listdir(out))
    for cfg in ({}, {"save_enhanced": None}):
        m2 = Model(cfg, device="cpu", semantic_model=_FakeSSL(), tokenizer=_FakeTok(), dnn=_FakeLM())
        assert m2.test_step(("se", None, src, src, fs, lengths, ["utt6"]), 0).shape == (90000,)
    assert len(os.listdir(out)) == n
    # the batched form equals one step at a time, file by file, with enrollments of DIFFERENT lengths kept as they are
    srcs = [torch.randn(1, k) * 0.1 for k in (70000, 170001, 80000)]
    enrs = [torch.randn(1, k) * 0.1 for k in (80000, 48000, 80000)]
    batches = [("tse", e, s, s, fs, torch.tensor([s.size(-1)]), [f"f{i}"]) for i, (s, e) in enumerate(zip(srcs, enrs))]
    lm3 = _FakeLM()
    m3 = Model({"save_enhanced": None}, device="cpu", semantic_model=_FakeSSL(), tokenizer=_FakeTok(), dnn=lm3)
    together = m3.test_steps(batches)
    assert sorted(c[2] for c in lm3.calls) == [150, 250] and sorted(c[1] for c in lm3.calls) == [2, 3]  # one pass per enrollment length: 3 + 2... segments
    for b, t in zip(batches, together):
        assert torch.equal(m3.test_step(b, 0), t)
    with pytest.raises(ValueError, match="semantic_model_path"):
        Model({"codec_ckpt_dir": "x", "llm_config": {}}, device="cpu", tokenizer=_FakeTok(), dnn=_FakeLM())


@pytest.mark.skipif(not __import__("oracle.ref_unise_shim", fromlist=["x"]).reference_available(), reason="/root/reference is only mounted in the build container")
def test_model_test_step_equals_the_references_test_step_on_its_own_components(qa_lib, tmp_path):
    """The adapter around the reference's OWN semantic model / LLM_SFT / BiCodec modules against the reference's OWN `Model.test_step` on
    the same batch tuple: the same file names and the same samples."""
    from tests.test_unise_driver_pin_cpu import _RefLM, _utt
    from oracle import bicodec_ref as BR
    from oracle import llm_ref as L
    from oracle import ref_llm_shim
    from oracle import ref_unise_shim as RU
    from oracle import ssl_ref as S
    from transformers import WavLMModel
    from unified_audio_amd import synth
    from unified_audio_amd.unise import Model

    sspec = S.SSLSpec(conv_dim=(32,) * 7, hidden_size=48, num_hidden_layers=2, num_attention_heads=2, intermediate_size=96,
                      num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2, num_buckets=32, max_bucket_distance=100, compress_exponent=0.0)
    wavlm = WavLMModel(S.hf_config(sspec, "wavlm")).eval()
    wavlm.load_state_dict(S.synth_state_dict(4, sspec, "wavlm"), strict=False)
    bspec = BR.BiCodecSpec(latent_dim=32, codebook_size=128, codebook_dim=8, spk_latent_dim=16, token_num=32, vocos_dim=16, vocos_inter=32,
                           vocos_layers=1, gen_channels=64, rates=(8, 5, 4, 2), kernel_sizes=(16, 11, 8, 4))
    detok = ref_bicodec_shim.load_reference_detokenizer(bspec, synth.bicodec_state_dict(5, bspec))
    lspec = L.LMSpec(hidden=64, n_layers=2, n_heads=2, global_size=4096, semantic_size=128, feats_dim=48)
    lm = ref_llm_shim.load_state(ref_llm_shim.load_reference_llm(lspec), L.lm_state_dict(8, lspec))
    ref = RU.load_reference_model(wavlm, lm, detok, save_dir=str(tmp_path / "ref"))
    ours = Model({"save_enhanced": str(tmp_path)}, device="cpu", semantic_model=ref.extract_semantic_features, tokenizer=ref.tokenizer, dnn=_RefLM(lm))
    fs = torch.tensor([16000])
    # one 5 s segment each (the two-segment and wrapped-tail cases of the same glue are in tests/test_unise_driver_pin_cpu.py): 5 generate passes on the host
    for mode, src, enroll, name in (("se", _utt(1, 50000), None, "n1"), ("tse", _utt(2, 60001), _utt(3, 48000), "n2"), ("ss", _utt(4, 50000), None, "n3")):
        batch = (mode, enroll, src, src, fs, torch.tensor([src.size(-1)]), [name])
        del RU.WRITTEN[:]
        ref.test_step(batch, 0)
        written = [(os.path.basename(p), torch.from_numpy(np.asarray(x)).clone(), sr) for p, x, sr in RU.WRITTEN]
        got = ours.test_step(batch, 0)
        got = list(got) if mode == "ss" else [got]
        want_names = [f"{name}_s1.wav", f"{name}_s2.wav"] if mode == "ss" else [f"{name}.wav"]
        assert [w[0] for w in written] == want_names and all(w[2] == 16000 for w in written)
        for (fname, x, _), g in zip(written, got):
            assert os.path.isfile(tmp_path / fname) and torch.equal(g, x)
