"""The FILE side of the drop-in boundary on the device: every constructor signature of the reference that reads weights or a
configuration FROM DISK is exercised with files this test writes, and must give results bit-identical to the `state_dict=` / `spec=`
path the other suites use.

    HCodecTokenizer(pt_path)                            QuarkAudio-HCodec/HCodec-1.0/audio_tokenizer.py:21-26   torch.load(pt_path)
    HCodecTokenizer(config={... 'ckpt_path': ...})      HCodec-1.5/audio_tokenizer.py:20-25,40-47                plain state_dict or {'state_dict': ...}
    HCodecTokenizer(pt_path, config_path, device)       HCodec-2.0/audio_tokenizer.py:19-46                      YAML + torch.load, shipped large_12.5hz values
    BiCodec.load_from_checkpoint(model_dir)             QuarkAudio-UniSE/model/bicodec/bicodec.py:69-115         config.yaml + model.safetensors
    BiCodecTokenizer(model_dir)                         model/bicodec/audio_tokenizer.py:33-47                   {model_dir}/config.yaml, {model_dir}/BiCodec/
    Model(config) + Lightning checkpoint + test_step    model/model.py:20-36,82-91,170-228; test.py:11-30
    tools/unise_infer.py --config ... --save_enhanced   test.py's command line
"""
import copy
import json
import os
import shutil
import subprocess
import sys

import pytest
import torch
import yaml

from oracle import hcodec20_ref as R20
from oracle import hcodec_ref as R
from oracle import llm_ref as L
from oracle import ssl_ref as S
from tests import ref_configs as RC
from unified_audio_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same_codec(tok_a, tok_b, wav, feat, adaptive=False):
    a, b = tok_a.tokenize(wav, feats=feat), tok_b.tokenize(wav, feats=feat)
    if adaptive:
        assert sorted(a) == sorted(b)
        for k in a:
            assert torch.equal(a[k], b[k]), k
        wa, wb = tok_a.detokenize(**a), tok_b.detokenize(**b)
    else:
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        wa, wb = tok_a.detokenize(*a), tok_b.detokenize(*b)
    assert wa.shape == wb.shape and torch.equal(wa, wb) and torch.isfinite(wa).all()


def test_hcodec10_tokenizer_from_a_pt_path(qa_lib, gpu_device, tmp_path):
    """`HCodecTokenizer(pt_path)`: `torch.load(pt_path, map_location='cpu')` of a plain state_dict, the hard-coded 1.0 architecture."""
    import unified_audio_amd as qa

    sd = synth.hcodec10_state_dict(11)
    pt = tmp_path / "weights.pt"
    torch.save(sd, pt)
    from_file = qa.HCodecTokenizer(str(pt))                      # the reference's call, audio_tokenizer.py:73
    from_dict = qa.HCodecTokenizer(state_dict=sd, device=gpu_device)
    assert from_file.model.spec == qa.SPEC_10 and from_file.hop_length == 640 and from_file.device.type == "cuda"
    wav = synth.synth_wav(12, 2, 31000).to(gpu_device)
    feat = synth.synth_feat(13, 2, R.pad_wav(wav.cpu()).shape[-1] // 320).transpose(1, 2).contiguous().to(gpu_device)
    _same_codec(from_file, from_dict, wav, feat)
    with pytest.raises(FileNotFoundError):
        qa.HCodecTokenizer(str(tmp_path / "missing.pt"))


@pytest.mark.parametrize("wrapped", [False, True], ids=["plain_state_dict", "state_dict_wrapper"])
def test_hcodec15_tokenizer_from_config_ckpt_path(qa_lib, gpu_device, tmp_path, wrapped):
    """`HCodecTokenizer(config=config)`: `load_sub_weights(config['ckpt_path'], prefix=None)` takes a plain state_dict or a
    `{'state_dict': ...}` checkpoint (HCodec-1.5/audio_tokenizer.py:20-25).  The config is the shipped one with 2-layer stacks (what the
    reference's Codec accepts just as well; the published depth is covered by tests/golden/hcodec15_b2_full_depth)."""
    import dataclasses

    import unified_audio_amd as qa

    cfg = copy.deepcopy(RC.HCODEC15_CONFIG)
    for agg in cfg["adaptive_config"]["aggregators"].values():
        agg["num_layers"] = 2
    cfg["adaptive_config"]["transformer_kwargs"]["num_layers"] = 2
    ospec = dataclasses.replace(R.SPEC_15, agg_layers=2, bt_layers=2, threshold=0.6)
    sd = synth.hcodec10_state_dict(21, ospec)
    ck = tmp_path / "hcodec15.pt"
    torch.save({"state_dict": sd, "epoch": 3} if wrapped else sd, ck)
    cfg["ckpt_path"] = str(ck)
    from_file = qa.HCodecTokenizer(config=cfg)                   # the reference's call, HCodec-1.5/audio_tokenizer.py:109
    from_dict = qa.HCodecTokenizer(state_dict=sd, device=gpu_device, spec=from_file.model.spec)
    assert from_file.model.spec.adaptive and from_file.model.spec.agg_layers == 2 and from_file.model.spec.threshold == pytest.approx(0.6)
    assert from_file.select_layers == (11, 14, 16)
    wav = synth.synth_wav(22, 2, 40000).to(gpu_device)
    feat = synth.synth_feat(23, 2, R.pad_wav(wav.cpu()).shape[-1] // 320, 1024).transpose(1, 2).contiguous().to(gpu_device)
    _same_codec(from_file, from_dict, wav, feat, adaptive=True)


def test_hcodec20_tokenizer_from_pt_path_config_path_device(qa_lib, gpu_device, tmp_path):
    """`HCodecTokenizer(pt_path, config_path, device)` with the SHIPPED large_12.5hz configuration (tests/ref_configs.HCODEC20_CONFIG =
    the file's values; 1.17 G parameters, a 4.7 GB checkpoint file)."""
    import unified_audio_amd as qa

    if shutil.disk_usage(tmp_path).free < 7 * 2 ** 30:
        pytest.skip("needs 7 GB of scratch disk for the full-size H-Codec 2.0 checkpoint")
    cfg_path = tmp_path / "large_12.5hz_config.yaml"
    cfg_path.write_text(yaml.safe_dump(RC.HCODEC20_CONFIG))
    o = R20.HCodec20Spec()
    sd = synth.hcodec20_state_dict(31, o)
    pt = tmp_path / "large_12.5hz_weights.pt"
    torch.save(sd, pt)
    from_file = qa.HCodecTokenizer(str(pt), str(cfg_path), gpu_device)   # the reference's call, HCodec-2.0/audio_tokenizer.py:89
    assert from_file.model.spec == qa.SPEC_20 and from_file.hop_length == 3840 and from_file.sampling_rate == 48000
    from_dict = qa.HCodecTokenizer(state_dict=sd, device=gpu_device, spec=qa.SPEC_20)
    del sd
    wav = synth.synth_wav_fullband(32, 1, 3 * 3840 * 4 + 100).to(gpu_device)
    feat = synth.synth_feat(33, 1, R.pad_wav(wav.cpu(), 3840).shape[-1] // o.hop, o.sem_in).transpose(1, 2).contiguous().to(gpu_device)
    _same_codec(from_file, from_dict, wav, feat)
    # the reference's default device='cpu' has no counterpart: the tokenizer lands on the HIP device, never on a host fallback
    assert qa.HCodecTokenizer(str(pt), str(cfg_path)).device.type == "cuda"


def _write_bicodec_dir(d, config, sd):
    from safetensors.torch import save_file

    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.yaml"), "w") as f:
        yaml.safe_dump(config, f)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, "model.safetensors"))


@pytest.mark.parametrize("which", ["published", "small"])
def test_bicodec_from_a_checkpoint_directory(qa_lib, gpu_device, tmp_path, which):
    """`BiCodec.load_from_checkpoint(model_dir)`: the architecture comes from `config.yaml['audio_tokenizer']`, the weights from
    `model.safetensors` - at the published Spark-TTS values and at a reduced configuration (other widths, depths: proof that the file is
    what builds the model, not the defaults)."""
    import unified_audio_amd as qa

    config = RC.BICODEC_CONFIG if which == "published" else RC.small_bicodec_config()
    spec = qa.BiCodecSpec.from_config(config["audio_tokenizer"])
    sd = synth.bicodec_state_dict(41, spec)
    sd["encoder.some_training_side_tensor"] = torch.zeros(3)           # encoder-side entries of the real checkpoint are ignored
    _write_bicodec_dir(str(tmp_path / "BiCodec"), config, sd)
    from_file = qa.BiCodec.load_from_checkpoint(str(tmp_path / "BiCodec"), device=gpu_device)
    from_dict = qa.BiCodec(spec, device=gpu_device).load_state_dict(sd)
    assert from_file.spec == spec and (spec == qa.SPEC_BICODEC) == (which == "published")
    sem, glob = synth.bicodec_tokens(42, 2, 50, spec)
    a = from_file.detokenize(sem.to(gpu_device), glob.to(gpu_device))
    b = from_dict.detokenize(sem.to(gpu_device), glob.to(gpu_device))
    assert a.shape == (2, 1, 50 * spec.hop) and torch.equal(a, b) and torch.isfinite(a).all()
    # the tokenizer facade of model/bicodec/audio_tokenizer.py:33-47: {model_dir}/config.yaml + {model_dir}/BiCodec
    with open(tmp_path / "config.yaml", "w") as f:
        yaml.safe_dump(RC.SPARKTTS_CONFIG, f)
    tok = qa.BiCodecTokenizer(str(tmp_path), device=gpu_device)
    assert tok.config["latent_hop_length"] == 320
    assert torch.equal(tok.detokenize(glob.to(gpu_device).unsqueeze(1), sem.to(gpu_device)), a)
    if which == "small":  # a config that disagrees with the weights fails at load time, by tensor name - not at the first call
        bad = copy.deepcopy(config)
        bad["audio_tokenizer"]["decoder"]["channels"] = 1024
        _write_bicodec_dir(str(tmp_path / "bad"), bad, sd)
        with pytest.raises(qa.QuarkAudioError, match="decoder"):
            qa.BiCodec.load_from_checkpoint(str(tmp_path / "bad"), device=gpu_device)


SSPEC = S.SSLSpec(conv_dim=(64,) * 7, hidden_size=96, num_hidden_layers=2, num_attention_heads=3, intermediate_size=192,
                  num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=2, num_buckets=32, max_bucket_distance=100, compress_exponent=0.0)
LSPEC = L.LMSpec(hidden=256, n_layers=2, n_heads=4, global_size=4096, semantic_size=128, feats_dim=96)


def _unise_files(tmp_path):
    """A checkpoint tree in the reference's layout: Lightning .ckpt with `dnn.*`, codec_ckpt_dir/{config.yaml, BiCodec/...}, a WavLM snapshot."""
    from safetensors.torch import save_file

    lm_sd = L.lm_state_dict(8, LSPEC)
    ckpt = tmp_path / "epoch=20-step=109367.ckpt"
    torch.save({"state_dict": {"dnn." + k: v for k, v in lm_sd.items()}, "epoch": 20, "global_step": 109367,
                "hyper_parameters": {"config": {"note": "what save_hyperparameters() stores, model.py:23"}}}, ckpt)
    bconfig = RC.small_bicodec_config()
    bspec_sd = None
    import unified_audio_amd as qa

    bspec = qa.BiCodecSpec.from_config(bconfig["audio_tokenizer"])
    bspec_sd = synth.bicodec_state_dict(5, bspec)
    codec_dir = tmp_path / "checkpoints"
    _write_bicodec_dir(str(codec_dir / "BiCodec"), bconfig, bspec_sd)
    with open(codec_dir / "config.yaml", "w") as f:
        yaml.safe_dump(RC.SPARKTTS_CONFIG, f)
    ssl_sd = S.synth_state_dict(4, SSPEC, "wavlm")
    wavlm = tmp_path / "wavlm-base-plus"
    wavlm.mkdir()
    save_file({k: v.contiguous() for k, v in ssl_sd.items()}, str(wavlm / "model.safetensors"))
    hf = S.hf_config(SSPEC, "wavlm").to_dict()
    (wavlm / "config.json").write_text(json.dumps(hf, default=str))
    config = {  # the keys of conf/config.yaml that Model.__init__ / test_step read (+ semantic_model_path: no network here)
        "ckpt_path": str(ckpt), "codec_ckpt_dir": str(codec_dir), "semantic_model_path": str(wavlm),
        "stft_config": dict(hop_length=320, win_length=640, n_fft=640, n_mels=80),
        "llm_config": dict(num_tasks=3, task_map=dict(se=0, tse=1, rtse=2), feats_dim=96,
                           llm_base_config=dict(cond_dim=80, global_size=4096, semantic_size=128, hidden_size=256, num_layers=2, num_attention_heads=4,
                                                dropout_p=0.1, max_position_embeddings=4096, label_smoothing=0.1)),
    }
    return config, lm_sd, bspec, bspec_sd, ssl_sd


def test_unise_model_from_config_and_lightning_checkpoint(qa_lib, gpu_device, tmp_path):
    """`Model(config)` builds the three stages from FILES - BiCodecTokenizer(model_dir=codec_ckpt_dir), LLM_SFT(**llm_config), the WavLM
    snapshot - restores the Lightning checkpoint's `dnn.*` entries, and `test_step(batch, batch_idx)` with the reference's batch tuple
    writes `save_enhanced/{name}.wav`: the same samples as the driver built from state_dicts."""
    import unified_audio_amd as qa
    from unified_audio_amd import audio_io
    from unified_audio_amd.unise import Model, UniSE

    config, lm_sd, bspec, bsd, ssl_sd = _unise_files(tmp_path)
    out = tmp_path / "enhanced"
    out.mkdir()
    config["save_enhanced"] = str(out)                       # test.py:15-17
    model = Model(config, device=gpu_device)
    fx = qa.SSLFeatureExtractor(qa.SSLSpec(**{f: getattr(SSPEC, f) for f in SSPEC.__dataclass_fields__}), device=gpu_device).load_state_dict(ssl_sd)
    assert model.semantic_model.spec == fx.spec                # config.json carried the architecture
    lm = qa.LLM_SFT(feats_dim=96, llm_base_config=dict(global_size=4096, semantic_size=128, hidden_size=256, num_layers=2, num_attention_heads=4),
                    device=gpu_device).load_state_dict(lm_sd)
    bic = qa.BiCodec(bspec, device=gpu_device).load_state_dict(bsd)
    drv = UniSE(lm, fx, tokenizer=qa.BiCodecTokenizer(model=bic))
    g = torch.Generator().manual_seed(3)
    fs = torch.tensor([16000])
    src, enroll = torch.randn(1, 90000, generator=g) * 0.1, torch.randn(1, 80000, generator=g) * 0.1
    est = model.test_step(("se", None, src, src, fs, torch.tensor([90000]), ["p232_001"]), 0)
    want, = drv.enhance("se", [src.to(gpu_device)])
    assert est.shape == (90000,) and torch.equal(est, want)
    back, sr = audio_io.read_wav(str(out / "p232_001.wav"))
    assert sr == 16000 and back.shape == (1, 90000) and float((back[0] - est.cpu().clamp(-1, 1)).abs().max()) <= 1.0 / 32768 + 1e-7
    est = model.test_step(("tse", enroll, src, src, fs, torch.tensor([90000]), ["mix_7"]), 1)
    want, = drv.enhance("tse", [src.to(gpu_device)], [enroll.to(gpu_device)])
    assert torch.equal(est, want) and os.path.isfile(out / "mix_7.wav")
    s1, s2 = model.test_step(("ss", None, src, src, fs, torch.tensor([90000]), ["two"]), 2)
    (w1, w2), = drv.enhance("ss", [src.to(gpu_device)])
    assert torch.equal(s1, w1) and torch.equal(s2, w2) and sorted(os.listdir(out)) == ["mix_7.wav", "p232_001.wav", "two_s1.wav", "two_s2.wav"]
    # a checkpoint without dnn.* entries is not a UniSE checkpoint
    with pytest.raises(KeyError):
        model.load_state_dict({"generator.x": torch.zeros(1)})


def test_enrollments_of_different_lengths_keep_their_lengths(qa_lib, gpu_device, tmp_path):
    """the CLI used to cut every enrollment to the shortest; the reference (one file per step) never does.  A batch
    with enrollments of 3 s and 5 s equals the two files run alone."""
    import unified_audio_amd as qa
    from unified_audio_amd.unise import Model

    config, *_ = _unise_files(tmp_path)
    model = Model(config, device=gpu_device)
    g = torch.Generator().manual_seed(5)
    fs = torch.tensor([16000])
    batches = [("tse", torch.randn(1, n_e, generator=g) * 0.1, torch.randn(1, n, generator=g) * 0.1, None, fs, torch.tensor([n]), [f"u{i}"])
               for i, (n, n_e) in enumerate(((100000, 48000), (85000, 80000), (60000, 48000)))]
    together = model.test_steps(batches)
    for b, t in zip(batches, together):
        assert torch.equal(model.test_step(b, 0), t)


def test_cli_runs_the_reference_command_line_end_to_end(qa_lib, gpu_device, tmp_path):
    """`python tools/unise_infer.py --config conf.yaml --save_enhanced DIR` - test.py's two flags - over a data_src_dir of wav files, and
    `--synthetic` (seeded weights at the published sizes) with positional wav files."""
    from unified_audio_amd import audio_io

    config, *_ = _unise_files(tmp_path)
    src_dir = tmp_path / "noisy"
    src_dir.mkdir()
    g = torch.Generator().manual_seed(9)
    for name, n in (("a.wav", 50000), ("b.wav", 90000)):
        audio_io.write_wav(str(src_dir / name), torch.randn(n, generator=g) * 0.1, 16000)
    config["dataset_config"] = {"test_kwargs": dict(batch_size=1, num_workers=1, prefetch=1, mode="se", data_enroll_dir=None, enroll_duration=5.0,
                                                    data_src_dir=str(src_dir), data_tgt_dir=str(src_dir))}
    conf = tmp_path / "config.yaml"
    conf.write_text(yaml.safe_dump(config))
    out = tmp_path / "out"
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "unise_infer.py"), "--config", str(conf), "--save_enhanced", str(out)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sorted(os.listdir(out)) == ["a.wav", "b.wav"]
    assert audio_io.read_wav(str(out / "b.wav"))[0].shape == (1, 90000)
    out2 = tmp_path / "out2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "unise_infer.py"), "--synthetic", "--mode", "se", "--save_enhanced", str(out2),
                        str(src_dir / "a.wav")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    w, sr = audio_io.read_wav(str(out2 / "a.wav"))
    assert sr == 16000 and w.shape == (1, 50000) and float(w.abs().max()) > 0
