#!/usr/bin/env python
"""Diagnostic for the H-Codec 2.0 range-stress golden: HIP-path encoder embedding against the oracle's on this host, normal and stress weights."""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_audio_amd as qa  # noqa: E402
from oracle import hcodec20_ref as R20  # noqa: E402
from oracle import hcodec_ref as R  # noqa: E402
from oracle import rvq_c, synth  # noqa: E402
from oracle.gen_golden import SPEC20_SMALL  # noqa: E402

dev = torch.device("cuda:0")
for name in ("hcodec20_small_b2", "hcodec20_small_b2_stress"):
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name + ".npz"))
    seed, o = int(g["seed"]), R20.HCodec20Spec(**SPEC20_SMALL)
    sd = synth.hcodec20_state_dict(seed, o)
    if "stress" in g.files and int(g["stress"]):
        sd = synth.stress_state_dict(sd)
    pspec = qa.HCodecSpec(version=20, enc_dim=o.enc_dim, enc_inter=o.enc_inter, enc_convnext_layers=o.enc_convnext_layers,
                          enc_layers=o.enc_transformer_layers, frame_stride=o.stride, tr_inter_cap=o.tr_inter_cap, dimension=o.dimension,
                          code_dim=o.dimension, sem_in=o.sem_in, sem_ch=o.sem_ch, sem_strides=o.sem_strides, codebook_size=o.codebook_size,
                          num_quantizers=o.num_quantizers, dec_dim=o.dec_dim, dec_inter=o.dec_inter, dec_heads=o.dec_dim // 64,
                          dec_layers=o.dec_transformer_layers, convnext_layers=o.dec_convnext_layers, n_fft=o.n_fft, hop=o.hop)
    tok = qa.HCodecTokenizer(state_dict=sd, device=dev, spec=pspec)
    tok.model.enable_taps(True)
    wav = synth.synth_wav_fullband(seed + 1, int(g["batch"]), int(g["samples"]))
    feat = synth.synth_feat(seed + 2, int(g["batch"]), R.pad_wav(wav, 3840).shape[-1] // o.hop, o.sem_in)
    ac, sc = tok.tokenize(wav.to(dev), feats=feat.transpose(1, 2).contiguous().to(dev))
    taps = {}
    with torch.no_grad():
        ac_o, sc_o = R20.encode(sd, R.pad_wav(wav, 3840), feat, o, taps)
    ref_ac, ref_sc = g["acoustic_codes"].astype(np.int64), g["semantic_codes"].astype(np.int64)
    print(name, "oracle-on-this-host == golden:", bool(np.array_equal(ac_o.numpy(), ref_ac)), bool(np.array_equal(sc_o.numpy(), ref_sc)),
          " HIP == golden:", bool(np.array_equal(ac.cpu().numpy(), ref_ac)), bool(np.array_equal(sc.cpu().numpy(), ref_sc)))
    for tapname, key, cbname, got, want in (("enc.emb", "enc.emb", "quantizer", ac, ref_ac), ("enc.sem", "enc.sem", "semantic_quantizer", sc, ref_sc)):
        e_o = taps[key]                                          # [B, D, N]
        B, D, N = e_o.shape
        e_h = tok.model.tap(tapname).cpu().reshape(B, N, D).transpose(1, 2)   # channel-last flat copy
        rel = float((e_h - e_o).norm() / e_o.norm())
        cb = R.rvq_codebooks(sd, cbname, o.num_quantizers).numpy()
        x = e_o.transpose(1, 2).reshape(B * N, D).contiguous().numpy()
        flat = lambda c: np.ascontiguousarray(np.asarray(c).transpose(0, 2, 1).reshape(B * N, -1))  # noqa: E731
        for who, codes in (("HIP", got.cpu().numpy()), ("oracle-here", (ac_o if key == "enc.emb" else sc_o).numpy())):
            excess, best, gap = rvq_c.check_f64(x, cb, flat(codes))
            scale = float((x.astype(np.float64) ** 2).sum(1).mean())
            d = flat(codes) != flat(want)
            print(f"   {tapname}: |emb_hip - emb_oracle| / |emb| = {rel:.2e}   E|x|^2 = {scale:.3f}  {who}: {int(d.any(1).sum())} vectors differ from the golden, "
                  f"largest gap at a differing decision {float(gap[d].max() / scale) if d.any() else 0.0:.2e}, largest excess {float(excess.max() / scale):.2e}")
