#!/usr/bin/env python
"""Diagnostic: which stage of H-Codec 2.0 decode / encode differs between a clip inside a batch and the same clip alone?
Prints the largest absolute difference per tap (0 = bit-identical)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unified_audio_amd as qa  # noqa: E402
from unified_audio_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SEC = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dev = torch.device("cuda:0")
spec = synth.Shapes20()
sd = synth.hcodec20_state_dict(1234, spec)
codec = qa.Codec(None, None, None, spec=qa.SPEC_20, device=dev).load_state_dict(sd)
T = int(SEC * 48000) // spec.frame_hop * spec.frame_hop
wav = synth.synth_wav_fullband(1235, B, T).to(dev)
feat = synth.synth_feat(1236, B, T // spec.hop, 768).to(dev)
codec.enable_taps(True)
P0 = "encoder.post_net.1.layers.0"
ENC = ["enc.stft", "enc.prior", P0 + ".self_attn.rnn", P0 + ".qkv", P0 + ".qkv_rope", P0 + ".att", P0 + ".x_attn", P0 + ".act", P0 + ".x_mlp",
       "encoder.post_net.1.layers.1.self_attn.rnn", "enc.emb", "enc.sem"]
DEC = ["dec.embed", "dec.prior_res1", "decoder.prior_net.3.layers.0.self_attn.rnn", "decoder.prior_net.3.layers.1.self_attn.rnn",
       "dec.transformer", "dec.prior", "dec.backbone", "dec.spec"]


def taps(names):
    out = {}
    for n in names:
        try:
            out[n] = codec.tap(n).clone()
        except Exception as e:  # noqa: BLE001
            out[n] = None
    return out


ac, sc = codec.encode(wav, feat)
te = taps(ENC)
rec = codec.decode(ac, sc)
td = taps(DEC)
ac2, sc2 = codec.encode(wav, feat)
te2 = taps(ENC)
print("determinism of the batch run:", {n.split(".")[-1]: (None if te[n] is None else float((te[n] - te2[n]).abs().max())) for n in ENC})
d3 = spec.enc_dim

for i in (0, B - 1):
    a1, s1 = codec.encode(wav[i:i + 1], feat[i:i + 1])
    te1 = taps(ENC)
    r1 = codec.decode(ac[i:i + 1], sc[i:i + 1])
    td1 = taps(DEC)
    print(f"clip {i}: codes equal {torch.equal(a1[0], ac[i]) and torch.equal(s1[0], sc[i])}, wav max diff {float((r1[0] - rec[i]).abs().max()):.3e}")
    for names, full, one in ((ENC, te, te1), (DEC, td, td1)):
        for n in names:
            if full[n] is None or one[n] is None:
                print(f"   {n}: (no tap)")
                continue
            per = full[n].numel() // B
            d = (full[n].view(B, per)[i] - one[n].view(-1)).abs()
            print(f"   {n}: max diff {float(d.max()):.3e}  (differing elements {int((d > 0).sum())} of {per})")
            if n.endswith(".att") and float(d.max()) > 0:
                N50 = per // spec.enc_dim
                dd = d.view(N50, spec.enc_dim // 64, 64)
                print("      differing (frame, head) pairs per head:", (dd.amax(dim=2) > 0).sum(dim=0).tolist())
                fr = torch.nonzero((dd.amax(dim=2) > 0).any(dim=1)).flatten()
                print("      differing frames: count", fr.numel(), "first", fr[:12].tolist(), "last", fr[-6:].tolist())
