#!/usr/bin/env python
"""UndefinedBehaviorSanitizer build of the HOST side of libquarkaudio_hip.so (the .cpp orchestration: weight folding, workspace
planning, graph assembly, C-ABI argument checks) and a run of every model family through it.  The device code is unchanged (clang
ignores -fsanitize for amdgcn); the .hip objects of the normal build are reused.  Any report aborts the run (halt_on_error).
usage: python tools/ubsan_host.py            (on a GPU box; writes nothing outside tools/_variants/ubsan)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unified_audio_amd import build as B  # noqa: E402

RUN = r'''
import sys, torch
sys.path.insert(0, %r)
import unified_audio_amd as qa
from unified_audio_amd import synth
from tests.util import MINI
dev = torch.device("cuda:0")
# H-Codec 1.0 (mini) and 1.5 (reduced depth), causal variant, taps on
for kw in (dict(MINI), dict(MINI, causal=True)):
    spec = qa.HCodecSpec(**kw)
    from oracle import hcodec_ref as R
    sd = synth.hcodec10_state_dict(11, R.HCodecSpec(**kw))
    c = qa.Codec(None, None, None, spec=spec, device=dev).load_state_dict(sd)
    c.enable_taps()
    wav = synth.synth_wav(12, 3, 16 * 40); feat = synth.synth_feat(13, 3, 80, 64)
    ac, sc = c.encode(wav.to(dev).unsqueeze(1), feat.to(dev)); w = c.decode(ac, sc)
    assert torch.isfinite(w).all()
import dataclasses
o15 = dataclasses.replace(R.SPEC_15, agg_layers=2, bt_layers=2)
sd = synth.hcodec10_state_dict(1500, o15)
tok = qa.HCodecTokenizer(state_dict=sd, device=dev, spec=qa.HCodecSpec(**{f: getattr(o15, f) for f in o15.__dataclass_fields__}))
wav = synth.synth_wav(1501, 2, 640 * 24 + 50); feat = synth.synth_feat(1502, 2, 50, o15.sem_in)
codes = tok.tokenize(wav.to(dev), feats=feat.transpose(1, 2).contiguous().to(dev))
rec = tok.detokenize(**codes); assert torch.isfinite(rec).all()
# LM (small), sampled and greedy
lm = qa.LLM_SFT(device=dev).load_state_dict(synth.lm_state_dict(4321))
mix = synth.synth_feats(50, 3, 20).to(dev); mel = torch.zeros(3, 6, 80)
lm.generate("se", None, None, mel, mix, global_length=4, do_sample=False)
lm.generate("tse", mel, mix, mel, mix, global_length=4, do_sample=True)
# BiCodec, SSL, mimi streaming
m = qa.BiCodec(device=dev).load_state_dict(synth.bicodec_state_dict(77))
sem, glob = synth.bicodec_tokens(78, 2, 25)
assert torch.isfinite(m.detokenize(sem.to(dev), glob.to(dev))).all()
st = qa.StreamingTransformer(128, 4, 2, 256, causal=True, context=6, device=dev, prefix="transformer").load_state_dict(synth.mimi_state_dict(40, 128, 2, 256))
x = torch.randn(2, 9, 128, device=dev)
with st.streaming(2):
    for i in range(9): st(x[:, i:i + 1])
# round 3 host paths: LM chains (B > 32) incl. a forced chain count, fused-MLP off, the
# rolling RoPE window of an unbounded stream, the persistent LSTM and its time-out recovery
from unified_audio_amd import _lib
mix40 = synth.synth_feats(51, 40, 12).to(dev); mel40 = torch.zeros(40, 5, 80)
lm.generate("se", None, None, mel40, mix40, global_length=3, do_sample=False)
_lib.set_knob("QA_LM_CHAINS", 3); lm.generate("se", None, None, mel40, mix40, global_length=3, do_sample=True); _lib.set_knob("QA_LM_CHAINS", 0)
_lib.set_knob("QA_LM_MLP_FUSED", 0); lm2 = qa.LLM_SFT(device=dev).load_state_dict(synth.lm_state_dict(4321)); _lib.set_knob("QA_LM_MLP_FUSED", 1)
lm2.generate("se", None, None, mel, mix, global_length=4, do_sample=False)
for k, v in (("QA_LSTM_PERSISTENT", 1),):
    _lib.set_knob(k, v)
    codes = tok.tokenize(wav.to(dev), feats=feat.transpose(1, 2).contiguous().to(dev)); assert torch.isfinite(tok.detokenize(**codes)).all()
_lib.set_knob("QA_LSTM_FAULT", 1); _lib.set_knob("QA_LSTM_SPIN_LIMIT", 2048)
codes = tok.tokenize(wav.to(dev), feats=feat.transpose(1, 2).contiguous().to(dev)); assert torch.isfinite(tok.detokenize(**codes)).all()
for k, v in (("QA_LSTM_FAULT", 0), ("QA_LSTM_PERSISTENT", -1), ("QA_MIMI_ROPE_WINDOW", 8)):
    _lib.set_knob(k, v)
with st.streaming(2):
    for i in range(9): st(x[:, i:i + 1])
    st(x[:, :5])
torch.cuda.synchronize()
print("ubsan host run: ok")
'''


def main():
    B.build_library()
    out = os.path.join(ROOT, "tools", "_variants", "ubsan")
    os.makedirs(out, exist_ok=True)
    objs = []
    for src in B.SOURCES:
        obj = os.path.join(B.BUILD, src.rsplit(".", 1)[0] + ".o")
        if src.endswith(".cpp"):
            obj = os.path.join(out, src.rsplit(".", 1)[0] + ".o")
            flags = [f for f in B.FLAGS if f != "-O3"] + ["-O1", "-g", "-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-Wno-option-ignored"]
            subprocess.run([B._hipcc(), *flags, "-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
        objs.append(obj)
    lib = os.path.join(out, "libquarkaudio_hip.so")
    import glob

    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.a")) + \
        sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone_cxx-x86_64.a"))
    if len(rt) < 2:
        print("UBSan runtime archives not found under /opt/rocm/lib/llvm")
        return 2
    # the runtime is linked into the shared object itself (python is not an instrumented executable)
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-Wl,--whole-archive", rt[0], rt[-1],
                    "-Wl,--no-whole-archive", "-lpthread", "-ldl"], check=True)
    env = dict(os.environ, QA_LIBRARY=lib, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", RUN % ROOT], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-2000:])
    print(r.stderr[-3000:])
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
