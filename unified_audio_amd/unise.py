"""Segmenting / batching driver of the UniSE inference step, in front of the HIP paths of this package.

    UniSE.enhance_tokens(mode, src, enroll)  <->  Model.test_step, 'se' and 'tse' branches
                                                  (QuarkAudio-UniSE/model/model.py:170-222): wrap-pad to a multiple of 5 s,
                                                  cut into 5 s segments, (se: divide by the utterance peak), WavLM
                                                  features, LLM_SFT.generate -> (global_ids, semantic_ids)

The reference handles ONE utterance per call (`batch_size == 1`, dataloader/data_module.py:340); here any number of
utterances is accepted: their segments are concatenated into one batch for the front-end and the LM (segments are
independent), and the results are handed back per utterance.  The last stage of `test_step` - BiCodec `detokenize` - is
`unified_audio_amd.BiCodecTokenizer`: with it `UniSE.enhance` returns waveforms for 'se', 'tse' and the three-pass
'ss' mode (model.py:223-290); without it `enhance_tokens` returns the tokens.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import torch

SEG_LEN = 5 * 16000        # model.py:176
HOP_LENGTH = 320           # conf/config.yaml:124-128 (stft_config)
WIN_LENGTH = 640


def wrap_pad(src: torch.Tensor, multiple: int = SEG_LEN) -> torch.Tensor:
    """np.pad(src, [(0,0),(0,pad_len)], 'wrap') of model.py:177-178: the utterance repeats itself to the next multiple."""
    T = src.size(-1)
    pad_len = math.ceil(T / multiple) * multiple - T
    if pad_len == 0:
        return src
    reps = math.ceil(pad_len / T)
    return torch.cat([src] + [src] * reps, dim=-1)[..., : T + pad_len]


def segment(src: torch.Tensor, normalise: bool) -> torch.Tensor:
    """src [1, T] -> seg_src [ceil(T / 80000), 80000] (model.py:176-182).  normalise = the 'se' branch: every segment is divided
    by the peak of the WHOLE utterance (`src.abs().max(dim=-1)`), the 'tse' branch does not normalise (model.py:199-203)."""
    if src.dim() != 2 or src.size(0) != 1:
        raise ValueError(f"src must be [1, T] like the reference's batch, got {tuple(src.shape)}")
    seg = wrap_pad(src).reshape(-1, SEG_LEN)
    if normalise:
        seg = seg / src.abs().max(dim=-1, keepdim=True)[0]
    return seg


def mel_frames(n_samples: int) -> int:
    """Number of frames `stft_logmel` (model.py:53-79) yields - the only property of the mel the LM consumes
    (llm_sft.py:108): the signal is padded to a multiple of the hop plus (win - hop) in total, center=False."""
    padded = math.ceil(n_samples / HOP_LENGTH) * HOP_LENGTH + (WIN_LENGTH - HOP_LENGTH)
    return (padded - WIN_LENGTH) // HOP_LENGTH + 1


class _Frames:
    """Stand-in for a mel tensor: LLM_SFT.generate only calls `.size(1)` on it."""

    def __init__(self, batch: int, frames: int):
        self._shape = (batch, frames, 80)

    def size(self, dim: int) -> int:
        return self._shape[dim]


def stft_logmel(x: torch.Tensor, hop_length: int = HOP_LENGTH, win_length: int = WIN_LENGTH, n_fft: int = 640, n_mels: int = 80) -> torch.Tensor:
    """Model.stft_logmel (model.py:53-79), restated with the filter bank of torchaudio.functional.melscale_fbanks (HTK scale, no
    normalisation, 0-8000 Hz).  NOT on the hot path: LLM_SFT.generate consumes only `mel.size(1)` (llm_sft.py:108), which
    `mel_frames()` gives without computing anything; this function exists for callers that want the reference's tensor."""
    assert x.ndim == 2
    pad_length = math.ceil(x.size(-1) / hop_length) * hop_length - x.size(-1)
    x = torch.nn.functional.pad(x, ((win_length - hop_length) // 2, pad_length + (win_length - hop_length) // 2))
    spec = torch.stft(x, n_fft, hop_length, win_length=win_length, window=torch.hann_window(win_length, device=x.device), onesided=True,
                      center=False, return_complex=True).transpose(1, 2)
    n_freqs = n_fft // 2 + 1
    all_freqs = torch.linspace(0, 16000 // 2, n_freqs)
    hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)  # noqa: E731
    m_pts = torch.linspace(hz2mel(0.0), hz2mel(8000.0), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down, up = -slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.minimum(down, up), min=0.0).to(x.device)
    return torch.log(spec.abs() @ fb + 1e-10)


class UniSE:
    def __init__(self, dnn, semantic_model, tokenizer=None, detokenize: Optional[Callable] = None, max_segments: int = 128):
        """dnn: unified_audio_amd.LLM_SFT; semantic_model: unified_audio_amd.SSLFeatureExtractor(SPEC_WAVLM_BASE_PLUS);
        tokenizer: unified_audio_amd.BiCodecTokenizer (or any object / callable with the reference's
        `detokenize(global_tokens [B, 1, 32], semantic_tokens [B, N]) -> wav [B, 1, t]`, model.py:193).
        max_segments: 5 s segments per pass through the three stages (the micro-batch).  The reference feeds one utterance per step
        (data_module.py:340); here all segments of a call are batched, 128 at a time - the measured optimum (profiles/r06_unise_micro_batch.txt,
        256 segments end to end: 773 audio-s/s at 64 per pass, 848 at 128, 848 at 256): the LM runs two chains of 64 sequences (112 k tok/s
        against 94 k with one chain and 44 k at 16 segments), four chains add nothing, memory stays bounded for long file lists, and - every
        stage being batch-invariant - the result does not depend on the value."""
        self.dnn = dnn
        self.semantic_model = semantic_model
        self.detokenize = detokenize if detokenize is not None else (tokenizer.detokenize if tokenizer is not None else None)
        self.max_segments = max(1, int(max_segments))

    def _generate(self, mode: str, seg_src: torch.Tensor, counts: Sequence[int], enroll_feats_per_utt: Optional[torch.Tensor],
                  enroll_samples: int):
        m = self.max_segments
        if seg_src.size(0) > m:  # micro-batches of <= max_segments segments; a segment's utterance is looked up through `owner`
            owner = torch.repeat_interleave(torch.arange(len(counts)), torch.tensor(list(counts)))
            gs, ss = [], []
            for a in range(0, seg_src.size(0), m):
                own = owner[a:a + m]
                utts, sub_counts = torch.unique_consecutive(own, return_counts=True)
                ef = enroll_feats_per_utt[utts.to(enroll_feats_per_utt.device)] if enroll_feats_per_utt is not None else None
                g, sids = self._generate(mode, seg_src[a:a + m], sub_counts.tolist(), ef, enroll_samples)
                gs.append(g)
                ss.append(sids)
            return torch.cat(gs, dim=0), torch.cat(ss, dim=0)
        mix_feats = self.semantic_model(seg_src)                                   # extract_semantic_features, model.py:38-51
        mix_mel = _Frames(seg_src.size(0), mel_frames(SEG_LEN))
        enroll_mel = enroll_feats = None
        if enroll_feats_per_utt is not None:
            # model.py:207-210: the utterance's enrollment is tiled over its segments
            enroll_feats = torch.cat([enroll_feats_per_utt[i:i + 1].expand(c, -1, -1) for i, c in enumerate(counts)], dim=0).contiguous()
            enroll_mel = _Frames(seg_src.size(0), mel_frames(enroll_samples))
        return self.dnn.generate(task_name=mode, enroll_mel=enroll_mel, enroll_feats=enroll_feats, mix_mel=mix_mel, mix_feats=mix_feats,
                                 do_sample=False)

    @staticmethod
    def _split(counts, *tensors):
        out, at = [], 0
        for c in counts:
            out.append(tuple(t[at:at + c] for t in tensors))
            at += c
        return out

    @torch.no_grad()
    def enhance_tokens(self, mode: str, srcs: Sequence[torch.Tensor], enrolls: Optional[Sequence[torch.Tensor]] = None
                       ) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        """srcs: utterances [1, T_i] (device tensors); enrolls (tse / rtse): one [1, T_e_i] per utterance (any lengths).
        Returns per utterance (global_ids [n_seg_i, 32], semantic_ids [n_seg_i, 250])."""
        if mode not in ("se", "tse", "rtse"):
            raise KeyError(mode)
        if mode != "se":
            if enrolls is None or len(enrolls) != len(srcs):
                raise ValueError(f"{mode} needs one enrollment per utterance")
            lens = [int(e.size(-1)) for e in enrolls]
            if len(set(lens)) > 1:
                # every utterance keeps ITS enrollment length (the reference feeds one file per step, model.py:197-219, and never trims
                # an enrollment to another file's): the prompt length of a generate call is common to its batch, so utterances are grouped
                # by enrollment length, one pass per group, and handed back in the caller's order
                out: List[Optional[Tuple[torch.Tensor, torch.Tensor]]] = [None] * len(srcs)
                for n in sorted(set(lens)):
                    idx = [i for i, v in enumerate(lens) if v == n]
                    for i, r in zip(idx, self.enhance_tokens(mode, [srcs[i] for i in idx], [enrolls[i] for i in idx])):
                        out[i] = r
                return out  # type: ignore[return-value]
        segs = [segment(s, normalise=(mode == "se")) for s in srcs]
        counts = [s.size(0) for s in segs]
        seg_src = torch.cat(segs, dim=0)
        ef, n_enr = None, 0
        if mode != "se":
            ef = self.semantic_model(torch.cat(list(enrolls), dim=0))             # [U, N_e, d]
            n_enr = enrolls[0].size(-1)
        global_ids, semantic_ids = self._generate(mode, seg_src, counts, ef, n_enr)
        return self._split(counts, global_ids, semantic_ids)

    def _wave(self, src: torch.Tensor, gids: torch.Tensor, sids: torch.Tensor) -> torch.Tensor:
        est = self.detokenize(gids.unsqueeze(1), sids).squeeze(1)                 # model.py:192
        return est.reshape(-1)[: src.size(-1)]

    @torch.no_grad()
    def enhance(self, mode: str, srcs: Sequence[torch.Tensor], enrolls: Optional[Sequence[torch.Tensor]] = None):
        """Model.test_step up to the waveform (model.py:170-290).  'se' / 'tse': one tensor [T_i] per utterance
        (`est.reshape(-1)[:src.size(-1)]`); 'ss': a pair (s1, s2) per utterance - SE on the first 5 s gives the enrollment, then TSE
        (speaker 1) and rTSE (speaker 2) over all segments (model.py:223-290)."""
        if self.detokenize is None:
            raise RuntimeError("UniSE.enhance needs a tokenizer (unified_audio_amd.BiCodecTokenizer) or a detokenize callable; "
                               "enhance_tokens returns the tokens")
        if mode == "ss":
            return self._separate(srcs)
        toks = self.enhance_tokens(mode, srcs, enrolls)
        return self._waves(srcs, toks)

    def _waves(self, srcs, toks):
        """model.py:192 for every utterance, with ONE detokenize call over all segments (segments are independent and the detokenizer
        is batch-invariant, so this equals the reference's per-utterance calls bit for bit)."""
        g = torch.cat([t[0] for t in toks], dim=0)
        s = torch.cat([t[1] for t in toks], dim=0)
        m = self.max_segments
        est = torch.cat([self.detokenize(g[a:a + m].unsqueeze(1), s[a:a + m]).squeeze(1) for a in range(0, g.size(0), m)], dim=0)
        out, at = [], 0
        for src, (gi, _) in zip(srcs, toks):
            out.append(est[at:at + gi.size(0)].reshape(-1)[: src.size(-1)])
            at += gi.size(0)
        return out

    @torch.no_grad()
    def enhance_pipelined(self, mode: str, srcs: Sequence[torch.Tensor], enrolls: Optional[Sequence[torch.Tensor]] = None,
                          segments_per_batch: int = 16, lm_graph: bool = False):
        """The same results as `enhance` ('se' / 'tse' / 'rtse'), produced by a THREE-STAGE PIPELINE over micro-batches of
        `segments_per_batch` 5 s segments on three streams: WavLM features of batch k + 1, LLM_SFT.generate of batch k and
        BiCodec.detokenize of batch k - 1 run concurrently.  The decode loop of the LM is bound by the latency of its dependent
        launches and leaves the matrix cores idle; the two GEMM-bound stages fill them.  Every stage owns its
        handle (workspace), so the three streams never share library state; tensors that cross streams are recorded on the consuming
        stream.  `lm_graph`: the LM replays one captured hipGraph per token (QA_LM_GRAPH) instead of issuing its ~17 000 launches.
        Segments are independent and every stage is batch-invariant, so the output is bit-identical to `enhance`
        (tests/test_unise_driver_gpu.py).
        MEASURED (profiles/r03_unise_pipeline_ab.txt, 6 x 16 segments): 176.8 ms per batch against 181.2 ms for `enhance` batch by
        batch - the overlap is almost nil: the workgroups of the GEMM stages (30 - 300 us each) hold the wave slots of every CU, so
        the 5 us launches of the LM queue behind them exactly as the LSTM step launches do, and the work adds
        up instead of overlapping; with replayed LM graphs 194.8 ms.  Throughput comes from the batch size instead: 64 segments per
        call run at 775 audio-s/s end to end against 463 at 16."""
        from . import _lib

        if self.detokenize is None:
            raise RuntimeError("UniSE.enhance_pipelined needs a tokenizer (unified_audio_amd.BiCodecTokenizer) or a detokenize callable")
        if mode not in ("se", "tse", "rtse"):
            raise KeyError(mode)
        segs = [segment(s, normalise=(mode == "se")) for s in srcs]
        counts = [s.size(0) for s in segs]
        seg_src = torch.cat(segs, dim=0)
        dev = seg_src.device
        n_seg, m = seg_src.size(0), max(1, int(segments_per_batch))
        owner = torch.repeat_interleave(torch.arange(len(srcs)), torch.tensor(counts)).tolist()  # utterance of every segment
        cur = torch.cuda.current_stream(dev)
        s_feat, s_lm, s_dec = (torch.cuda.Stream(dev) for _ in range(3))
        for st in (s_feat, s_lm, s_dec):
            st.wait_stream(cur)
        ef, n_enr = None, 0
        if mode != "se":
            if enrolls is None or len(enrolls) != len(srcs):
                raise ValueError(f"{mode} needs one enrollment per utterance")
            if len({e.size(-1) for e in enrolls}) != 1:
                raise ValueError("enrollments of one call must have the same length")
            with torch.cuda.stream(s_feat):
                ef = self.semantic_model(torch.cat(list(enrolls), dim=0))
            n_enr = enrolls[0].size(-1)
        nb = (n_seg + m - 1) // m
        feats, toks, wavs = [None] * nb, [None] * nb, [None] * nb  # alive until the final join: nothing is freed under a stream
        ev_f = [torch.cuda.Event() for _ in range(nb)]
        ev_l = [torch.cuda.Event() for _ in range(nb)]
        old_graph = _lib.set_knob("QA_LM_GRAPH", 1) if lm_graph else None
        # the detokenizer's range check of its inputs is a host synchronisation per call: the tokens here are the LM's own (offsets
        # subtracted inside the active vocabulary slices, in range by construction), so the check is switched off inside the pipeline
        bic = getattr(getattr(self.detokenize, "__self__", None), "model", None)
        old_check = getattr(bic, "check_tokens", None)
        if old_check:
            bic.check_tokens = False
        try:
            for k in range(nb + 2):
                if k < nb:  # stage 1: features of micro-batch k
                    with torch.cuda.stream(s_feat):
                        x = seg_src[k * m:(k + 1) * m]
                        f = self.semantic_model(x)
                        e = None
                        if ef is not None:
                            e = ef[torch.tensor(owner[k * m:(k + 1) * m], device=dev)].contiguous()
                            e.record_stream(s_lm)
                        f.record_stream(s_lm)
                        feats[k] = (f, e, x.size(0))
                        ev_f[k].record(s_feat)
                j = k - 2
                if 0 <= j < nb:  # stage 3: waveform of micro-batch k - 2 (enqueued BEFORE the long LM enqueue of this turn)
                    with torch.cuda.stream(s_dec):
                        s_dec.wait_event(ev_l[j])
                        g, s = toks[j]
                        w = self.detokenize(g.unsqueeze(1), s).squeeze(1)
                        w.record_stream(cur)
                        wavs[j] = w
                j = k - 1
                if 0 <= j < nb:  # stage 2: tokens of micro-batch k - 1
                    with torch.cuda.stream(s_lm):
                        s_lm.wait_event(ev_f[j])
                        f, e, b = feats[j]
                        g, s = self.dnn.generate(task_name=mode, enroll_mel=None if e is None else _Frames(b, mel_frames(n_enr)), enroll_feats=e,
                                                 mix_mel=_Frames(b, mel_frames(SEG_LEN)), mix_feats=f, do_sample=False)
                        g.record_stream(s_dec)
                        s.record_stream(s_dec)
                        toks[j] = (g, s)
                        ev_l[j].record(s_lm)
        finally:
            if old_graph is not None:
                _lib.set_knob("QA_LM_GRAPH", old_graph)
            if old_check:
                bic.check_tokens = old_check
        cur.wait_stream(s_dec)
        cur.wait_stream(s_lm)
        cur.wait_stream(s_feat)
        est = torch.cat(wavs, dim=0)
        out, at = [], 0
        for src, c in zip(srcs, counts):
            out.append(est[at:at + c].reshape(-1)[: src.size(-1)])
            at += c
        return out

    def _separate(self, srcs: Sequence[torch.Tensor]):
        # pass 1 (model.py:224-242): SE on the first 5 s of every mixture (wrap-padded if shorter; no peak normalisation in this mode)
        heads = []
        for s in srcs:
            if s.dim() != 2 or s.size(0) != 1:
                raise ValueError(f"src must be [1, T] like the reference's batch, got {tuple(s.shape)}")
            heads.append(s[:, :SEG_LEN] if s.size(-1) > SEG_LEN else wrap_pad(s))
        head = torch.cat(heads, dim=0)
        g0, s0 = self._generate("se", head, [1] * len(srcs), None, 0)
        enroll = self.detokenize(g0.unsqueeze(1), s0).squeeze(1)[:, :SEG_LEN]                      # (U, t)
        enroll = enroll / (enroll.abs().amax(dim=-1, keepdim=True) + 1e-5) * 0.99                  # model.py:243 (per mixture)
        ef = self.semantic_model(enroll)
        # passes 2 and 3 (model.py:249-288): TSE then rTSE over all (un-normalised) segments with the tiled enrollment
        segs = [segment(s, normalise=False) for s in srcs]
        counts = [s.size(0) for s in segs]
        seg_src = torch.cat(segs, dim=0)
        out = []
        tse = self._split(counts, *self._generate("tse", seg_src, counts, ef, SEG_LEN))
        rtse = self._split(counts, *self._generate("rtse", seg_src, counts, ef, SEG_LEN))
        w1, w2 = self._waves(srcs, tse), self._waves(srcs, rtse)
        return list(zip(w1, w2))


class TestDataset:
    """The reference's inference data iterator (QuarkAudio-UniSE/dataloader/data_module.py:296-410, built from
    `config['dataset_config']['test_kwargs']`): every `*.flac` / `*.wav` of `data_src_dir` gives one batch tuple
    `(mode, enroll, src, tgt, fs, lengths, names)` with `batch_size == 1` (:340), first channel only, 16 kHz; the enrollment is wrapped /
    cut to `enroll_duration` seconds and scaled to a peak of 0.99 (:346-352).  Differences, both stated in INTEGRATION.md: wav files only
    (soundfile / FLAC are not available offline - a FLAC file raises), and a file at another rate is resampled on the device with
    qa_resample (torchaudio's sinc kernel) where the reference uses librosa's soxr_hq.  Sharding over ranks follows :364 (rank-strided)."""

    __test__ = False  # not a pytest class

    def __init__(self, data_enroll_dir, data_src_dir, data_tgt_dir, mode: str, enroll_duration: float = 5.0, batch_size: int = 1,
                 num_workers: int = 1, prefetch: int = 0, *, device="cuda:0", rank: int = 0, world_size: int = 1):
        import pathlib

        if batch_size != 1:
            raise AssertionError("batch_size == 1 (data_module.py:340); UniSE.enhance batches utterances itself")
        self.mode, self.enroll_duration = mode, float(enroll_duration)
        self.data_enroll_dir = pathlib.Path(data_enroll_dir) if data_enroll_dir is not None else None
        self.data_src_dir, self.data_tgt_dir = pathlib.Path(data_src_dir), pathlib.Path(data_tgt_dir)
        self.wav_names = [p.name for p in self.data_src_dir.glob("*.flac")] + [p.name for p in self.data_src_dir.glob("*.wav")]
        self.device, self.rank, self.world_size = torch.device(device), rank, world_size

    def load_wav(self, path):
        from . import audio_io

        if str(path).lower().endswith(".flac"):
            raise ValueError(f"{path}: FLAC needs soundfile, which is not available here - convert to wav")
        return audio_io.load_audio(str(path), 16000, self.device)

    def process_one_sample(self, name):
        import pathlib

        src = self.load_wav(self.data_src_dir / name)
        tgt = self.load_wav(self.data_tgt_dir / name)
        enroll = None
        if self.data_enroll_dir is not None:
            enroll = self.load_wav(self.data_enroll_dir / name)
            length = int(self.enroll_duration * 16000)
            enroll = wrap_pad(enroll, length)[..., :length] if enroll.shape[-1] < length else enroll[..., :length]
            enroll = enroll / (enroll.abs().max() + 1e-5) * 0.99
        return enroll, src, tgt, 16000, src.shape[-1], pathlib.Path(name).stem

    def __len__(self):
        return len(range(self.rank, len(self.wav_names), self.world_size))

    def __iter__(self):
        for i in range(self.rank, len(self.wav_names), self.world_size):
            enroll, src, tgt, fs, length, name = self.process_one_sample(self.wav_names[i])
            yield (self.mode, enroll, src, tgt, torch.tensor([fs], dtype=torch.int64), torch.tensor([length], dtype=torch.int64), [name])


class Model:
    """The test path of the reference's `Model` (QuarkAudio-UniSE/model/model.py:20-36 constructor, :82-91 state-dict rules, :170-290
    `test_step`), as `test.py:11-30` drives it: `Model(config)`, the Lightning checkpoint `config['ckpt_path']`, then one
    `test_step(batch, batch_idx)` per file with `batch = (mode, enroll, src, tgt, fs, lengths, names)`, writing
    `config['save_enhanced']/{name}.wav` ('se', 'tse') or `{name}_s1.wav` / `{name}_s2.wav` ('ss').

    config keys read, as in the reference: `codec_ckpt_dir` (-> BiCodecTokenizer(model_dir=...): `BiCodec/config.yaml` +
    `BiCodec/model.safetensors`), `llm_config` (-> LLM_SFT(**...)), `stft_config`, `save_enhanced`, `ckpt_path`.  One key is added:
    `semantic_model_path` - a local snapshot of microsoft/wavlm-base-plus (the reference downloads it, model.py:30; there is no
    network here); `semantic_model=` passes a loaded SSLFeatureExtractor instead.
    `test_steps(batches)` is the batched form: any number of the same tuples in one pass (segments of all files share the launches).
    """

    def __init__(self, config, *, device: str | torch.device = "cuda:0", semantic_model=None, tokenizer=None, dnn=None, max_segments: int = 128):
        from .bicodec import BiCodecTokenizer
        from .llm import LLM_SFT
        from .ssl import SPEC_WAVLM_BASE_PLUS, SSLFeatureExtractor

        self.config = config
        self.stft_conf = dict(config.get("stft_config") or dict(hop_length=HOP_LENGTH, win_length=WIN_LENGTH, n_fft=640, n_mels=80))
        if (self.stft_conf["hop_length"], self.stft_conf["win_length"]) != (HOP_LENGTH, WIN_LENGTH):
            raise ValueError(f"stft_config {self.stft_conf}: the LM's step count follows hop 320 / win 640 (conf/config.yaml:124-128)")
        self.device = torch.device(device)
        self.tokenizer = tokenizer if tokenizer is not None else BiCodecTokenizer(model_dir=config["codec_ckpt_dir"], device=self.device)
        self.dnn = dnn if dnn is not None else LLM_SFT(**config["llm_config"], device=self.device)
        if semantic_model is None:
            path = config.get("semantic_model_path")
            if path is None:
                raise ValueError("Model needs config['semantic_model_path'] (a local microsoft/wavlm-base-plus snapshot) or semantic_model=: "
                                 "the reference downloads the model (model.py:30), this machine has no network")
            semantic_model = SSLFeatureExtractor.from_pretrained(path, device=self.device, default_spec=SPEC_WAVLM_BASE_PLUS)
        self.semantic_model = semantic_model
        self.driver = UniSE(self.dnn, self.semantic_model, tokenizer=self.tokenizer, max_segments=max_segments)
        if config.get("ckpt_path") and dnn is None:
            import os

            if os.path.isfile(str(config["ckpt_path"])):  # test.py:30 hands it to trainer.test(..., ckpt_path=)
                self.load_checkpoint(config["ckpt_path"])

    def load_state_dict(self, state_dict, strict: bool = True):
        """model.py:82-91: the checkpoint holds `dnn.*` only (tokenizer / semantic_model are excluded on save and loading is non-strict)."""
        sd = {k: v for k, v in state_dict.items() if k.startswith("dnn.")}
        if not sd:
            raise KeyError("no 'dnn.*' entries: not a UniSE checkpoint (model.py:82-91)")
        self.dnn.load_state_dict(sd)
        return self

    def load_checkpoint(self, ckpt_path):
        """A Lightning checkpoint `{'state_dict': {'dnn.*': ...}, ...}` (what `trainer.test(model, dm, ckpt_path=...)` restores, test.py:30)."""
        ck = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        return self.load_state_dict(ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck)

    def eval(self):
        return self

    def to(self, *args, **kwargs):
        return self

    def extract_semantic_features(self, wavs: torch.Tensor) -> torch.Tensor:  # model.py:38-51
        return self.semantic_model(wavs.to(self.device))

    def stft_logmel(self, x: torch.Tensor) -> torch.Tensor:  # model.py:53-79
        c = self.stft_conf
        return stft_logmel(x, c["hop_length"], c["win_length"], c["n_fft"], c["n_mels"])

    def _save(self, name: str, est: torch.Tensor, fs: int):
        import os

        from . import audio_io

        out = self.config.get("save_enhanced") if hasattr(self.config, "get") else None
        if out is not None:  # model.py:195-196 (the directory is made by test.py:17)
            audio_io.write_wav(os.path.join(str(out), f"{name}.wav"), est, int(fs))

    @staticmethod
    def _unpack(batch):
        mode, enroll, src, tgt, fs, lengths, names = batch
        if src.dim() != 2 or src.size(0) != 1:
            raise ValueError(f"src must be [1, T]: the reference's test batches hold one file (data_module.py:340), got {tuple(src.shape)}")
        return mode, enroll, src, int(fs[0]), names[0]

    @torch.no_grad()
    def test_step(self, batch, batch_idx: int = 0):
        """model.py:170-290.  Returns what it wrote: the estimate [T] ('se' / 'tse'), the pair ('ss'), None for any other mode (the
        reference's if / elif chain falls through silently)."""
        res = self.test_steps([batch])
        return res[0] if res else None

    @torch.no_grad()
    def test_steps(self, batches):
        """Any number of test batches in ONE pass per mode: all 5 s segments of all files go through WavLM, the LM and BiCodec together
        (every stage is batch-invariant, so each file's result equals its own test_step bit for bit)."""
        items = [self._unpack(b) for b in batches]
        out = [None] * len(items)
        for mode in ("se", "tse", "ss"):
            idx = [i for i, it in enumerate(items) if it[0] == mode]
            if not idx:
                continue
            srcs = [items[i][2].to(self.device, torch.float32) for i in idx]
            enrolls = None
            if mode == "tse":
                if any(items[i][1] is None for i in idx):
                    raise ValueError("'tse' needs an enrollment in every batch (data_enroll_dir)")
                enrolls = [items[i][1].to(self.device, torch.float32) for i in idx]
            for i, est in zip(idx, self.driver.enhance(mode, srcs, enrolls)):
                _, _, _, fs, name = items[i]
                if mode == "ss":
                    self._save(f"{name}_s1", est[0], fs)
                    self._save(f"{name}_s2", est[1], fs)
                else:
                    self._save(name, est, fs)
                out[i] = est
        return out
