"""Segmenting / batching driver of the UniSE inference step, in front of the HIP paths of this package.

    UniSE.enhance_tokens(mode, src, enroll)  <->  Model.test_step, 'se' and 'tse' branches
                                                  (QuarkAudio-UniSE/model/model.py:170-222): wrap-pad to a multiple of 5 s,
                                                  cut into 5 s segments, (se: divide by the utterance peak), WavLM
                                                  features, LLM_SFT.generate -> (global_ids, semantic_ids)

The reference handles ONE utterance per call (`batch_size == 1`, dataloader/data_module.py:340); here any number of
utterances is accepted: their segments are concatenated into one batch for the front-end and the LM (segments are
independent), and the tokens are handed back per utterance.  The last stage of `test_step` - BiCodec `detokenize` - is not
part of this package: pass a callable `detokenize(global_ids[B,1,32], semantic_ids[B,N]) -> wav[B,1,t]` to
get waveforms, otherwise tokens are returned.
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


class UniSE:
    def __init__(self, dnn, semantic_model, detokenize: Optional[Callable] = None):
        """dnn: unified_audio_amd.LLM_SFT; semantic_model: unified_audio_amd.SSLFeatureExtractor(SPEC_WAVLM_BASE_PLUS)."""
        self.dnn = dnn
        self.semantic_model = semantic_model
        self.detokenize = detokenize

    @torch.no_grad()
    def enhance_tokens(self, mode: str, srcs: Sequence[torch.Tensor], enrolls: Optional[Sequence[torch.Tensor]] = None
                       ) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        """srcs: utterances [1, T_i] (device tensors); enrolls (tse): one [1, T_e] per utterance, all of the same length.
        Returns per utterance (global_ids [n_seg_i, 32], semantic_ids [n_seg_i, 250])."""
        if mode not in ("se", "tse"):
            raise KeyError(mode)
        segs = [segment(s, normalise=(mode == "se")) for s in srcs]
        counts = [s.size(0) for s in segs]
        seg_src = torch.cat(segs, dim=0)
        mix_feats = self.semantic_model(seg_src)                                   # extract_semantic_features, model.py:38-51
        mix_mel = _Frames(seg_src.size(0), mel_frames(SEG_LEN))
        enroll_mel = enroll_feats = None
        if mode == "tse":
            if enrolls is None or len(enrolls) != len(srcs):
                raise ValueError("tse needs one enrollment per utterance")
            if len({e.size(-1) for e in enrolls}) != 1:
                raise ValueError("enrollments of one call must have the same length")
            ef = self.semantic_model(torch.cat(list(enrolls), dim=0))             # [U, N_e, d]
            # model.py:207-210: the utterance's enrollment is tiled over its segments
            enroll_feats = torch.cat([ef[i:i + 1].expand(c, -1, -1) for i, c in enumerate(counts)], dim=0).contiguous()
            enroll_mel = _Frames(seg_src.size(0), mel_frames(enrolls[0].size(-1)))
        global_ids, semantic_ids = self.dnn.generate(task_name=mode, enroll_mel=enroll_mel, enroll_feats=enroll_feats, mix_mel=mix_mel,
                                                     mix_feats=mix_feats, do_sample=False)
        out, at = [], 0
        for c in counts:
            out.append((global_ids[at:at + c], semantic_ids[at:at + c]))
            at += c
        return out

    @torch.no_grad()
    def enhance(self, mode: str, srcs: Sequence[torch.Tensor], enrolls: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
        """test_step up to `est.reshape(-1)[:src.size(-1)]` (model.py:192-193); needs the `detokenize` callable."""
        if self.detokenize is None:
            raise RuntimeError("UniSE.enhance needs a detokenize callable (BiCodec is not part of this package); use enhance_tokens")
        outs = []
        for src, (gids, sids) in zip(srcs, self.enhance_tokens(mode, srcs, enrolls)):
            est = self.detokenize(gids.unsqueeze(1), sids).squeeze(1)
            outs.append(est.reshape(-1)[: src.size(-1)])
        return outs
