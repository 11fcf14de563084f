"""Minimal wav file I/O for the inference entry points (the reference uses soundfile / librosa: `sf.write(... samplerate=fs)` at
QuarkAudio-UniSE/model/model.py:196, `librosa.load(..., sr=16000, mono=False)` in the tokenizers' __main__).  Neither package is
available offline; RIFF/WAVE PCM (16 / 24 / 32 bit) and IEEE float32 are handled with the standard library + numpy, channel 0 is
taken like the reference does (`wav[:1, :]`), and resampling runs on the device through qa_resample (torchaudio's sinc kernel)."""
from __future__ import annotations

import struct
from typing import Tuple

import numpy as np
import torch


def read_wav(path: str) -> Tuple[torch.Tensor, int]:
    """-> (float32 tensor [1, T] in [-1, 1), sample_rate); first channel only."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, body = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        chunk = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", chunk[:16])
            if fmt[0] == 0xFFFE and len(chunk) >= 26:  # WAVE_FORMAT_EXTENSIBLE: the real tag is the first word of the sub-format GUID
                fmt = (struct.unpack("<H", chunk[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            body = chunk
        pos += 8 + size + (size & 1)
    if fmt is None or body is None:
        raise ValueError(f"{path}: missing fmt / data chunk")
    tag, ch, sr, _, _, bits = fmt
    if tag == 3 and bits == 32:
        x = np.frombuffer(body, dtype="<f4").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(body, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(body, dtype="<i4").astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(body[: len(body) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (np.where(v >= 1 << 23, v - (1 << 24), v)).astype(np.float32) / 8388608.0
    else:
        raise ValueError(f"{path}: unsupported wav format tag {tag} with {bits} bits")
    x = x[: len(x) // ch * ch].reshape(-1, ch)[:, 0]
    return torch.from_numpy(np.ascontiguousarray(x))[None], int(sr)


def write_wav(path: str, wav, sample_rate: int, subtype: str = "PCM_16") -> None:
    """wav: 1-D (or [1, T]) float array / tensor in [-1, 1]; PCM_16 like soundfile's default for .wav, or FLOAT."""
    x = wav.detach().cpu().numpy() if torch.is_tensor(wav) else np.asarray(wav)
    x = x.reshape(-1).astype(np.float32)
    if subtype == "FLOAT":
        body, tag, bits = x.astype("<f4").tobytes(), 3, 32
    else:
        body, tag, bits = (np.clip(np.rint(x * 32768.0), -32768, 32767)).astype("<i2").tobytes(), 1, 16
    block = bits // 8
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, tag, 1, sample_rate, sample_rate * block, block, bits)
    with open(path, "wb") as f:
        f.write(hdr + b"data" + struct.pack("<I", len(body)) + body)


def load_audio(path: str, target_sr: int = 16000, device="cuda:0") -> torch.Tensor:
    """read_wav + resampling to `target_sr` on the device (qa_resample) -> float32 [1, T] on `device`."""
    import ctypes as C  # noqa: F401

    from . import _lib

    wav, sr = read_wav(path)
    wav = wav.to(device)
    if sr == target_sr:
        return wav
    lib = _lib.load_library()
    n = int(lib.qa_resample_length(wav.shape[-1], sr, target_sr))
    out = torch.empty((1, n), dtype=torch.float32, device=wav.device)
    _lib.check(lib.qa_resample(wav.data_ptr(), 1, wav.shape[-1], sr, target_sr, out.data_ptr(), torch.cuda.current_stream(wav.device).cuda_stream))
    return out
