"""wav read / write helpers of the inference entry points (stand-ins for soundfile / librosa, which are absent offline)."""
import io
import struct
import wave

import numpy as np
import torch

from unified_audio_amd import audio_io as A


def test_pcm16_roundtrip_and_stdlib_compatibility(tmp_path):
    x = (torch.rand(1, 4001, generator=torch.Generator().manual_seed(0)) * 2 - 1) * 0.9
    p = str(tmp_path / "a.wav")
    A.write_wav(p, x, 16000)
    with wave.open(p, "rb") as w:  # the standard library reads what we wrote
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 16000, 4001)
    y, sr = A.read_wav(p)
    assert sr == 16000 and y.shape == (1, 4001) and float((y - x).abs().max()) <= 0.5 / 32768 + 1e-7


def test_float_and_multichannel_and_24bit(tmp_path):
    x = torch.linspace(-1, 1, 1000)
    p = str(tmp_path / "f.wav")
    A.write_wav(p, x, 48000, subtype="FLOAT")
    y, sr = A.read_wav(p)
    assert sr == 48000 and torch.equal(y[0], x)
    # stereo PCM16 written by the standard library: channel 0 is taken (audio_tokenizer.py: wav[:1, :])
    st = np.stack([np.arange(100), -np.arange(100)], axis=1).astype("<i2")
    q = str(tmp_path / "s.wav")
    with wave.open(q, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(8000); w.writeframes(st.tobytes())
    y, sr = A.read_wav(q)
    assert sr == 8000 and torch.equal(y[0], torch.arange(100).float() / 32768.0)
    # 24-bit PCM
    vals = np.array([0, 1, -1, 8388607, -8388608], dtype=np.int32)
    body = b"".join(struct.pack("<i", int(v))[:3] for v in vals)
    r = str(tmp_path / "t.wav")
    with open(r, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 48000, 3, 24) + b"data" + struct.pack("<I", len(body)) + body)
    y, _ = A.read_wav(r)
    assert np.allclose(y[0].numpy(), vals / 8388608.0)


def test_interoperates_with_scipy_wavfile(tmp_path):
    """A second independent reader / writer (scipy.io.wavfile): float32 and PCM16 files cross both ways, PCM32 and extensible-format
    headers written by scipy are read."""
    from scipy.io import wavfile

    g = np.random.default_rng(3)
    x = (g.uniform(-0.9, 0.9, size=3001)).astype(np.float32)
    p = str(tmp_path / "ours_float.wav")
    A.write_wav(p, torch.from_numpy(x), 16000, subtype="FLOAT")
    sr, y = wavfile.read(p)
    assert sr == 16000 and y.dtype == np.float32 and np.array_equal(y, x)
    p = str(tmp_path / "ours_pcm16.wav")
    A.write_wav(p, torch.from_numpy(x), 22050)
    sr, y = wavfile.read(p)
    assert sr == 22050 and y.dtype == np.int16 and np.abs(y / 32768.0 - x).max() <= 0.5 / 32768 + 1e-7
    for dt, scale in ((np.float32, 1.0), (np.int16, 32768.0), (np.int32, 2147483648.0)):
        q = str(tmp_path / f"scipy_{np.dtype(dt).name}.wav")
        data = x if dt is np.float32 else np.round(x * (scale - 1)).astype(dt)
        wavfile.write(q, 48000, np.stack([data, data[::-1]], axis=1))  # stereo: channel 0 is taken
        y, sr = A.read_wav(q)
        assert sr == 48000 and y.shape == (1, 3001)
        assert np.allclose(y[0].numpy(), data.astype(np.float64) / scale, atol=1e-7)
