"""Imported by HCodec-*/vq/heads.py:88 (of the original file), never called on the inference path."""


def _hz_to_mel(*a, **k):
    raise NotImplementedError


def _mel_to_hz(*a, **k):
    raise NotImplementedError
