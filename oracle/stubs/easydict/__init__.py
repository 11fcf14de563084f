class EasyDict(dict):
    """import-time stand-in (HCodec-1.5/adaptive/modeling_flexicodec_new.py:13)."""

    __getattr__ = dict.get
