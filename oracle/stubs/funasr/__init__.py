class AutoModel:
    """import-time stand-in (HCodec-1.5/adaptive/modeling_flexicodec_new.py:19)."""

    def __init__(self, *a, **k):
        raise NotImplementedError
