from . import layers, quantize  # noqa: F401
