"""TEST INFRASTRUCTURE ONLY - import-time stand-in for descript-audio-codec (HCodec-1.5/adaptive/*.py import it; nothing
on the Codec.encode/decode path constructs these classes)."""
from . import nn  # noqa: F401
