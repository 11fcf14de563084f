"""TEST INFRASTRUCTURE ONLY - import-time stand-in for torchaudio (absent here)."""
from . import functional, transforms  # noqa: F401
