"""unified_audio_amd - MI355X-native (gfx950) hot path of alibaba/unified-audio (QuarkAudio).

Only the path BASELINE.json's north_star names lives here: H-Codec encode -> RVQ -> decode and the UniSE AR-LM generate
loop, as hand-written HIP kernels behind the C-ABI in include/quarkaudio.h, plus the thin Python mirror of the
reference's own interface (`Codec.encode/decode`, `HCodecTokenizer.tokenize/detokenize`, `LLM_SFT.generate`), and - the first
"next" row of SURVEY.md section 8f - the SSL feature extraction in front of `Codec.encode` (`SSLFeatureExtractor`).
There is no CPU / PyTorch fallback: if libquarkaudio_hip.so is missing or no gfx950 device is visible, calls raise.
"""
from ._lib import QuarkAudioError, lib_path, load_library  # noqa: F401
from .hcodec import Codec, HCodecSpec, HCodecTokenizer, SPEC_10, SPEC_15, SPEC_20  # noqa: F401
from .llm import LLM_SFT, sample_logits  # noqa: F401
from .bicodec import BiCodec, BiCodecSpec, BiCodecTokenizer, SPEC_BICODEC  # noqa: F401
from .mimi import StreamingTransformer  # noqa: F401
from .unise import UniSE  # noqa: F401
from .ssl import SPEC_HUBERT_BASE, SPEC_WAVLM_BASE_PLUS, SPEC_XLSR53, SSLFeatureExtractor, SSLSpec  # noqa: F401

__all__ = ["StreamingTransformer", "BiCodecTokenizer", "BiCodec", "BiCodecSpec", "SPEC_BICODEC", "sample_logits", "UniSE", "SSLFeatureExtractor", "SSLSpec", "SPEC_HUBERT_BASE", "SPEC_XLSR53", "SPEC_WAVLM_BASE_PLUS", "LLM_SFT", "Codec", "HCodecSpec", "HCodecTokenizer", "SPEC_10", "SPEC_15", "SPEC_20", "QuarkAudioError", "load_library", "lib_path"]
