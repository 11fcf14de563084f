"""TEST INFRASTRUCTURE - re-export of the seeded synthetic weight / input generators.

The generators are pure data generation and live in unified_audio_amd/synth.py so that bench.py can build its inputs without
importing anything under oracle/ (only tests/, smoke() and bench.py's cpu_baseline leg import this package)."""
from unified_audio_amd.synth import (  # noqa: F401
    hcodec10_state_dict, hcodec20_state_dict, mimi_state_dict, stress_lm_state_dict, stress_state_dict, synth_feat, synth_wav, synth_wav_fullband, _Gen, _t)
