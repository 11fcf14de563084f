"""TEST INFRASTRUCTURE ONLY - import-time stand-in for `x_transformers` (absent here).  The reference's
QuarkAudio-UniSE/model/llm/conformer.py:17 imports two names from it for the conformer condition encoder, which
LLM_SFT constructs but never calls on the generate path ("constructed but never called")."""
