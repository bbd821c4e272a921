"""f5e_tts_amd -- MI355X-native flow-matching inference hot path of F5E-TTS.

Only what the path needs: the C-ABI HIP library (``csrc/`` -> ``libf5e_hip.so``, bound in ``_C.py``), the engine that
plans workspaces / repacks weights / drives the ODE loop (``engine.py``), and host-side mirrors of the reference's
module API (``model/``, ``infer/``, ``eval/``).  There is NO CPU or eager-PyTorch fallback: every compute entry point
raises if the HIP library is missing or the tensors are not on a gfx950 device.
"""
__version__ = "0.1.0"
