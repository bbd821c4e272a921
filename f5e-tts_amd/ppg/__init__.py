"""PPG extractor on libf5e_hip.so (reference src/f5_tts/ppg/ppg_model.py, asr_model.py, wenet/)."""
from .ppg_model import ConformerPPG, PPGModelWapper, build_ppg_model, kaldiFbank, load_cmvn  # noqa: F401
