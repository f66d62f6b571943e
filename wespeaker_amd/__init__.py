"""wespeaker_amd: MI355X-native speaker-embedding extraction + PLDA scoring behind WeSpeaker's
Python API (`load_model`, `Speaker.extract_embedding*`, `TwoCovPLDA`, `score_plda`).

The hot path (Kaldi fbank, ECAPA-TDNN forward, two-covariance PLDA LLR) is hand-written HIP for
gfx950 behind the C-ABI of include/wespeaker_amd.h; this package is the thin host mirror of the
reference's interface.  Importing the package never touches the GPU; the first call that needs
the native library raises if it (or a GPU) is missing -- there is no CPU fallback.
"""
from .speaker import Speaker, load_model, load_model_pt  # noqa: F401
from .plda import TwoCovPLDA, score_plda  # noqa: F401
from .engine import Frontend, NativeSpeakerModel, SpeakerModelLanes  # noqa: F401
from . import score  # noqa: F401  (bin/score.py + bin/score_norm.py on the GPU)

__version__ = "0.1.0"
