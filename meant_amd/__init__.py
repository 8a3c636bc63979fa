"""meant_amd -- MI355X-native (gfx950 / CDNA4) implementation of the MEANT encoder hot path.

Importing this package loads meant_amd/libmeant_hip.so (hand-written HIP kernels behind the C ABI
of include/meant_hip.h) and raises if it has not been built: there is one backend, no fallback.
"""
from ._lib import lib, LIB_PATH, MeantHipError, MeantLibraryMissing  # noqa: F401
from . import ops  # noqa: F401
from .modules import (  # noqa: F401
    RMSNorm, LayerNorm, Linear, RotaryEmbedding,
    attention, xPosAttention, temporal, flash_attention, xPosAttention_flash,
    visionEncoder, languageEncoder, temporalEncoder,
    meant, meant_vision, meant_tweet, meant_vqa, meant_language_pretrainer, meant_vision_pretrainer, TimeSformer,
)

from . import parallel, train, data  # noqa: F401,E402

__version__ = "0.1.0"
