"""Drop-in `meant` package: put <repo>/dropin on PYTHONPATH (ahead of the reference checkout) and the
reference's drivers (`from meant import meant, meant_vision, meant_tweet, temporal, ...`,
in_loop_train.py:27-29) pick up the MI355X-native modules.  Mirrors the hot-path names of
meant/__init__.py:1-11; whole-module pickles saved by the reference (class path `meant.meant.meant`)
resolve here too.  As in the reference, the from-imports below rebind the package attributes
`meant`, `attention`, ... from the submodules to the classes.
"""
from .attention import attention  # noqa: F401
from .meant import meant, languageEncoder, visionEncoder, temporalEncoder  # noqa: F401
from .meant_vision import meant_vision  # noqa: F401
from .meant_tweet import meant_tweet  # noqa: F401
from .meant_vqa import meant_vqa  # noqa: F401
from .xPosAttention import xPosAttention  # noqa: F401
from .temporal import temporal  # noqa: F401
from meant_amd.modules import xPosAttention_flash, flash_attention, RMSNorm  # noqa: F401
from .hf_wrapper import meant_language_pretrainer, meant_vision_pretrainer  # noqa: F401
