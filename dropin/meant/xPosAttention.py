"""`meant.xPosAttention` module path of the reference -> native classes."""
from meant_amd.modules import *  # noqa: F401,F403
from meant_amd.modules import xPosAttention  # noqa: F401
