"""`meant.meant_vqa` module path of the reference -> native classes."""
from meant_amd.modules import *  # noqa: F401,F403
from meant_amd.modules import meant_vqa  # noqa: F401
