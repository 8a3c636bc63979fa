"""`meant.meant_tweet` module path of the reference -> native classes."""
from meant_amd.modules import *  # noqa: F401,F403
from meant_amd.modules import meant_tweet  # noqa: F401
