"""`utils.rms_norm` module path of the reference -> native RMSNorm."""
from meant_amd.modules import RMSNorm  # noqa: F401
