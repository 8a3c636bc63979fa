"""`meant.meant` module path of the reference (meant/meant.py) -> native classes."""
from meant_amd.modules import meant, visionEncoder, languageEncoder, temporalEncoder  # noqa: F401
