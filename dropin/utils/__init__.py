"""Drop-in `utils` package exporting RMSNorm (reference: utils/__init__.py, `from utils import RMSNorm`
at meant/meant.py:13)."""
from meant_amd.modules import RMSNorm  # noqa: F401
