"""`from meant.hf_wrapper import meant_language_pretrainer, meant_vision_pretrainer` (meant/__init__.py:11,
meant/hf_wrapper.py:111-150): the two pretrainer classes of the reference package, served by the MI355X-native
implementations (pretrain_mlm.py:74-88 / pretrain_mim.py:77-99 are the same classes with a mask argument).  The
Hugging Face baseline wrappers of that file (vl_BERT_Wrapper, ViltWrapper, ...) are not part of the MEANT hot path."""
from meant_amd.modules import meant_language_pretrainer, meant_vision_pretrainer  # noqa: F401
