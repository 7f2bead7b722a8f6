"""MI355X-native fake-quantization kernels behind LLM-QAT's operator API (see api.py)."""
from .api import *  # noqa: F401,F403
from .api import __all__, __version__  # noqa: F401
