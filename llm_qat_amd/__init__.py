"""Importable name for the package that lives in `llm-qat_amd/`.

`llm-qat_amd` (the directory name the project layout prescribes) is not a valid Python
identifier, so this stub points its search path at that directory and re-exports its
public names:  `import llm_qat_amd`, `from llm_qat_amd.utils_quant import QuantizeLinear`.
"""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "llm-qat_amd"))

from .api import *  # noqa: E402,F401,F403
from .api import __all__, __version__  # noqa: E402,F401
