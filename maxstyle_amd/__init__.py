"""maxstyle_amd - MI355X-native implementation of MaxStyle's inner adversarial style-optimisation path.

Importing this package loads libmaxstyle_hip.so and raises ImportError if it has not been built."""
from ._lib import LIB_PATH, MaxStyleHipError  # noqa: F401  (fails loudly when the extension is missing)
from .maxstyle import MaxStyle  # noqa: F401

__all__ = ["MaxStyle", "MaxStyleHipError", "LIB_PATH"]
