"""maxstyle_amd - MI355X-native implementation of MaxStyle's inner adversarial style-optimisation path.

Importing this package loads libmaxstyle_hip.so and raises ImportError if it has not been built."""
from ._lib import LIB_PATH, MaxStyleHipError  # noqa: F401  (fails loudly when the extension is missing)
from .maxstyle import MaxStyle  # noqa: F401
from .mixstyle import MixStyle  # noqa: F401
from .networks import MyEncoder, MyDecoder, Dual_Branch_Encoder, res_convdown, res_up_family, _disable_tracking_bn_stats  # noqa: F401
from .solver import AdvancedTripletReconSegmentationModel, basic_loss_fn, cross_entropy_2D  # noqa: F401

__all__ = ["MaxStyle", "MixStyle", "MyEncoder", "MyDecoder", "Dual_Branch_Encoder", "AdvancedTripletReconSegmentationModel",
           "basic_loss_fn", "cross_entropy_2D", "MaxStyleHipError", "LIB_PATH"]
