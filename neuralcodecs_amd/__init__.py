"""neuralcodecs_amd -- MI355X-native Encode/RVQ/Decode engine for DAC / SNAC / Encodec.

The compute lives in ``libnc_mi355x.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/nc_mi355x.h``); this package is the thin host-side mirror of the reference's model
classes (NeuralCodecs.Torch/Models/{DAC,SNAC,Encodec}.cs).
"""
from .config import DACConfig, EncodecConfig, SNACConfig  # noqa: F401
from .dac import DAC  # noqa: F401
from .snac import SNAC  # noqa: F401
from .encodec import EncodedFrame, Encodec  # noqa: F401

__all__ = ["DAC", "SNAC", "Encodec", "EncodedFrame", "DACConfig", "SNACConfig", "EncodecConfig"]
from . import audio  # noqa: F401,E402
