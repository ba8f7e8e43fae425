"""fastenhancer_amd — MI355X-native (gfx950) forward path for the FastEnhancer
streaming speech-enhancement models, call-compatible with the inference surface
of aask1357/fastenhancer (see DESIGN.md / INTEGRATION.md).

The compute path is libfastenhancer_hip.so (hand-written HIP, C ABI in
include/fastenhancer_hip.h).  There is no CPU fallback: importing the package
works anywhere, but every compute entry point raises if the library or a GPU is
missing."""
from .config import FEConfig  # noqa: F401
from .engine import Engine  # noqa: F401

__version__ = "0.1.0"
