"""easydgl_amd — MI355X-native (gfx950) hot path of EasyDGL's self-modulating attention.

Host side mirrors the reference's operator/model interface (src/module/{coding,temporal}.py,
src/model/{Base,EasyDGL}.py, src/util.py:ranking); the compute is libeasydgl_hip.so (include/easydgl_hip.h).
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)
from .model import EasyDGL, Sequential  # noqa: F401
from .util import ranking  # noqa: F401
