from .base import Sequential  # noqa: F401
from .easydgl import EasyDGL  # noqa: F401
from .ctsma import CTSMA  # noqa: F401
from .tgat import TGAT  # noqa: F401
from .tisasrec import TiSASRec  # noqa: F401
