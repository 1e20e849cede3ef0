from .base import Sequential  # noqa: F401
from .easydgl import EasyDGL  # noqa: F401
