from . import coding, temporal  # noqa: F401
