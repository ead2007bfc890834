from . import stream  # noqa: F401
