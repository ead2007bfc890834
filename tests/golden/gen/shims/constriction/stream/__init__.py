from . import model, queue  # noqa: F401
