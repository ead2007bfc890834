from . import transforms, models  # noqa: F401
