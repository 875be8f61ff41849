from .batch import Batch  # noqa: F401
