from .timer import AverageTimer, timeit  # noqa: F401
