"""Wall-clock helpers with the reference's semantics (`src/icepy4d/utils/timer.py:8-60`): `timeit` prints the
duration of a call, `AverageTimer` keeps exponentially smoothed named laps ("matching",
"geometric_verification", "preselection" are the lap names `match()` uses, `matchers.py:208, 224, 560`)."""
import logging
import time
from collections import OrderedDict
from functools import wraps


def timeit(func):
    @wraps(func)
    def wrapper(*args, **kwargs):
        t0 = time.perf_counter()
        out = func(*args, **kwargs)
        print(f"Function {func.__name__} took {time.perf_counter() - t0:.4f} seconds")
        return out

    return wrapper


class AverageTimer:
    def __init__(self, smoothing: float = 0.3, logger=None):
        self.smoothing = smoothing
        self.times = OrderedDict()
        self.will_print = OrderedDict()
        self.logger = logger
        self.reset()

    def reset(self):
        self.start = self.last_time = time.time()
        for name in self.will_print:
            self.will_print[name] = False

    def update(self, name: str = "default"):
        now = time.time()
        dt = now - self.last_time
        if name in self.times:
            dt = self.smoothing * dt + (1 - self.smoothing) * self.times[name]
        self.times[name] = dt
        self.will_print[name] = True
        self.last_time = now

    def print(self, text: str = "Timer"):
        msg = f"[Timer] | [{text}] " + "".join(
            f"{k}={v:.3f}, " for k, v in self.times.items() if self.will_print[k])
        (self.logger.info if self.logger is not None else logging.info)(msg)
        self.reset()
