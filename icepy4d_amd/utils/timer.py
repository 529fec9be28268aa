"""Wall-clock helpers used by `match()`.

Behavioural contract taken from the reference (`src/icepy4d/utils/timer.py:8-60`): `timeit` prints how long a call took;
`AverageTimer.update(name)` records the time since the previous update under `name` (exponentially smoothed with the
previous value of that lap, factor `smoothing`), `print(text)` logs the laps updated since the last print and restarts
the clock. `match()` uses the lap names "matching", "geometric_verification" and "preselection"
(`matchers.py:208, 224, 560`)."""
import functools
import logging
import time
from collections import OrderedDict


def timeit(func):
    """Decorator: print `Function <name> took <t> seconds` after every call."""
    @functools.wraps(func)
    def timed(*args, **kwargs):
        began = time.perf_counter()
        try:
            return func(*args, **kwargs)
        finally:
            print(f"Function {func.__name__} took {time.perf_counter() - began:.4f} seconds")

    return timed


class AverageTimer:
    def __init__(self, smoothing: float = 0.3, logger=None):
        self.smoothing = float(smoothing)
        self.logger = logger
        self.times = OrderedDict()        # lap name -> smoothed seconds
        self.will_print = OrderedDict()   # lap name -> updated since the last print?
        self.reset()

    def reset(self) -> None:
        self.start = self.last_time = time.time()
        for lap in list(self.will_print):
            self.will_print[lap] = False

    def update(self, name: str = "default") -> None:
        now = time.time()
        lap = now - self.last_time
        previous = self.times.get(name)
        self.times[name] = lap if previous is None else self.smoothing * lap + (1.0 - self.smoothing) * previous
        self.will_print[name] = True
        self.last_time = now

    def print(self, text: str = "Timer") -> None:
        laps = [f"{lap}={sec:.3f}" for lap, sec in self.times.items() if self.will_print.get(lap)]
        line = f"[Timer] | [{text}] " + "".join(item + ", " for item in laps)
        (self.logger.info if self.logger is not None else logging.info)(line)
        self.reset()
