"""Host-side helpers (``dsp``) around the velvet-noise hot path."""
import functools
import statistics
import time


class timed:
    """``@timed(repititions=5)``: call the wrapped function that many times, report the mean
    wall time, return the last result.  Same decorator name and (misspelt) knob as the
    reference's timing helper (``utils/__init__.py:5-26``) so that scripts using it keep working."""

    def __init__(self, repititions: int = 5):
        self.repititions = int(repititions)

    def _one_run(self, func, args, kwargs):
        began = time.perf_counter()
        value = func(*args, **kwargs)
        return time.perf_counter() - began, value

    def __call__(self, func):
        @functools.wraps(func)
        def measured(*args, **kwargs):
            runs = [self._one_run(func, args, kwargs) for _ in range(self.repititions)]
            mean_ms = 1e3 * statistics.fmean(seconds for seconds, _ in runs) if runs else 0.0
            print(f"Function '{func.__name__}' executed {self.repititions} times "
                  f'averaging {mean_ms:.4f} milliseconds.')
            return runs[-1][1] if runs else None
        return measured
