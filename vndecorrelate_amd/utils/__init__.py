"""Host-side helpers (``dsp``) around the velvet-noise hot path."""
import time
from functools import wraps


def timed(repititions: int = 5):
    """Decorator: run the function ``repititions`` times and print the mean wall
    time in milliseconds (same knob name, typo included, as the reference's
    ``utils/__init__.py:5-26``)."""

    def decorator(func):
        @wraps(func)
        def wrapper(*args, **kwargs):
            total = 0.0
            result = None
            for _ in range(repititions):
                start = time.perf_counter()
                result = func(*args, **kwargs)
                total += time.perf_counter() - start
            print(f"Function '{func.__name__}' executed {repititions} times averaging "
                  f'{(total / repititions) * 1000:.4f} milliseconds.')
            return result

        return wrapper

    return decorator
