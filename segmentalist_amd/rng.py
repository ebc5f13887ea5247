"""
RNG indirection.  The reference consumes the process-global Python
`random` stream; Python 3's `random.shuffle` uses a different algorithm than Python 2.7's,
so seeded runs recorded under Python 2 (e.g. the example notebook) need the old one.
"""
import random

_kind = "py3"


def set_shuffle(kind):
    """kind: "py3" (stdlib random.shuffle) or "py2" (the Python-2.7 algorithm)."""
    global _kind
    assert kind in ("py2", "py3")
    _kind = kind


def shuffle(x):
    if _kind == "py3":
        random.shuffle(x)
    else:
        for i in reversed(range(1, len(x))):
            j = int(random.random() * (i + 1))
            x[i], x[j] = x[j], x[i]
