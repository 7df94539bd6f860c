def prod(iterable):
    """Integer product without overflow (cf. reference common/math.py)."""
    r = 1
    for x in iterable:
        r *= int(x)
    return r
