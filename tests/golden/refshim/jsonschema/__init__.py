def validate(*a, **kw):
    return None
