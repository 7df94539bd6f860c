def fix_code(code, *a, **kw):
    return code
