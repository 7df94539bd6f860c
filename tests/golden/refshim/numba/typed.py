class List(list):
    @classmethod
    def empty_list(cls, *a, **kw):
        return cls()


class Dict(dict):
    @classmethod
    def empty(cls, *a, **kw):
        return cls()
