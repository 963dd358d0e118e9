registry = {}


def register(id, entry_point=None, kwargs=None, **other):
    registry[id] = (entry_point, dict(kwargs or {}))


def make(id, **kwargs):
    entry_point, base = registry[id]
    return entry_point(**{**base, **kwargs})
