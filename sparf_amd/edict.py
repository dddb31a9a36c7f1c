"""EasyDict: dict with attribute access (stand-in for the `easydict` package, which is not installed in this image).

The reference does `from easydict import EasyDict as edict` everywhere
(e.g. /root/reference/source/models/renderer.py:20).  Only the behaviour the
renderer path relies on is provided: attribute access == item access,
recursive wrapping of nested dicts, `update`, `pop`.  Put `compat/` on
PYTHONPATH to satisfy that import when the real package is absent.
"""


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        if d is None:
            d = {}
        if kwargs:
            d = dict(d, **kwargs)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, e=None, **f):
        d = dict(e or {})
        d.update(f)
        for k, v in d.items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]
