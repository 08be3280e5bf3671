"""TEST-HARNESS FIXTURE — the `parameterized` package is not installed in this image (no network).

The reference's einsum_test.py (cuTENSOR/python/cutensor/torch/einsum_test.py:22-23,35-36,153-154) uses exactly
two names from it: `param(name, **kwargs)` and the `@parameterized.expand([...])` decorator on unittest methods.
This module provides those two with the package's behaviour for that use: one generated `test_..._<i>_<name>`
method per param on the enclosing class, the undecorated method removed from collection.
"""
import inspect
import re


class param:
    def __init__(self, *args, **kwargs):
        self.args = args
        self.kwargs = kwargs


class parameterized:
    @staticmethod
    def expand(params):
        params = [p if isinstance(p, param) else param(*p) for p in params]

        def decorator(fn):
            frame_locals = inspect.currentframe().f_back.f_locals   # the class body being executed
            for i, p in enumerate(params):
                suffix = re.sub(r"\W+", "_", str(p.args[0])) if p.args else ""
                name = "%s_%d_%s" % (fn.__name__, i, suffix)

                def make(p=p):
                    def test(self):
                        return fn(self, *p.args, **p.kwargs)
                    return test
                t = make()
                t.__name__ = name
                t.__doc__ = fn.__doc__
                frame_locals[name] = t
            return None   # like the real package: the template itself is not collected
        return decorator
