"""Scratch shim used ONLY by tests/golden/make_golden.py when importing the reference
(`ujson` is not installed in the build image).  Re-exports the stdlib json API."""
import json as _json
from json import load, loads  # noqa: F401


def _default(o):
    for attr in ("toDict", "todict"):
        if hasattr(o, attr):
            return getattr(o, attr)()
    return str(o)


def dumps(obj, *a, **kw):
    kw.setdefault("default", _default)
    return _json.dumps(obj, *a, **kw)


def dump(obj, fp, *a, **kw):
    kw.setdefault("default", _default)
    return _json.dump(obj, fp, *a, **kw)
