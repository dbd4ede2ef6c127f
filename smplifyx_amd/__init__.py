"""Importable alias for the product package.

The product package lives in the directory ``smplify-x-partial_amd/`` (the name the
build contract fixes).  A hyphenated directory cannot be imported with a plain
``import`` statement, so this stub re-points its own ``__path__`` at that directory
and executes the real ``__init__``: ``import smplifyx_amd.fitting`` resolves to
``smplify-x-partial_amd/fitting.py``.
"""
import os as _os

_REAL = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
    "smplify-x-partial_amd")
__path__ = [_REAL]
_init = _os.path.join(_REAL, "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
del _f, _init
