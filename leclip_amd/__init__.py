"""Importable alias for the hyphen-named package directory.

The product package lives in
``language-enhanced-clip-for-multi-label-image-recognition_amd/`` (the name the
project contract fixes).  A hyphen cannot appear in a Python ``import``
statement, so this stub re-points its ``__path__`` at that directory: every
``leclip_amd.<sub>`` import resolves to a file there and nothing is duplicated.
"""
import os as _os

_REAL = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
    "language-enhanced-clip-for-multi-label-image-recognition_amd",
)
__path__ = [_REAL]
REPO_ROOT = _os.path.dirname(_REAL)
PACKAGE_DIR = _REAL

with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
