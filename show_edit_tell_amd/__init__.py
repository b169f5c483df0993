"""Importable alias for the `show-edit-tell_amd/` package directory.

The package directory carries the reference repo's name (with hyphens, which Python cannot
import); this shim makes `import show_edit_tell_amd` resolve to it.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "show-edit-tell_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
