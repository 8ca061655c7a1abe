"""Importable alias of the `noble-curves_amd/` package directory (a hyphen is not a legal
Python identifier).  `import noble_curves_amd` executes noble-curves_amd/__init__.py with
this module as the package, so submodules resolve inside noble-curves_amd/."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "noble-curves_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
