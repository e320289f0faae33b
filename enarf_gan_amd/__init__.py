"""Import alias for the `enarf-gan_amd/` package directory.

The package directory carries the project's name (`enarf-gan_amd`, with a hyphen), which Python's
import statement cannot spell. This stub makes `import enarf_gan_amd` resolve submodules from that
directory and runs its `__init__.py` in this module's namespace. It holds no code of its own.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "enarf-gan_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f, _real, _os
