"""Import shim: the package sources live in ./dvd-gan_amd/ (a directory name Python cannot
import directly).  `import dvd_gan_amd` resolves every submodule from there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dvd-gan_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
