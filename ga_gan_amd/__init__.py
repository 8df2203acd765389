"""Import alias: the package directory is named `ga-gan_amd`, which Python cannot
spell in an import statement.  `import ga_gan_amd` resolves every submodule from
that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'ga-gan_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _os, _f
