"""Import shim: the product package lives in the directory `mr-gan_amd/` (a name Python cannot import
directly), so `import mr_gan_amd` resolves its submodules there."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mr-gan_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
