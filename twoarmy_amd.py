"""Import alias: `import twoarmy_amd` == the package directory
`goal-conditioned-reinforcement-learning-with-environmental-and-policy-priors_amd/` (whose literal
name is not a Python identifier).  A meta-path finder maps every `twoarmy_amd.x.y` to the already
imported real module object, so there is exactly one module object per file under either name.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys

REAL = "goal-conditioned-reinforcement-learning-with-environmental-and-policy-priors_amd"
ALIAS = __name__


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, module):
        self._module = module

    def create_module(self, spec):
        return self._module

    def exec_module(self, module):
        pass

    # runpy (`python -m twoarmy_amd.soa.train_ppo`) asks the loader for the code object
    def get_code(self, fullname):
        real = REAL + fullname[len(ALIAS):]
        return importlib.util.find_spec(real).loader.get_code(real)

    def is_package(self, fullname):
        return hasattr(self._module, "__path__")


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(ALIAS + "."):
            return None
        real = importlib.import_module(REAL + fullname[len(ALIAS):])
        return importlib.machinery.ModuleSpec(fullname, _AliasLoader(real), origin=getattr(real, "__file__", None),
                                              is_package=hasattr(real, "__path__"))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

_pkg = importlib.import_module(REAL)
_pkg.REAL_NAME = REAL
sys.modules[ALIAS] = _pkg
