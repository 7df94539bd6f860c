"""`import libertem...` served by this package.

    import libertem_amd.compat
    libertem_amd.compat.install()          # before the first `import libertem`
    from libertem.api import Context       # -> libertem_amd.api.Context
    from libertem.udf.masks import ApplyMasksUDF

Scripts and UDFs written against the reference's module names (`libertem.api`, `libertem.udf`, `libertem.udf.masks`,
`libertem.masks`, `libertem.common.buffers`, `libertem.io.dataset.memory`, ...) then run on this package without edits,
as far as they stay on the path it implements; a module this package does not have is an ImportError as usual.  The
alias is per process and explicit: nothing is installed on import of `libertem_amd` itself, and a real `libertem`
that is already imported is left alone (`install(force=True)` replaces it for modules imported from then on).
"""
import importlib
import importlib.abc
import importlib.util
import sys

_PREFIX = 'libertem'
_TARGET = 'libertem_amd'


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name != _PREFIX and not name.startswith(_PREFIX + '.'):
            return None
        real = _TARGET + name[len(_PREFIX):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except (ImportError, ValueError):
            return None
        return importlib.util.spec_from_loader(name, self, is_package=True)

    def create_module(self, spec):
        # the SAME module object under both names: classes, caches and isinstance checks are shared
        return importlib.import_module(_TARGET + spec.name[len(_PREFIX):])

    def exec_module(self, module):
        pass


_finder = None


def install(force=False):
    """Serve `libertem` and its submodules from `libertem_amd`.  Returns True when the alias is (now) active."""
    global _finder
    if _finder is not None:
        return True
    if _PREFIX in sys.modules and not force:
        mod = sys.modules[_PREFIX]
        if getattr(mod, '__name__', None) != _TARGET:
            return False                       # the reference itself is imported: left alone
    if force:
        for k in [k for k in sys.modules if k == _PREFIX or k.startswith(_PREFIX + '.')]:
            del sys.modules[k]
    _finder = _AliasFinder()
    sys.meta_path.insert(0, _finder)
    return True


def uninstall():
    """Remove the alias (modules already imported under the `libertem` name stay in sys.modules)."""
    global _finder
    if _finder is not None:
        try:
            sys.meta_path.remove(_finder)
        except ValueError:
            pass
        _finder = None
