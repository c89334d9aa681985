"""TEST SEAM of the product's model runner (model/gast_net.py::_Runner.ops_factory): which runners get the numpy mirror of the op
set (tests/fake_backend.py) instead of gast_hip.binding.HipOps.  This module lives under tests/ only; the product looks it up by
import and finds nothing outside the test tree (bench.py --dry-run-cpu, a launcher dry run on CPU, puts tests/ on sys.path itself)."""
import weakref

_REG = weakref.WeakKeyDictionary()


def register(runner, factory):
    _REG[runner] = factory


def factory_for(runner):
    return _REG.get(runner)
