"""Module-level drop-in: make ``import planners...`` resolve to this package's mirror of the reference's ``planners``
package, so that the reference's ``agent.py`` / ``loader.py`` / ``simulator.py`` run unchanged on top of the MI355X path.

    import mind_amd.dropin; mind_amd.dropin.install()      # before the reference's modules are imported
    from run_sim import main; main()

``install()`` imports every module of ``mind_amd.planners`` and registers it in ``sys.modules`` under the reference's
name as well (``planners.mind.planner`` -> ``mind_amd.planners.mind.planner`` ...): both names are the SAME module
objects, relative imports inside the package keep working, and the dotted config names of the reference's planner JSON
(``planners.mind.configs.planning.demo_1``) resolve here.  (Putting ``mind_amd/`` itself on ``sys.path`` does not work:
the package's modules import their siblings relative to ``mind_amd``.)
"""
import importlib
import pkgutil
import sys


def install(prefix="planners"):
    import mind_amd.planners as root
    names = ["mind_amd.planners"]
    for m in pkgutil.walk_packages(root.__path__, "mind_amd.planners."):
        names.append(m.name)
    for k in [k for k in sys.modules if k == prefix or k.startswith(prefix + ".")]:
        del sys.modules[k]                                   # a previously imported reference package of that name
    for name in names:
        mod = importlib.import_module(name)
        sys.modules[prefix + name[len("mind_amd.planners"):]] = mod
    return sys.modules[prefix]
