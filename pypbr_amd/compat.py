"""Run reference scripts unchanged: `import pypbr_amd.compat; pypbr_amd.compat.install()` registers
`pypbr`, `pypbr.models`, `pypbr.materials`, `pypbr.utils`, `pypbr.io` and `pypbr.blending` as aliases of the
pypbr_amd modules, so that e.g. examples/example_brdf.py's

    from pypbr.models import CookTorranceBRDF
    from pypbr.io import load_material_from_folder

resolve to the MI355X implementation.  Only the Cook-Torrance path and the calls either side of it
(load, blend, resize, tile) exist here; everything else of PyPBR (transforms, authoring utilities,
...) is out of scope and raises ImportError/AttributeError as an absent module would."""
import sys
import types


def install(force: bool = False) -> types.ModuleType:
    """Registers the aliases.  Refuses to shadow an already-imported real `pypbr` unless `force`."""
    import pypbr_amd
    from pypbr_amd import blending, io, materials, models, utils

    existing = sys.modules.get("pypbr")
    if existing is not None and not getattr(existing, "__pypbr_amd_alias__", False) and not force:
        raise RuntimeError("a different `pypbr` package is already imported from %s"
                           % getattr(existing, "__file__", "?"))
    pkg = types.ModuleType("pypbr")
    pkg.__doc__ = "alias of pypbr_amd (MI355X Cook-Torrance path)"
    pkg.__path__ = []                      # a package, with no files of its own
    pkg.__pypbr_amd_alias__ = True
    pkg.__version__ = pypbr_amd.__version__
    sys.modules["pypbr.blending.functional"] = blending
    for name, mod in (("models", models), ("materials", materials), ("utils", utils), ("io", io), ("blending", blending)):
        setattr(pkg, name, mod)
        sys.modules["pypbr." + name] = mod
    sys.modules["pypbr"] = pkg
    return pkg


def uninstall() -> None:
    for name in [k for k, v in sys.modules.items() if k == "pypbr" or k.startswith("pypbr.")]:
        mod = sys.modules[name]
        if name == "pypbr" and not getattr(mod, "__pypbr_amd_alias__", False):
            return
        del sys.modules[name]
