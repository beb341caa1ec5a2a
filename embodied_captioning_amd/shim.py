"""Make the reference's import paths resolve to this package, so ``scripts/run_exp.py`` / ``run_pseudolabeler.py`` and
the agents import the HIP-backed captioner unchanged:

    import embodied_captioning_amd.shim as shim; shim.install()
    from experimenting_env.utils.predictor_utils import Captioner          # -> this package

Only the captioner modules are aliased; an existing ``experimenting_env`` package (the real
reference checkout) keeps everything else and just gets these submodules overridden."""
import importlib
import sys
import types

_ALIASES = {
    "experimenting_env.captioner.captioning_predictor": "embodied_captioning_amd.captioner.captioning_predictor",
    "experimenting_env.captioner.utils.utils": "embodied_captioning_amd.captioner.utils.utils",
    "experimenting_env.captioner.utils.utils_captioner": "embodied_captioning_amd.captioner.utils.utils_captioner",
    "experimenting_env.captioner.models.blip.blip": "embodied_captioning_amd.captioner.models.blip.blip",
    "experimenting_env.captioner.models.blip2.blip2": "embodied_captioning_amd.captioner.models.blip2.blip2",
    "experimenting_env.captioner.models.coca.coca": "embodied_captioning_amd.captioner.models.coca.coca",
}


def _ensure_pkg(name: str):
    if name in sys.modules:
        return sys.modules[name]
    mod = types.ModuleType(name)
    mod.__path__ = []
    sys.modules[name] = mod
    if "." in name:
        parent, _, child = name.rpartition(".")
        setattr(_ensure_pkg(parent), child, mod)
    return mod


def install(override_predictor_utils: bool = True) -> None:
    aliases = dict(_ALIASES)
    if override_predictor_utils and "experimenting_env.utils.predictor_utils" not in sys.modules:
        aliases["experimenting_env.utils.predictor_utils"] = "embodied_captioning_amd.utils.predictor_utils"
    for alias, target in aliases.items():
        mod = importlib.import_module(target)
        parent, _, child = alias.rpartition(".")
        setattr(_ensure_pkg(parent), child, mod)
        sys.modules[alias] = mod
