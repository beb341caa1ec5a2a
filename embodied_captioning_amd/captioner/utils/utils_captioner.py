"""`select_captioner(cfg)` - reference ``experimenting_env/captioner/utils/utils_captioner.py:4-11`` extended with
the 'blip' architecture BASELINE.json names."""


def select_captioner(cfg):
    arch_name = cfg.arch_name
    assert arch_name.lower() in ["coca", "blip2", "blip"], \
        "Currently, only 'coca', 'blip2' and 'blip' architectures are supported."
    if arch_name.lower() == "blip":
        from ..models.blip.blip import BLIP
        return BLIP(cfg)
    if arch_name.lower() == "coca":
        from ..models.coca.coca import CoCa
        return CoCa(cfg)
    from ..models.blip2.blip2 import BLIP2
    return BLIP2(cfg)
