"""BACKBONES registry: mmdet's own when mmdet is importable (so ``type='SimplePanoSwinTransformer'`` in a
config resolves to this implementation), otherwise a minimal stand-in with the same ``register_module`` /
``build`` surface -- the same fallback the reference uses (simple_panoswin_transformer.py:36-40)."""


class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self.module_dict[key] = cls
            return cls

        return _register(module) if module is not None else _register

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg):
        cfg = dict(cfg)
        cls = self.get(cfg.pop("type"))
        if cls is None:
            raise KeyError(f"unknown {self.name} type")
        return cls(**cfg)


def _find_backbones():
    try:
        from mmdet.models.builder import BACKBONES as reg          # mmdet 2.x (the reference's fork)
        return reg, True
    except Exception:
        return Registry("backbone"), False


BACKBONES, USING_MMDET_REGISTRY = _find_backbones()


def build_backbone(cfg):
    """mmdet.models.builder.build_backbone equivalent (mmdet/models/builder.py:38-40)."""
    if USING_MMDET_REGISTRY:
        from mmdet.models.builder import build_backbone as _bb
        return _bb(cfg)
    return BACKBONES.build(cfg)
