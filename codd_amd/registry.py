"""A minimal mmcv-compatible Registry (reference model/builder.py:7-21).

The reference registers its plug-ins in mmseg's ``MODELS`` registry and builds them from config
dicts (``MODELS.build(cfg)``; model/codd.py:44-54, hitnet.py:28-30, motion.py:69).  mmcv / mmseg
are not installed here, so the same surface is provided locally; when mmseg IS importable the
classes are ALSO registered into ``mmseg.models.builder.MODELS`` so that the reference's config
files build this implementation unchanged.
"""


class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self.module_dict[key] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError("cfg must be a dict containing the key 'type'")
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        typ = args.pop("type")
        cls = self.module_dict.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        return cls(**args)


MODELS = Registry("models")
BACKBONES = MODELS
ESTIMATORS = MODELS
LOSSES = Registry("losses")


def _mirror_into_mmseg(cls):  # pragma: no cover - mmseg absent in this image
    try:
        from mmseg.models.builder import MODELS as MM
        MM.register_module(name=cls.__name__, force=True, module=cls)
    except Exception:
        pass
    return cls


def register(cls):
    MODELS.register_module()(cls)
    return _mirror_into_mmseg(cls)


def build_backbone(cfg):
    return MODELS.build(cfg)


def build_loss(cfg):
    """Losses are training-only (out of scope); configs that carry a ``loss`` key still build."""
    return None


def build_estimator(cfg, train_cfg=None, test_cfg=None):
    """reference model/builder.py:10-21."""
    assert cfg.get("train_cfg") is None or train_cfg is None
    assert cfg.get("test_cfg") is None or test_cfg is None
    return ESTIMATORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))
