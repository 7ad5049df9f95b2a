"""DEV-ONLY: import the upstream reference (/root/reference) in the build container.

The reference depends on mmcv / mmseg / lietorch / pytorch3d, none of which is installed.
This module injects minimal registry stubs into ``sys.modules`` so that the pure-torch
parts of the reference (HITNetMF, Fusion, RAFT3D's encoder / update block / all-pairs
correlation / convex upsampling / projective ops, utils.warp) import and run on CPU.

It is used ONLY by ``tests/golden/make_golden.py`` to generate the committed golden
vectors and to pin ``oracle/``.  Nothing in the product, the tests or the bench imports
it at run time (the reference does not exist on the GPU box).
"""
import sys
import types

import torch.nn as nn

REF_ROOT = "/root/reference"


class _Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def build(self, cfg, default_args=None):
        cfg = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                cfg.setdefault(k, v)
        typ = cfg.pop("type")
        return self.module_dict[typ](**cfg)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    if "mmseg" in sys.modules and getattr(sys.modules["mmseg"], "_codd_stub", False):
        return sys.modules["mmseg.models.builder"].MODELS
    MODELS = _Registry("models")

    class _NullLoss(nn.Module):
        def __init__(self, **kw):
            super().__init__()

    def build_loss(cfg):
        return _NullLoss()

    builder = _mod("mmseg.models.builder", MODELS=MODELS, BACKBONES=MODELS, LOSSES=MODELS,
                   build_backbone=MODELS.build, build_loss=build_loss)
    models = _mod("mmseg.models", builder=builder, LOSSES=MODELS)
    mmseg = _mod("mmseg", models=models, _codd_stub=True)

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None, **kw):
            super().__init__()

    def auto_fp16(*a, **k):
        def deco(f):
            return f
        return deco

    HOOKS = _Registry("hooks")

    class LrUpdaterHook:
        def __init__(self, **kw):
            pass

    _mod("mmcv.runner", BaseModule=BaseModule, auto_fp16=auto_fp16, HOOKS=HOOKS,
         LrUpdaterHook=LrUpdaterHook)
    _mod("mmcv.utils", mkdir_or_exist=lambda *a, **k: None)

    class _BatchNorm(nn.Module):
        pass

    _mod("mmcv.utils.parrots_wrapper", _BatchNorm=_BatchNorm)
    noop = lambda *a, **k: None
    _mod("mmcv.cnn", constant_init=noop, kaiming_init=noop, normal_init=noop,
         trunc_normal_init=noop)
    _mod("mmcv", is_list_of=lambda seq, t: all(isinstance(s, t) for s in seq),
         runner=sys.modules["mmcv.runner"], utils=sys.modules["mmcv.utils"],
         cnn=sys.modules["mmcv.cnn"])

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    class PointsRenderer(nn.Module):
        def __init__(self, rasterizer=None, compositor=None):
            super().__init__()

    _mod("pytorch3d")
    _mod("pytorch3d.renderer", PerspectiveCameras=_Dummy, PointsRasterizationSettings=_Dummy,
         PointsRenderer=PointsRenderer, PointsRasterizer=_Dummy, AlphaCompositor=_Dummy)
    _mod("pytorch3d.structures", Pointclouds=_Dummy)
    _mod("lietorch", SE3=_Dummy)
    _mod("lietorch_extras")
    return MODELS


def import_reference():
    """Returns the reference's ``model`` package (with stubs installed)."""
    install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # the HF `datasets` pip package would shadow the reference's; we never import it here
    import model  # noqa: E402  (reference package)
    return model
