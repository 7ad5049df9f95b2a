"""codd_amd: MI355X-native implementation of CODD's per-frame stereo -> motion -> fusion path.

Importing the package registers the plug-in classes (reference registry surface:
ConsistentOnlineDynamicDepth, HITNetMF, HITUNet, TileInitialization, TilePropagation, Motion,
RAFT3D, HRNet, Fusion) in ``codd_amd.registry.MODELS``.
"""
from . import registry  # noqa: F401
from . import stereo  # noqa: F401
from . import hrnet  # noqa: F401
from . import motion  # noqa: F401
from . import fusion  # noqa: F401
from . import estimator  # noqa: F401
from . import ablation  # noqa: F401
from .registry import MODELS, build_estimator  # noqa: F401

__all__ = ["MODELS", "build_estimator"]
