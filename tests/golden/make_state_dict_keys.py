"""DEV-ONLY generator of tests/golden/state_dict_keys.json (runs only where /root/reference exists).

Builds the REFERENCE's full ``ConsistentOnlineDynamicDepth`` from the reference's own model config
(configs/models/codd.py, loaded with runpy) through the stub registry of tools/ref_import.py and dumps the name and
shape of every state-dict entry: the checkpoint boundary a published ``.pth`` (reference inference.py:123) has to
cross.  mmseg's ``HRNet`` is absent from /root/reference (un-vendored dependency), so a parameter-free placeholder is
registered for it: the keys below ``motion.raft3d.cnet.0.`` are NOT in the "reference" section; they are listed
separately, generated from codd_amd's own HRNet (mmseg's naming rule as implemented in codd_amd/hrnet.py) and marked
unpinned.  Only names and shapes are stored -- no reference source, no weights.

    python tests/golden/make_state_dict_keys.py
"""
import json
import os
import runpy
import sys

import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "state_dict_keys.json")


def own_keys():
    """codd_amd's estimator built from its own config mirror (before the reference is imported)."""
    import codd_amd  # noqa: F401
    from codd_amd import configs
    from codd_amd.registry import build_estimator
    est = build_estimator(configs.codd(iters=16))
    return {k: list(v.shape) for k, v in est.state_dict().items()}


def reference_keys():
    import ref_import
    MODELS = ref_import.install_stubs()

    class HRNet(nn.Module):  # placeholder for mmseg.models.backbones.HRNet (absent): contributes no keys
        def __init__(self, **kw):
            super().__init__()

    MODELS.register_module(name="HRNet", module=HRNet)
    ref_import.import_reference()
    from model.builder import build_estimator
    cfg = runpy.run_path(os.path.join(ref_import.REF_ROOT, "configs", "models", "codd.py"))["model"]
    est = build_estimator(cfg)
    return {k: list(v.shape) for k, v in est.state_dict().items()}


def main():
    own = own_keys()
    ref = reference_keys()
    hr = "motion.raft3d.cnet.0."
    assert not any(k.startswith(hr) for k in ref)
    out = dict(
        source="reference configs/models/codd.py built through tools/ref_import.py (names + shapes only)",
        reference=ref,
        hrnet_unpinned={k: v for k, v in own.items() if k.startswith(hr)},
        hrnet_note="mmseg HRNet is not under /root/reference: names follow mmseg's module tree as implemented in "
                   "codd_amd/hrnet.py (conv1/bn1/conv2/bn2/layer1/transition{1,2,3}/stage{2,3,4}.<m>.branches.<b>.<i>."
                   "{conv,bn}{1,2}, fuse_layers.<i>.<j>...), never checked against an mmseg checkpoint",
        # entries a published checkpoint carries beyond the inference graph (dropped by apis.load_checkpoint)
        training_only_examples=["stereo.loss.convx.weight", "stereo.loss.convy.weight"],
    )
    only_ref = sorted(set(ref) - set(own))
    only_own = sorted(k for k in set(own) - set(ref) if not k.startswith(hr))
    print(f"reference {len(ref)} keys, codd_amd {len(own)} keys ({len(out['hrnet_unpinned'])} HRNet); "
          f"only in reference: {only_ref[:5]}; only in codd_amd (non-HRNet): {only_own[:5]}")
    json.dump(out, open(OUT, "w"), indent=0, sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
