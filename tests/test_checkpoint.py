"""Checkpoint boundary: state-dict names / shapes against the fixture dumped from the reference's own
model (tests/golden/state_dict_keys.json, generator tests/golden/make_state_dict_keys.py) and a strict load of a
checkpoint file in the shape mmcv writes (reference inference.py:123 -> mmcv.runner.load_checkpoint)."""
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))


def _build():
    import codd_amd  # noqa: F401
    from codd_amd import configs
    from codd_amd.registry import build_estimator
    return build_estimator(configs.codd(iters=16)).eval()


def test_state_dict_names_and_shapes_equal_the_reference_model():
    own = {k: list(v.shape) for k, v in _build().state_dict().items()}
    ref, hr = FIX["reference"], FIX["hrnet_unpinned"]
    assert len(ref) == 313 and len(hr) == 864  # 1 177 entries in total
    # every entry of the reference's ConsistentOnlineDynamicDepth (stereo, fusion, RAFT3D fnet / update block /
    # ResizeConcatConv): same name, same shape -- PINNED to the reference
    assert {k: own.get(k) for k in ref} == ref
    # HRNet below motion.raft3d.cnet.0.: mmseg naming rule (UNPINNED: mmseg is not under /root/reference)
    assert {k: own.get(k) for k in hr} == hr
    assert set(own) == set(ref) | set(hr)


@pytest.mark.parametrize("prefix", ["", "module."])
def test_strict_load_of_an_mmcv_style_checkpoint(tmp_path, prefix):
    """{"meta": ..., "state_dict": {[module.]name: tensor}} with the training-only entries a published file carries
    (stereo.loss.conv{x,y}.weight) must load with strict=True, tensor for tensor."""
    from codd_amd import apis
    est = _build()
    g = torch.Generator().manual_seed(7)
    sd = {}
    for k, shape in {**FIX["reference"], **FIX["hrnet_unpinned"]}.items():
        if k.endswith("num_batches_tracked"):
            sd[prefix + k] = torch.tensor(1234)
        elif k.endswith("running_var"):
            sd[prefix + k] = torch.rand(shape, generator=g) + 0.5
        else:
            sd[prefix + k] = torch.randn(shape, generator=g)
    for k in FIX["training_only_examples"]:
        sd[prefix + k] = torch.randn(1, 1, 3, 3, generator=g)
    path = str(tmp_path / "codd.pth")
    torch.save(dict(meta=dict(mmseg_version="0.30.0", CLASSES=None), state_dict=sd), path)
    res = apis.load_checkpoint(est, path, strict=True, log=lambda *_: None)
    assert res["missing"] == [] and res["unexpected"] == [] and res["meta"]["mmseg_version"] == "0.30.0"
    own = est.state_dict()
    for k in own:
        assert torch.equal(own[k], sd[prefix + k].to(own[k].dtype)), k


def test_strict_load_rejects_a_renamed_key(tmp_path):
    from codd_amd import apis
    est = _build()
    sd = {k: v.clone() for k, v in est.state_dict().items()}
    sd["motion.raft3d.cnet.0.stage9.weight"] = sd.pop("motion.raft3d.cnet.0.conv1.weight")
    path = str(tmp_path / "bad.pth")
    torch.save(dict(state_dict=sd), path)
    with pytest.raises(RuntimeError):
        apis.load_checkpoint(est, path, strict=True, log=lambda *_: None)


def test_key_map_remaps_foreign_hrnet_names(tmp_path):
    """The HRNet names are unpinned (mmseg is not vendored): a checkpoint that spells them differently loads through
    ``key_map`` -- (regex, replacement) pairs or a callable --, a collision is an error, None drops a tensor."""
    from codd_amd import apis
    est = _build()
    g = torch.Generator().manual_seed(11)
    own = est.state_dict()
    want = {k: torch.randn(v.shape, generator=g).to(v.dtype) if v.is_floating_point() else v.clone() for k, v in own.items()}
    pre = "motion.raft3d.cnet.0."
    foreign = {(("backbone." + k[len(pre):]) if k.startswith(pre) else k): v for k, v in want.items()}
    foreign["aux_head.conv_seg.weight"] = torch.zeros(3)
    path = str(tmp_path / "foreign.pth")
    torch.save(dict(state_dict=foreign), path)
    with pytest.raises(RuntimeError):  # without the map: 864 missing + 865 unexpected
        apis.load_checkpoint(est, path, strict=True, log=lambda *_: None)
    res = apis.load_checkpoint(est, path, strict=True, log=lambda *_: None,
                               key_map=lambda k: None if k.startswith("aux_head.") else (pre + k[9:] if k.startswith("backbone.") else k))
    assert res["missing"] == [] and res["unexpected"] == []
    now = est.state_dict()
    assert all(torch.equal(now[k], want[k]) for k in now)
    # the (regex, replacement) form
    foreign.pop("aux_head.conv_seg.weight")
    torch.save(dict(state_dict=foreign), path)
    res = apis.load_checkpoint(_build(), path, strict=True, log=lambda *_: None, key_map=[(r"^backbone\.", pre)])
    assert res["missing"] == [] and res["unexpected"] == []
    with pytest.raises(RuntimeError, match="two checkpoint tensors"):
        apis.load_checkpoint(_build(), path, log=lambda *_: None, key_map=lambda k: "x")
