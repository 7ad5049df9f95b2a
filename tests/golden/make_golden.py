"""DEV-ONLY generator of the golden vectors (runs only in the build container, where
/root/reference exists): imports the reference through tools/ref_import.py (mmcv / mmseg /
lietorch / pytorch3d stubbed), feeds it the deterministic synthetic weights and inputs of
codd_amd.synth, and stores the reference's OUTPUTS as small .npz fixtures.  Inputs and weights are
NOT stored: tests regenerate them from the same closed-form generators (tests/golden/cases.py).

    python tests/golden/make_golden.py

Covers every stage of the hot path that the reference can run without its absent CUDA
dependencies: whole HITNetMF (S0-S8), Fusion.memory_query (F0-F6), BasicEncoder (M1),
CorrBlock.corr + pyramid (M3), BasicUpdateBlock + ConvGRU (M6), cvx_upsample (M8),
inv_project / project / depth_sampler (M5), disp_warp.  The lietorch / pytorch3d / mmseg-HRNet
pieces cannot be run here: parity unpinned for those.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cases  # noqa: E402

# build the synthetic weights from the product's parameter spec BEFORE the reference (and its stub
# registry) is imported, so that the product classes are not mirrored into the stub registry.
_SD = cases.state_dict()

import ref_import  # noqa: E402

ref_import.import_reference()
from mmseg.models.builder import MODELS  # noqa: E402  (stub registry holding the reference classes)
from model.builder import build_estimator  # noqa: E402
from model.motion.raft3d import projective_ops as pops, se3_field  # noqa: E402
from model.motion.raft3d.blocks.corr import CorrBlock  # noqa: E402
from model.motion.raft3d.blocks.extractor import BasicEncoder  # noqa: E402
from model.motion.raft3d.raft3d import BasicUpdateBlock  # noqa: E402
from model.motion.raft3d.sampler_ops import depth_sampler  # noqa: E402
from utils import disp_warp  # noqa: E402
from utils.misc import collect_metric  # noqa: E402


def load(module, sd, prefix):
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    missing = module.load_state_dict(sub, strict=True)
    return module.eval()


def main():
    sd = cases.state_dict()
    out = {}
    with torch.no_grad():
        # ---- stereo: whole HITNetMF ------------------------------------------------------------
        est = build_estimator(dict(type="ConsistentOnlineDynamicDepth", stereo=dict(
            type="HITNetMF", backbone=dict(type="HITUNet"),
            initialization=dict(type="TileInitialization", max_disp=320),
            propagation=dict(type="TilePropagation"))))
        load(est.stereo, sd, "stereo.")
        for name, (H, W) in cases.STEREO_SIZES.items():
            l, r = cases.stereo_pair(H, W)
            o = est.stereo.stereo_matching(l, r)
            out[f"stereo_{name}_pred_disp"] = o["pred_disp"]
            out[f"stereo_{name}_left_feat"] = o["left_feat"]
            fea_l, fea_r = est.stereo.extract_feat(l), est.stereo.extract_feat(r)
            _, hyps = est.stereo.tile_init(fea_l, fea_r)
            for i, h in enumerate(hyps):
                out[f"stereo_{name}_init_d{i}"] = h[:, 0:1]
        # ---- BASELINE.json configs[0]: stereo-only, 2-frame 512x256 sequence through the reference's own
        #      ConsistentOnlineDynamicDepth.inference (every 2nd pixel is stored: 128 KiB) -------------
        img, r_img = cases.cfg1_sequence()
        meta = [dict(img_shape=(256, 512, 3), disp_range=(1, 210), intrinsics=[1050.0, 1050.0, 256.0, 128.0])]
        out["cfg1_pred_disp"] = est.inference(img, r_img, meta, evaluate=False)[:, :, ::2, ::2]
        # ---- fusion ----------------------------------------------------------------------------
        fus = load(MODELS.build(dict(type="Fusion", in_channels=24, fusion_channel=32,
                                     corr_cfg=dict(type="px2patch", patch_size=3))), sd, "fusion.")
        o, st = cases.fusion_case()
        fus.memory_query(o, st)
        for k in ("pred_disp", "fusion_weights", "reset_weights", "left_feat"):
            out[f"fusion_{k}"] = o[k]
        o1, _ = cases.fusion_case()
        fus.memory_query(o1, {})
        out["fusion_first_left_feat"] = o1["left_feat"]
        # the same with the non-default correlation patch (corr_cfg.patch_size = 5: 25-tap px2patch windows)
        fus5 = load(MODELS.build(dict(type="Fusion", in_channels=24, fusion_channel=32,
                                      corr_cfg=dict(type="px2patch", patch_size=5))), cases.fusion_p5_state_dict(), "fusion.")
        o5, st5 = cases.fusion_case()
        fus5.memory_query(o5, st5)
        for k in ("pred_disp", "fusion_weights", "reset_weights"):
            out[f"fusion_p5_{k}"] = o5[k]
        # ---- RAFT3D pure-torch blocks ------------------------------------------------------------
        fnet = load(BasicEncoder(output_dim=128, norm_fn="instance"), sd, "motion.raft3d.fnet.")
        out["fnet"] = fnet(cases.image(64, 128))
        ub = load(BasicUpdateBlock(), sd, "motion.raft3d.update_block.")
        net, inp, corr, flow, twist, dz = cases.update_inputs()
        res = ub(net, inp, corr, flow, dz, twist)  # NB reference call order (raft3d.py:238-240)
        for k, v in zip(("net", "mask", "ae", "delta", "weight"), res):
            out[f"update_{k}"] = v[:, ::7] if k == "mask" else v
        f1, f2 = cases.fmaps()
        cb = CorrBlock(f1, f2, radius=3)
        for i, c in enumerate(cb.corr_pyramid):
            out[f"corr_lvl{i}"] = c.reshape(c.shape[0], c.shape[1] * c.shape[2], -1)[:, ::5, ::3]
        data, mask = cases.cvx_inputs()
        out["cvx_upsample"] = se3_field.cvx_upsample(data, mask)[:, ::3, ::5]
        depth, K, coords = cases.proj_inputs()
        X = pops.inv_project(depth, K)
        out["inv_project"] = X
        out["project"] = pops.project(X, K)
        out["depth_sampler"] = depth_sampler(depth, coords)[0]
        img, disp = cases.warp_inputs()
        out["disp_warp"] = disp_warp(img, disp, padding_mode="zeros")[0]
        # ---- evaluation metrics: the reference's own inference(evaluate=True) on a stereo-only model ----
        img, r_img, gt, flow, meta = cases.metric_case()
        est.reset_inference_state()
        res = est.inference(img, r_img, meta, evaluate=True, gt_disp=[gt], gt_flow=[flow])
        est2 = est.inference(img, r_img, meta, evaluate=False)
        out["metric_pred_disp"] = est2
        out["metric_names"] = None
        names = [k for k in res.keys()]
        out["metric_values"] = torch.stack([res[k].double().reshape(()) for k in names]).float()
        globals()["_METRIC_NAMES"] = names
        # KITTI-style ground truth: no disparity on frame 1 (dummy-mask branch), gt_disp2, gt_disp_occ
        img, r_img, gtk, flowk, gt2k, occk, meta = cases.kitti_metric_case()
        est.reset_inference_state()
        resk = est.inference(img, r_img, meta, evaluate=True, gt_disp=[gtk], gt_flow=[flowk], gt_disp2=[gt2k],
                             gt_disp_occ=[occk])
        out["metric_kitti_values"] = torch.stack([resk[k].double().reshape(()) for k in names]).float()
        # ---- scene-flow columns: the reference's own calc_metric (model/codd.py:435-575) driven frame by frame ----
        # Ts is a lietorch.SE3 in the reference (un-vendored): the stand-in below supplies only indexing and the
        # action T * X (oracle/se3.py); masks, depth clipping, induced_flow, BF scaling and the sums are the reference's.
        from oracle import se3 as ose3

        class SE3Field:
            def __init__(self, data):
                self.data = data

            def __getitem__(self, idx):
                return SE3Field(self.data[idx])

            def __mul__(self, X):
                return ose3.act(self.data, X)

        sc = cases.sceneflow_case()
        h, w = sc["h"], sc["w"]
        for tag, with_occ in (("sf", False), ("sfocc", True)):
            est.reset_inference_state()
            st = est.inference_state
            for f in range(sc["pred"].shape[1]):
                st["gt_disp"].append(sc["gt"][:, f, :, :h, :w])
                st["gt_flow"].append(sc["flow"][:, f, :, :h, :w])
                st["gt_disp_change"].append(sc["dchange"][:, f, :, :h, :w])
                if with_occ:
                    st["gt_flow_occ"].append(sc["occ"][:, f, :, :h, :w])
                st["pred_disp"].append(sc["pred"][:, f])
                est.calc_metric(f, sc["pred"][:, f, :, :h, :w], st["gt_disp"][-1], sc["meta"], h, w,
                                Ts=SE3Field(sc["Ts"][:, f]) if f > 0 else None)
            res = collect_metric(st)
            out[f"metric_{tag}_values"] = torch.stack([torch.as_tensor(res[k]).double().reshape(()) for k in names]).float()
        # ---- ablation plug-ins (GT / Kalman) ---------------------------------------------------------
        from model.fusion.others import GTFusion, KalmanFusion, NullFusion  # noqa: E402
        from model.motion.others import GTMotion  # noqa: E402
        c = cases.ablation_case()
        mem5 = [c["img_prev"], c["feat_prev"], torch.ones(1, 3, 32, 48), c["warp"], torch.zeros(1, 3, 32, 48)]
        kf = KalmanFusion()
        kf.memory_query(dict(pred_disp=c["pred"].clone()), {})  # first frame: resets P
        o = dict(pred_disp=c["pred"].clone())
        kf.memory_query(o, dict(memory=mem5))
        out["ablation_kalman"] = o["pred_disp"]
        o = dict(pred_disp=c["pred"].clone())
        GTFusion().memory_query(o, dict(memory=mem5, gt_disp=[c["gt"]]))
        out["ablation_gtfusion"] = o["pred_disp"]
        st = dict(memory=[c["img_prev"], c["feat_prev"], c["disp_prev"]], gt_disp_change=[c["gt_disp_change"]],
                  gt_flow=[c["gt_flow"]], gt_flow_occ=[c["gt_flow_occ"]])
        try:
            GTMotion()(st, {}, None)
        except Exception as e:  # the lietorch stub cannot build SE3.Identity; memory is written before that
            print("GTMotion tail:", type(e).__name__, e)
        for i, k in enumerate(("img", "feat", "conf", "disp", "flow")):
            out[f"ablation_gtmotion_{k}"] = st["memory"][i]
    path = os.path.join(HERE, "reference_outputs.npz")
    out.pop("metric_names")
    arrays = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in out.items()}
    arrays["metric_names"] = np.array(globals()["_METRIC_NAMES"])
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
