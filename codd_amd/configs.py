"""Model configs mirroring the reference's (configs/models/codd.py:18-101, stereo.py:12-39)."""
import copy

MAX_DISP = 320

STEREO = dict(type="HITNetMF", backbone=dict(type="HITUNet"),
              initialization=dict(type="TileInitialization", max_disp=MAX_DISP),
              propagation=dict(type="TilePropagation"))

HRNET = dict(type="HRNet", norm_cfg=dict(type="SyncBN", requires_grad=False), norm_eval=True,
             extra=dict(stage1=dict(num_modules=1, num_branches=1, block="BOTTLENECK", num_blocks=(2,),
                                    num_channels=(64,)),
                        stage2=dict(num_modules=1, num_branches=2, block="BASIC", num_blocks=(2, 2),
                                    num_channels=(18, 36)),
                        stage3=dict(num_modules=3, num_branches=3, block="BASIC", num_blocks=(2, 2, 2),
                                    num_channels=(18, 36, 72)),
                        stage4=dict(num_modules=2, num_branches=4, block="BASIC", num_blocks=(2, 2, 2, 2),
                                    num_channels=(18, 36, 72, 144))))


def stereo_only():
    return dict(type="ConsistentOnlineDynamicDepth", stereo=copy.deepcopy(STEREO), test_cfg=dict(mode="whole"))


def codd(iters=16):
    return dict(type="ConsistentOnlineDynamicDepth", stereo=copy.deepcopy(STEREO),
                motion=dict(type="Motion", iters=iters, raft3d=dict(type="RAFT3D", cnet_cfg=copy.deepcopy(HRNET))),
                fusion=dict(type="Fusion", in_channels=24, fusion_channel=32,
                            corr_cfg=dict(type="px2patch", patch_size=3)),
                test_cfg=dict(mode="whole"))
