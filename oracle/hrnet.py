"""ORACLE (test infrastructure, not product): RAFT3D context network.

``cnet = Sequential(mmseg HRNet, ResizeConcatConv)`` (reference raft3d.py:152-160, 109-137; HRNet
config configs/models/codd.py:44-74 = HRNetV2-W18-small-v2, BatchNorm in eval mode).

mmseg is an un-vendored, un-pinned dependency (reference README.md:42): PARITY UNPINNED.  The
architecture and the state-dict key names below follow mmsegmentation 0.x
``mmseg/models/backbones/hrnet.py`` (conv1/bn1/conv2/bn2, layer1, transition{1,2,3},
stage{2,3,4}.<module>.branches.<b>.<blk>.{conv1,bn1,conv2,bn2},
stage*.fuse_layers.<i>.<j>...) and are validated by the shape / channel contract
[18, 36, 72, 144] @ 1/4 .. 1/32 and by BN-folding equivalence in tests.
"""
import torch
import torch.nn.functional as F

STAGES = dict(  # configs/models/codd.py:48-73
    stage2=dict(num_modules=1, channels=(18, 36), num_blocks=2),
    stage3=dict(num_modules=3, channels=(18, 36, 72), num_blocks=2),
    stage4=dict(num_modules=2, channels=(18, 36, 72, 144), num_blocks=2),
)


def _cbn(sd, ck, bk, x, stride=1, pad=0, relu=False):
    x = F.conv2d(x, sd[ck + ".weight"], None, stride, pad)
    x = F.batch_norm(x, sd[bk + ".running_mean"], sd[bk + ".running_var"], sd[bk + ".weight"],
                     sd[bk + ".bias"], False, 0.0, 1e-5)
    return F.relu(x) if relu else x


def _bottleneck(sd, p, x, has_down):
    y = _cbn(sd, p + ".conv1", p + ".bn1", x, relu=True)
    y = _cbn(sd, p + ".conv2", p + ".bn2", y, 1, 1, relu=True)
    y = _cbn(sd, p + ".conv3", p + ".bn3", y)
    if has_down:
        x = _cbn(sd, p + ".downsample.0", p + ".downsample.1", x)
    return F.relu(y + x)


def _basic(sd, p, x):
    y = _cbn(sd, p + ".conv1", p + ".bn1", x, 1, 1, relu=True)
    y = _cbn(sd, p + ".conv2", p + ".bn2", y, 1, 1)
    return F.relu(y + x)


def _hr_module(sd, p, xs, num_blocks):
    nb = len(xs)
    xs = list(xs)
    for i in range(nb):
        for k in range(num_blocks):
            xs[i] = _basic(sd, f"{p}.branches.{i}.{k}", xs[i])
    outs = []
    for i in range(nb):
        y = 0
        for j in range(nb):
            if j == i:
                y = y + xs[j]
            elif j > i:
                t = _cbn(sd, f"{p}.fuse_layers.{i}.{j}.0", f"{p}.fuse_layers.{i}.{j}.1", xs[j])
                y = y + F.interpolate(t, size=xs[i].shape[2:], mode="bilinear", align_corners=False)
            else:
                t = xs[j]
                for k in range(i - j):
                    q = f"{p}.fuse_layers.{i}.{j}.{k}"
                    t = _cbn(sd, q + ".0", q + ".1", t, 2, 1, relu=(k != i - j - 1))
                y = y + t
        outs.append(F.relu(y))
    return outs


def hrnet(sd, p, x):
    x = _cbn(sd, p + ".conv1", p + ".bn1", x, 2, 1, relu=True)
    x = _cbn(sd, p + ".conv2", p + ".bn2", x, 2, 1, relu=True)
    x = _bottleneck(sd, p + ".layer1.0", x, True)
    x = _bottleneck(sd, p + ".layer1.1", x, False)
    ys = [_cbn(sd, p + ".transition1.0.0", p + ".transition1.0.1", x, 1, 1, relu=True),
          _cbn(sd, p + ".transition1.1.0.0", p + ".transition1.1.0.1", x, 2, 1, relu=True)]
    for m in range(STAGES["stage2"]["num_modules"]):
        ys = _hr_module(sd, f"{p}.stage2.{m}", ys, 2)
    ys = ys + [_cbn(sd, p + ".transition2.2.0.0", p + ".transition2.2.0.1", ys[-1], 2, 1, relu=True)]
    for m in range(STAGES["stage3"]["num_modules"]):
        ys = _hr_module(sd, f"{p}.stage3.{m}", ys, 2)
    ys = ys + [_cbn(sd, p + ".transition3.3.0.0", p + ".transition3.3.0.1", ys[-1], 2, 1, relu=True)]
    for m in range(STAGES["stage4"]["num_modules"]):
        ys = _hr_module(sd, f"{p}.stage4.{m}", ys, 2)
    return ys


def cnet(sd, p, image):
    """reference raft3d.py:155-158 + ResizeConcatConv :109-137 -> [B,512,H/8,W/8]."""
    ys = hrnet(sd, p + ".0", image)
    size = ys[1].shape[2:]
    cat = torch.cat([F.interpolate(y, size=size, mode="bilinear", align_corners=True) for y in ys], 1)
    return F.relu(F.conv2d(cat, sd[p + ".1.convs.0.weight"]))


def state_dict_spec(p):
    """(name, shape) list of the cnet state dict (mmseg naming), for the synthetic filler."""
    spec = []

    def cv(k, co, ci, ks):
        spec.append((k + ".weight", (co, ci, ks, ks)))

    def bn(k, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            spec.append((f"{k}.{leaf}", (c,)))
        spec.append((k + ".num_batches_tracked", ()))

    h = p + ".0"
    cv(h + ".conv1", 64, 3, 3); bn(h + ".bn1", 64); cv(h + ".conv2", 64, 64, 3); bn(h + ".bn2", 64)
    for b, cin in ((0, 64), (1, 256)):
        q = f"{h}.layer1.{b}"
        cv(q + ".conv1", 64, cin, 1); bn(q + ".bn1", 64); cv(q + ".conv2", 64, 64, 3); bn(q + ".bn2", 64)
        cv(q + ".conv3", 256, 64, 1); bn(q + ".bn3", 256)
        if b == 0:
            cv(q + ".downsample.0", 256, 64, 1); bn(q + ".downsample.1", 256)
    cv(h + ".transition1.0.0", 18, 256, 3); bn(h + ".transition1.0.1", 18)
    cv(h + ".transition1.1.0.0", 36, 256, 3); bn(h + ".transition1.1.0.1", 36)
    cv(h + ".transition2.2.0.0", 72, 36, 3); bn(h + ".transition2.2.0.1", 72)
    cv(h + ".transition3.3.0.0", 144, 72, 3); bn(h + ".transition3.3.0.1", 144)
    for st, cfg in STAGES.items():
        ch = cfg["channels"]
        for m in range(cfg["num_modules"]):
            q = f"{h}.{st}.{m}"
            for i, c in enumerate(ch):
                for k in range(cfg["num_blocks"]):
                    r = f"{q}.branches.{i}.{k}"
                    cv(r + ".conv1", c, c, 3); bn(r + ".bn1", c); cv(r + ".conv2", c, c, 3); bn(r + ".bn2", c)
            for i in range(len(ch)):
                for j in range(len(ch)):
                    if j > i:
                        cv(f"{q}.fuse_layers.{i}.{j}.0", ch[i], ch[j], 1); bn(f"{q}.fuse_layers.{i}.{j}.1", ch[i])
                    elif j < i:
                        for k in range(i - j):
                            co = ch[i] if k == i - j - 1 else ch[j]
                            cv(f"{q}.fuse_layers.{i}.{j}.{k}.0", co, ch[j], 3); bn(f"{q}.fuse_layers.{i}.{j}.{k}.1", co)
    spec.append((p + ".1.convs.0.weight", (512, sum(STAGES["stage4"]["channels"]), 1, 1)))
    return spec
