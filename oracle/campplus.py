"""ORACLE (test infrastructure, never shipped): CPU fp32 restatement of the reference's CAM++
forward as one straight-line function over a state_dict.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows (file:line in /root/reference):
  wespeaker/models/campplus.py:409-413  CAMPPlus.forward ((B,T,F) -> (B,F,T) -> head -> xvector)
  wespeaker/models/campplus.py:282-330  FCM (conv1+bn+relu, 2x2 BasicResBlocks stride (2,1), conv2 stride (2,1), reshape (B, C*F', T))
  wespeaker/models/campplus.py:245-279  BasicResBlock
  wespeaker/models/campplus.py:55-83    TDNNLayer (conv1d k5 stride 2 pad 2 -> BN -> ReLU)
  wespeaker/models/campplus.py:138-170  CAMDenseTDNNLayer (BN-ReLU -> 1x1 -> BN-ReLU -> CAMLayer)
  wespeaker/models/campplus.py:86-135   CAMLayer (local conv * sigmoid(W2 relu(W1 (mean + segmean_100))))
  wespeaker/models/campplus.py:173-201  dense block: x = cat([x, layer(x)])
  wespeaker/models/campplus.py:204-238  TransitLayer, DenseLayer
PINNED by tests/golden/campplus_ref.npz (outputs of the reference's own nn.Module).
"""
import torch
import torch.nn.functional as F

from .ecapa import _bn, _t, tstp

BLOCKS = ((12, 1), (24, 2), (16, 2))     # (layers, dilation); kernel 3, growth 32, bn_channels 128


def _bn_relu(sd, prefix, x):
    return F.relu(_bn(sd, prefix + ".batchnorm", x))


def _res_block(sd, p, x, stride):
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, _t(sd, p + ".conv1.weight"), None,
                                              stride=(stride, 1), padding=1)))
    out = _bn(sd, p + ".bn2", F.conv2d(out, _t(sd, p + ".conv2.weight"), None, padding=1))
    sc = x
    if p + ".shortcut.0.weight" in sd:
        sc = _bn(sd, p + ".shortcut.1", F.conv2d(x, _t(sd, p + ".shortcut.0.weight"), None,
                                                 stride=(stride, 1)))
    return F.relu(out + sc)


def _seg_pooling(x, seg_len=100):
    seg = F.avg_pool1d(x, kernel_size=seg_len, stride=seg_len, ceil_mode=True)
    shape = seg.shape
    seg = seg.unsqueeze(-1).expand(shape[0], shape[1], shape[2], seg_len).reshape(shape[0], shape[1], -1)
    return seg[..., :x.shape[-1]]


def _cam_layer(sd, p, x, dilation):
    y = F.conv1d(x, _t(sd, p + ".linear_local.weight"), None, padding=dilation, dilation=dilation)
    ctx = x.mean(-1, keepdim=True) + _seg_pooling(x)
    ctx = F.relu(F.conv1d(ctx, _t(sd, p + ".linear1.weight"), _t(sd, p + ".linear1.bias")))
    m = torch.sigmoid(F.conv1d(ctx, _t(sd, p + ".linear2.weight"), _t(sd, p + ".linear2.bias")))
    return y * m


@torch.no_grad()
def campplus_forward(sd, feats):
    """feats (B, T, F) float32 -> (B, E)."""
    x = torch.as_tensor(feats, dtype=torch.float32).permute(0, 2, 1).unsqueeze(1)
    out = F.relu(_bn(sd, "head.bn1", F.conv2d(x, _t(sd, "head.conv1.weight"), None, padding=1)))
    for layer in ("head.layer1", "head.layer2"):
        out = _res_block(sd, layer + ".0", out, 2)
        out = _res_block(sd, layer + ".1", out, 1)
    out = F.relu(_bn(sd, "head.bn2", F.conv2d(out, _t(sd, "head.conv2.weight"), None, stride=(2, 1),
                                              padding=1)))
    x = out.reshape(out.shape[0], out.shape[1] * out.shape[2], out.shape[3])
    x = F.conv1d(x, _t(sd, "xvector.tdnn.linear.weight"), None, stride=2, padding=2)
    x = _bn_relu(sd, "xvector.tdnn.nonlinear", x)
    for k, (layers, dil) in enumerate(BLOCKS):
        for j in range(layers):
            p = "xvector.block%d.tdnnd%d" % (k + 1, j + 1)
            h = F.conv1d(_bn_relu(sd, p + ".nonlinear1", x), _t(sd, p + ".linear1.weight"), None)
            h = _cam_layer(sd, p + ".cam_layer", _bn_relu(sd, p + ".nonlinear2", h), dil)
            x = torch.cat([x, h], dim=1)
        p = "xvector.transit%d" % (k + 1)
        x = F.conv1d(_bn_relu(sd, p + ".nonlinear", x), _t(sd, p + ".linear.weight"), None)
    x = _bn_relu(sd, "xvector.out_nonlinear", x)
    stats = tstp(x)
    emb = F.conv1d(stats.unsqueeze(-1), _t(sd, "xvector.dense.linear.weight"), None).squeeze(-1)
    return _bn(sd, "xvector.dense.nonlinear.batchnorm", emb)
