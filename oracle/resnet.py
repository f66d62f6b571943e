"""ORACLE (test infrastructure, never shipped): CPU fp32 restatement of the reference's ResNet
speaker-embedding forward as one straight-line function over a state_dict.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows (file:line in /root/reference):
  wespeaker/models/resnet.py:171-182  _get_frame_level_feat  ((B,T,F) -> (B,1,F,T), conv1+bn1+relu, layer1-4)
  wespeaker/models/resnet.py:63-69    BasicBlock.forward   (conv-bn-relu, conv-bn, += shortcut, relu)
  wespeaker/models/resnet.py:101-107  Bottleneck.forward   (1x1, 3x3 (stride here), 1x1 x4, += shortcut, relu)
  wespeaker/models/resnet.py:163-169  _make_layer (first block of a stage carries the stride)
  wespeaker/models/resnet.py:192-204  forward: TSTP -> seg_1 [-> relu -> seg_bn_1 -> seg_2]
  wespeaker/models/pooling_layers.py:78-85 TSTP
PINNED by tests/golden/resnet_ref.npz (outputs of the reference's own nn.Modules).
"""
import torch
import torch.nn.functional as F

from .ecapa import _bn, _t, tstp

LAYOUTS = {            # name -> (bottleneck?, blocks per stage)   resnet.py:207-260
    "ResNet18": (False, [2, 2, 2, 2]), "ResNet34": (False, [3, 4, 6, 3]),
    "ResNet50": (True, [3, 4, 6, 3]), "ResNet101": (True, [3, 4, 23, 3]),
    "ResNet152": (True, [3, 8, 36, 3]), "ResNet221": (True, [6, 16, 48, 3]),
    "ResNet293": (True, [10, 20, 64, 3]),
}


def _conv_bn(sd, conv, bn, x, stride=1, padding=0):
    return _bn(sd, bn, F.conv2d(x, _t(sd, conv + ".weight"), None, stride=stride, padding=padding))


def _block(sd, p, x, stride, bottleneck):
    if bottleneck:
        out = F.relu(_conv_bn(sd, p + ".conv1", p + ".bn1", x))
        out = F.relu(_conv_bn(sd, p + ".conv2", p + ".bn2", out, stride=stride, padding=1))
        out = _conv_bn(sd, p + ".conv3", p + ".bn3", out)
    else:
        out = F.relu(_conv_bn(sd, p + ".conv1", p + ".bn1", x, stride=stride, padding=1))
        out = _conv_bn(sd, p + ".conv2", p + ".bn2", out, padding=1)
    sc = x
    if p + ".shortcut.0.weight" in sd:
        sc = _conv_bn(sd, p + ".shortcut.0", p + ".shortcut.1", x, stride=stride)
    return F.relu(out + sc)


@torch.no_grad()
def resnet_forward(sd, feats, model_name="ResNet34"):
    """feats (B, T, F) float32 -> embedding (B, E): embed_a, or embed_b when the state_dict has
    seg_2.* (two_emb_layer=True; callers take outputs[-1])."""
    bottleneck, blocks = LAYOUTS[model_name]
    x = torch.as_tensor(feats, dtype=torch.float32).permute(0, 2, 1).unsqueeze(1)
    out = F.relu(_conv_bn(sd, "conv1", "bn1", x, padding=1))
    for s, nb in enumerate(blocks):
        for b in range(nb):
            stride = (1 if s == 0 else 2) if b == 0 else 1
            out = _block(sd, "layer%d.%d" % (s + 1, b), out, stride, bottleneck)
    stats = tstp(out)
    emb = F.linear(stats, _t(sd, "seg_1.weight"), _t(sd, "seg_1.bias"))
    if "seg_2.weight" in sd:
        o = F.relu(emb)
        o = F.batch_norm(o, _t(sd, "seg_bn_1.running_mean"), _t(sd, "seg_bn_1.running_var"), None,
                         None, False, 0.0, 1e-5)
        emb = F.linear(o, _t(sd, "seg_2.weight"), _t(sd, "seg_2.bias"))
    return emb
