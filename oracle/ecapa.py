"""ORACLE (test infrastructure, never shipped): CPU fp32 restatement of the reference's
ECAPA-TDNN forward, written as one straight-line function over a state_dict.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows (file:line in /root/reference):
  wespeaker/models/ecapa_tdnn.py:208-234  ECAPA_TDNN._get_frame_level_feat / forward
  wespeaker/models/ecapa_tdnn.py:85-106   Conv1dReluBn   = conv -> ReLU -> BN   (NOT conv-BN-ReLU)
  wespeaker/models/ecapa_tdnn.py:58-78    Res2Conv1dReluBn.forward (7 serial k=3 convs, last split passed through)
  wespeaker/models/ecapa_tdnn.py:120-126  SE_Connect.forward (plain mean over T)
  wespeaker/models/ecapa_tdnn.py:156-157  SE_Res2Block.forward (x + block(x))
  wespeaker/models/pooling_layers.py:119-144  ASTP.forward

PINNED: tests/test_oracle_golden.py compares this function with golden vectors generated in
this container from the reference's own nn.Modules (oracle/make_golden.py imports them from
/root/reference); see tests/golden/ecapa_*.npz.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm1d default


def _t(sd, key):
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.as_tensor(v)


def _bn(sd, prefix, x):
    return F.batch_norm(x, _t(sd, prefix + ".running_mean"), _t(sd, prefix + ".running_var"),
                        _t(sd, prefix + ".weight") if prefix + ".weight" in sd else None,
                        _t(sd, prefix + ".bias") if prefix + ".bias" in sd else None,
                        training=False, eps=BN_EPS)


def _conv_relu_bn(sd, prefix, x, padding=0, dilation=1):
    y = F.conv1d(x, _t(sd, prefix + ".conv.weight"), _t(sd, prefix + ".conv.bias"),
                 padding=padding, dilation=dilation)
    return _bn(sd, prefix + ".bn", F.relu(y))


def _res2(sd, prefix, x, dilation, scale=8):
    width = x.shape[1] // scale
    spx = torch.split(x, width, 1)
    out = []
    sp = spx[0]
    for i in range(scale - 1):
        if i >= 1:
            sp = sp + spx[i]
        sp = F.conv1d(sp, _t(sd, "%s.convs.%d.weight" % (prefix, i)),
                      _t(sd, "%s.convs.%d.bias" % (prefix, i)),
                      padding=dilation, dilation=dilation)
        sp = _bn(sd, "%s.bns.%d" % (prefix, i), F.relu(sp))
        out.append(sp)
    out.append(spx[scale - 1])
    return torch.cat(out, dim=1)


def _se(sd, prefix, x):
    s = x.mean(dim=2)
    s = F.relu(F.linear(s, _t(sd, prefix + ".linear1.weight"), _t(sd, prefix + ".linear1.bias")))
    s = torch.sigmoid(F.linear(s, _t(sd, prefix + ".linear2.weight"), _t(sd, prefix + ".linear2.bias")))
    return x * s.unsqueeze(2)


def _se_res2block(sd, prefix, x, dilation):
    p = prefix + ".se_res2block"
    y = _conv_relu_bn(sd, p + ".0", x)
    y = _res2(sd, p + ".1", y, dilation)
    y = _conv_relu_bn(sd, p + ".2", y)
    y = _se(sd, p + ".3", y)
    return x + y


def astp(sd, prefix, x):
    """pooling_layers.py:119-144.  global_context_att is inferred from linear1's in_dim."""
    w1 = _t(sd, prefix + ".linear1.weight")
    glob = w1.shape[1] == 3 * x.shape[1]
    if glob:
        mean = x.mean(dim=-1, keepdim=True).expand_as(x)
        std = torch.sqrt(torch.var(x, dim=-1, keepdim=True) + 1e-7).expand_as(x)   # unbiased
        x_in = torch.cat((x, mean, std), dim=1)
    else:
        x_in = x
    a = torch.tanh(F.conv1d(x_in, w1, _t(sd, prefix + ".linear1.bias")))
    a = torch.softmax(F.conv1d(a, _t(sd, prefix + ".linear2.weight"),
                               _t(sd, prefix + ".linear2.bias")), dim=2)
    mean = torch.sum(a * x, dim=2)
    var = torch.sum(a * (x ** 2), dim=2) - mean ** 2
    std = torch.sqrt(var.clamp(min=1e-7))
    return torch.cat([mean, std], dim=1)


def tstp(x):
    """pooling_layers.py:78-85 (unbiased variance + 1e-7 inside the sqrt)."""
    mean = x.mean(dim=-1).flatten(start_dim=1)
    std = torch.sqrt(torch.var(x, dim=-1) + 1e-7).flatten(start_dim=1)
    return torch.cat((mean, std), 1)


@torch.no_grad()
def ecapa_forward(sd, feats, return_intermediates=False, dtype=torch.float32):
    """feats: (B, T, F) float32 (already CMN'd) -> embeddings (B, E) float32.

    sd: state_dict with the reference's key names (tensors or numpy arrays).
    dtype=torch.float64 evaluates the same network in double precision (a ground truth for
    judging fp32 rounding noise; the reference itself runs in float32)."""
    if dtype != torch.float32:
        sd = {k: (torch.as_tensor(v).to(dtype) if torch.as_tensor(v).is_floating_point()
                  else torch.as_tensor(v)) for k, v in sd.items()}
    x = torch.as_tensor(feats).to(dtype).permute(0, 2, 1)
    out1 = _conv_relu_bn(sd, "layer1", x, padding=2)
    out2 = _se_res2block(sd, "layer2", out1, 2)
    out3 = _se_res2block(sd, "layer3", out2, 3)
    out4 = _se_res2block(sd, "layer4", out3, 4)
    cat = torch.cat([out2, out3, out4], dim=1)
    h = F.relu(F.conv1d(cat, _t(sd, "conv.weight"), _t(sd, "conv.bias")))
    pooled = astp(sd, "pool", h)
    y = _bn(sd, "bn", pooled)
    emb = F.linear(y, _t(sd, "linear.weight"), _t(sd, "linear.bias"))
    if "bn2.running_mean" in sd:
        emb = _bn(sd, "bn2", emb)
    if return_intermediates:
        return emb, {"out1": out1, "out2": out2, "out3": out3, "out4": out4,
                     "h": h, "pooled": pooled}
    return emb
