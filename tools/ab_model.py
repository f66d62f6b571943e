#!/usr/bin/env python
"""Same-box A/B of two builds of libwespeaker_amd.so on one model (box-to-box variance is +-5 %, so
every "X is n % faster" claim in DESIGN.md was measured this way, inside ONE gpurun call):

    cp wespeaker_amd/lib/libwespeaker_amd.so tools/bin/libws_old.so      # the build to compare against
    ... change a kernel, python -m wespeaker_amd.build ...
    python tools/ab_model.py ECAPA_TDNN_GLOB_c512 192 256 f16 old
    python tools/ab_model.py ECAPA_TDNN_GLOB_c512 192 256 f16 new

argv: model, embed_dim, batch, precision (fp32 | f16x3 | f16), old | new.  Environment switches
 select code paths inside one build instead."""
import sys, os, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
import wespeaker_amd._lib as L
if sys.argv[5] == "old":
    L.LIB_PATH = os.path.join(root, "tools/bin/libws_old.so")
import torch
from wespeaker_amd import Frontend, NativeSpeakerModel
from fixtures import synth
from bench import device_wavs
name, ed, batch, prec = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dev = torch.device("cuda:0")
fe = Frontend(16000, 80, device=dev)
sd = synth.synth_state_dict(name, 80, ed, seed=42)
m = NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=ed, device=dev, max_batch=batch, max_frames=198)
wav = device_wavs(batch, 32000, dev, 0)
m.set_precision(prec)
for _ in range(3): m.extract(fe, wav)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): m.extract(fe, wav)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
print("%s %s %s: emb/s %.0f  ms %.3f" % (name, prec, sys.argv[5], batch/dt, dt*1e3))
