#!/usr/bin/env python
"""Cycle stamps of cam_dense_layer_kernel inside the real CAM++ forward (trace build: tools/build_trace_lib.sh).

Stamps of wavefronts 0 and 3 of workgroup 9, per input width C_i:
  0 entry | 1 first K-tile staged (after the barrier) | 2 K loop done | 3 after the barrier behind it |
  4 h in LDS (+ first k3 weights requested) | 5 mask ready | 6 exit
"""
import sys, os, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import wespeaker_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "tools/bin/libws_trace.so")
import torch
from wespeaker_amd import Frontend, NativeSpeakerModel
from fixtures import synth
from bench import device_wavs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
fe = Frontend(16000, 80, device=dev)
sd = synth.synth_state_dict("CAMPPlus", 80, 512, seed=42)
m = NativeSpeakerModel("CAMPPlus", sd, feat_dim=80, embed_dim=512, device=dev, max_batch=B, max_frames=198)
wav = device_wavs(B, 32000, dev, 0)
for _ in range(4):
    m.extract(fe, wav)
torch.cuda.synchronize()
f = getattr(L.lib(), "_ZN5wsamd24cam_trace_buffer_addressEv"); f.restype = ctypes.c_void_p
buf = (ctypes.c_ulonglong * 2048)()
hip = ctypes.CDLL("libamdhip64.so")
r = hip.hipMemcpy(buf, ctypes.c_void_p(f()), 16384, 2)
t = list(buf)
print("hipMemcpy", r, "batch", B)
print("cin   nk | wave0: start  Kloop (per K-tile)  bar  h->LDS  mask  conv  total | wave3: Kloop conv total")
fine = []
for c in range(32):
    a, b = t[c * 64: c * 64 + 32], t[c * 64 + 32: c * 64 + 64]
    if not a[6]:
        continue
    cin = c * 32 if c else 1024
    nk = cin // 32
    print("%4d %4d | %6d %7d (%5d) %5d %6d %6d %6d %7d | %7d %6d %7d" % (
        cin, nk, a[1] - a[0], a[2] - a[1], (a[2] - a[1]) // nk, a[3] - a[2], a[4] - a[3], a[5] - a[4], a[6] - a[5],
        a[6] - a[0], b[2] - b[1], b[6] - b[5], b[6] - b[0]))
    fine.append("%4d | h: bias %5d blocks %5d tail+halo %5d k3 issue %5d barrier %5d; wave3: %5d %5d %5d %5d %5d |" % (
        cin, a[16] - a[3], a[17] - a[16], a[18] - a[17], a[19] - a[18], a[4] - a[19],
        b[16] - b[3], b[17] - b[16], b[18] - b[17], b[19] - b[18], b[4] - b[19]))
    fine.append("%4d | h: regs->LDS %5d, k3 weights + barrier %5d | mask: sums %5d bar %5d ctx %5d fc1 %5d bar %5d fc2 %5d bar %5d"
                " | k3: taps %5d %5d %5d store %5d" % (
        cin, a[7] - a[3], a[4] - a[7], a[8] - a[4], 0, a[9] - a[8], a[10] - a[9], a[11] - a[10], a[12] - a[11],
        a[5] - a[12], a[13] - a[5], a[14] - a[13], a[15] - a[14], a[6] - a[15]))
print("\n".join(fine))
