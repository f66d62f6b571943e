#!/usr/bin/env python
"""s_memtime stamps of one tile of the fused attentive-pooling GEMM inside the real model.

Needs a trace build of the library (the product build has no stamps):

    mkdir -p /tmp/trobj && cd /tmp/trobj
    for f in $REPO/wespeaker_amd/csrc/*.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWS_TRACE -c $f -o $(basename $f).o; done
    hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/tools/bin/libws_trace.so *.o

The stamps of wavefront 0 of workgroup 8 are written to wsamd::g_trace (conv_gemm.hip, WS_EMARK / WS_STAMP /
WS_MARK) and read back here through the mangled accessor."""
import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import wespeaker_amd._lib as L
L.LIB_PATH = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/bin/libws_trace.so")
import torch
from wespeaker_amd import Frontend, NativeSpeakerModel
from fixtures import synth
from bench import device_wavs
dev = torch.device("cuda:0")
fe = Frontend(16000, 80, device=dev)
sd = synth.synth_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
m = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", sd, feat_dim=80, embed_dim=192, device=dev, max_batch=256, max_frames=198)
wav = device_wavs(256, 32000, dev, 0)
m.set_precision("f16")
for _ in range(5): m.extract(fe, wav)
torch.cuda.synchronize()
h = L.lib()
f = getattr(h, "_ZN5wsamd20trace_buffer_addressEv"); f.restype = ctypes.c_void_p
addr = f()
buf = (ctypes.c_ulonglong * 512)()
hip = ctypes.CDLL("libamdhip64.so")
r = hip.hipMemcpy(buf, ctypes.c_void_p(addr), 4096, 2)
t = list(buf)
e = t[62 * 8: 62 * 8 + 8]
print("hipMemcpy", r)
print("entry->loop0 %d" % (t[0] - e[6]))
print("kt0:", [t[i + 1] - t[i] for i in range(3)], "kt1:", [t[8 + i + 1] - t[8 + i] for i in range(3)], "kt0->kt1", t[8] - t[0])
print("loop end -> epi entry", e[0] - t[8 + 3])
print("epi: lds write+sync %d | pool main %d | sync %d | red write+sync %d | combine %d | exit %d" %
      (e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3], e[5] - e[4], e[7] - e[5]))
print("kernel total for this tile %d" % (e[7] - e[6]))
