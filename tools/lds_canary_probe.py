"""Open issue probe: an LDS canary kernel (tools/lds_canary.hip -> tools/bin/liblds_canary.so:
hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/lds_canary.hip -o tools/bin/liblds_canary.so) next to an engine:
does any engine kernel write outside its own LDS allocation?   python tools/lds_canary_probe.py ECAPA_TDNN_GLOB_c512 192 f16x3"""
import sys, os, ctypes, torch
sys.path.insert(0, os.getcwd())
from fixtures import synth
from bench import device_wavs
from wespeaker_amd.engine import Frontend, NativeSpeakerModel
lib = ctypes.CDLL(os.path.join(os.getcwd(), "tools/bin/liblds_canary.so"))
lib.lds_canary_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
pname, pE, pprec = sys.argv[1], int(sys.argv[2]), sys.argv[3]
lds = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
Bm = NativeSpeakerModel(pname, synth.synth_state_dict(pname, 80, pE, seed=12), feat_dim=80, embed_dim=pE, max_batch=64, max_frames=198)
Bm.set_precision(pprec)
fe2 = Frontend(16000, 80)
w = device_wavs(64, 32000, dev, 40)
feats = fe2.fbank(w, cmn=True)
bad = torch.zeros(1, dtype=torch.int32, device=dev); first = torch.full((1,), 1 << 30, dtype=torch.int32, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for rep in range(20):
    with torch.cuda.stream(s1):
        lib.lds_canary_launch(2048, lds, 400000, bad.data_ptr(), first.data_ptr(), s1.cuda_stream)
    with torch.cuda.stream(s2):
        for i in range(3):
            y = Bm.embed(feats) if os.environ.get("EMBED") == "1" else Bm.extract(fe2, w)
torch.cuda.synchronize()
print("partner", pname, pprec, "canary LDS", lds, "B: corrupted words", int(bad.item()), "first index", int(first.item()))
