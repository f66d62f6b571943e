"""Open issue probe: the fbank kernel on one stream, a pure MFMA burner (registers only: no LDS, no
memory) on another.  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/mfma_burner.hip -o tools/bin/libmfma_burner.so
   python tools/burner_vs_fbank.py"""
import sys, os, ctypes, torch
sys.path.insert(0, os.getcwd())
from bench import device_wavs
from wespeaker_amd.engine import Frontend
lib = ctypes.CDLL(os.path.join(os.getcwd(), "tools/bin/libmfma_burner.so"))
lib.mfma_burner_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
fe = Frontend(16000, 80)
w = device_wavs(64, 32000, dev, 40)
ref = fe.fbank(w, cmn=False).clone(); torch.cuda.synchronize()
out = torch.zeros(4096 * 256, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for kind, name in ((0, "v_mfma_f32_32x32x2_f32"), (1, "v_mfma_f32_32x32x16_f16"), (2, "v_mfma_f32_16x16x32_f16"),
                   (3, "cvt to f16 denormals"), (4, "cvt to f16 normals"), (5, "16x16x32_f16 on denormals"),
                   (6, "hi/lo split, denormal lo"), (7, "v_cvt_f32_f16_sdwa"), (8, "LDS write_b64/read_b128"),
                   (9, "v_cvt_pk_f16_f32 + packed fp32")):
    for grid in (1024,):
        bad = 0
        for rep in range(30):
            with torch.cuda.stream(s2):
                lib.mfma_burner_launch(kind, grid, 3000, out.data_ptr(), s2.cuda_stream)
            with torch.cuda.stream(s1):
                outs = [fe.fbank(w, cmn=False) for _ in range(4)]
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
        print("burner %-26s grid %4d: fbank differs in %3d of 120 launches" % (name, grid, bad))
