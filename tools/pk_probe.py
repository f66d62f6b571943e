"""which packed-fp32 instruction form goes wrong next to a binary16 engine on another stream?
Runs tools/pk_probe.hip (one form per workgroup, results checked in the kernel) on stream 1 while the partner engine
runs forwards on stream 2.   python tools/pk_probe.py [out.json]"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from bench import device_wavs
from fixtures import synth
from wespeaker_amd.engine import Frontend, NativeSpeakerModel

FORMS = ['v_pk_mul_f32 op_sel:[0,0] op_sel_hi:[0,0]', 'v_pk_mul_f32 op_sel:[0,0] op_sel_hi:[0,1]', 'v_pk_mul_f32 op_sel:[0,0] op_sel_hi:[1,0]', 'v_pk_mul_f32 op_sel:[0,0] op_sel_hi:[1,1]', 'v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0]', 'v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,1]', 'v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]', 'v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1]', 'v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,0]', 'v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]', 'v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[1,0]', 'v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[1,1]', 'v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,0]', 'v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,1]', 'v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[1,0]', 'v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[1,1]', 'v_pk_add_f32 op_sel:[0,1] op_sel_hi:[0,0]', 'v_pk_add_f32 op_sel:[0,0] op_sel_hi:[0,0]', 'v_pk_add_f32 op_sel:[1,1] op_sel_hi:[0,0]', 'v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]', 'v_pk_add_f32 op_sel:[0,0] op_sel_hi:[1,1]', 'v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 op_sel:[0,0,0] op_sel_hi:[0,0,0]', 'v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[0,0,0]', 'v_pk_fma_f32 op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 op_sel:[0,0,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 op_sel:[0,1,1] op_sel_hi:[0,0,0]', 'v_pk_mov_b32 op_sel:[0,1]', 'v_pk_mov_b32 op_sel:[1,0]', 'v_pk_mul_f16 op_sel:[0,1] op_sel_hi:[1,0]', 'v_pk_add_f16 op_sel:[0,1] op_sel_hi:[1,0]', 'v_pk_fma_f16 op_sel:[0,1,0] op_sel_hi:[1,0,1]', 'v_pk_mul_f16 (plain)']
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pk_probe.json"
lib = ctypes.CDLL(os.path.join(os.getcwd(), "tools/bin/libpk_probe.so"))
lib.pk_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
fe = Frontend(16000, 80)
w = device_wavs(64, 32000, dev, 40)
feats = fe.fbank(w, cmn=True)
P = NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", synth.synth_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=12),
                       feat_dim=80, embed_dim=192, max_batch=64, max_frames=198)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = torch.zeros(len(FORMS) * 64, dtype=torch.int32, device=dev)
report = []
for prec in (None, "f16x3", "fp32"):
    if prec:
        P.set_precision(prec)
    for mode in (0, 1):
        bad.zero_()
        torch.cuda.synchronize()
        for rep in range(40):
            if prec:
                with torch.cuda.stream(s2):
                    P.embed(feats)
            with torch.cuda.stream(s1):
                for _ in range(6):
                    assert lib.pk_probe_launch(mode, 1536 * 3, 400, bad.data_ptr(), s1.cuda_stream) == 0
        torch.cuda.synchronize()
        b = bad.view(len(FORMS), 64).cpu()
        rec = {"partner": prec, "operands": "lds" if mode else "registers",
               "mismatches_by_form": {FORMS[i]: int(b[i].sum()) for i in range(len(FORMS)) if int(b[i].sum())},
               "forms_clean": sum(1 for i in range(len(FORMS)) if not int(b[i].sum())),
               "mismatches_by_lane_group": [int(b[:, g * 16:(g + 1) * 16].sum()) for g in range(4)]}
        report.append(rec)
        print(json.dumps(rec), flush=True)
# which instruction class of the partner does it?  tools/mfma_burner.hip kinds next to the same probe
bl = ctypes.CDLL(os.path.join(os.getcwd(), "tools/bin/libmfma_burner.so"))
bl.mfma_burner_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
outb = torch.zeros(4096 * 256, device=dev)
KINDS = ["v_mfma_f32_32x32x2_f32", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x32_f16", "cvt to f16 denormals",
         "cvt to f16 normals", "16x16x32_f16 on denormals", "hi/lo split, denormal lo", "v_cvt_f32_f16_sdwa",
         "LDS write_b64/read_b128", "v_cvt_pk_f16_f32 + packed fp32"]
for kind, name in enumerate(KINDS):
    bad.zero_()
    torch.cuda.synchronize()
    for rep in range(20):
        with torch.cuda.stream(s2):
            bl.mfma_burner_launch(kind, 1024, 3000, outb.data_ptr(), s2.cuda_stream)
        with torch.cuda.stream(s1):
            for _ in range(6):
                lib.pk_probe_launch(0, 1536 * 3, 400, bad.data_ptr(), s1.cuda_stream)
    torch.cuda.synchronize()
    b = bad.view(len(FORMS), 64).cpu()
    rec = {"partner": "burner: " + name, "operands": "registers", "mismatches_total": int(b.sum()),
           "mismatches_by_lane_group": [int(b[:, g * 16:(g + 1) * 16].sum()) for g in range(4)]}
    report.append(rec)
    print(json.dumps(rec), flush=True)
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(report, open(out_path, "w"), indent=1)
