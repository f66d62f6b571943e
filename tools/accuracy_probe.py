"""Accuracy of the two GEMM back-ends against a float64 evaluation of the same network."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ecapa as oecapa, fbank as ofbank
from wespeaker_amd import NativeSpeakerModel
from fixtures import synth

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)

for name in ("ECAPA_TDNN_GLOB_c512", "ECAPA_TDNN_GLOB_c1024"):
    sd = synth.synth_ecapa_state_dict(name, 80, 192, seed=42)
    feats = np.stack([ofbank.speaker_features(synth.synth_wav(i)) for i in range(8)])
    ref64 = oecapa.ecapa_forward(sd, feats, dtype=torch.float64).numpy()
    ref32 = oecapa.ecapa_forward(sd, feats).numpy()
    m = NativeSpeakerModel(name, sd, max_batch=8, max_frames=200)
    e32 = m(torch.from_numpy(feats))[-1].cpu().numpy()
    m.set_precision("f16x3")
    e16 = m(torch.from_numpy(feats))[-1].cpu().numpy()
    cos = lambda a, b: 1 - np.sum(a * b, -1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1)
    print(name)
    print("  fp32-MFMA  vs torch-fp32 oracle: rel %.2e  1-cos %.2e" % (rel(e32, ref32).max(), cos(e32.astype(np.float64), ref32.astype(np.float64)).max()))
    print("  f16x3-MFMA vs torch-fp32 oracle: rel %.2e  1-cos %.2e" % (rel(e16, ref32).max(), cos(e16.astype(np.float64), ref32.astype(np.float64)).max()))
    print("  f16x3 vs fp32-MFMA             : rel %.2e" % rel(e16, e32).max())
    print("  vs float64 ground truth: torch-fp32 CPU %.2e | fp32-MFMA %.2e | f16x3-MFMA %.2e"
          % (rel(ref32, ref64).max(), rel(e32, ref64).max(), rel(e16, ref64).max()))
