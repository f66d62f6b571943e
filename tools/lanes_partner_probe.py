"""Open issue probe: engine A (model, precision) on one stream, a partner (matmul | copy | fp32 | f16 |
f16x3 engine of a second model) on another: do A's embeddings keep their serial bits?  EMBED=1 skips fbank.
   python tools/lanes_partner_probe.py ECAPA_TDNN_GLOB_c512 192 fp32 f16x3 [partner_model partner_embed_dim]"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from fixtures import synth
from bench import device_wavs
from wespeaker_amd.engine import Frontend, NativeSpeakerModel
name, E, prec, partner = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
pname, pE = (sys.argv[5], int(sys.argv[6])) if len(sys.argv) > 6 else (name, E)
use_embed = os.environ.get("EMBED") == "1"
sd = synth.synth_state_dict(name, 80, E, seed=11)
dev = torch.device("cuda:0")
A = NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=E, max_batch=64, max_frames=198)
Bm = NativeSpeakerModel(pname, synth.synth_state_dict(pname, 80, pE, seed=12), feat_dim=80, embed_dim=pE, max_batch=64, max_frames=198)
fe = Frontend(16000, 80); fe2 = Frontend(16000, 80)
w = device_wavs(64, 32000, dev, 40)
A.set_precision(prec)
feats = fe.fbank(w, cmn=True)
runA = (lambda: A.embed(feats)) if use_embed else (lambda: A.extract(fe, w))
ref = runA(); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(4096, 4096, device=dev)
worst = 0.0
for rep in range(6):
    outs = []
    for i in range(4):
        with torch.cuda.stream(s2):
            if partner == "matmul":
                y = x @ x
            elif partner == "copy":
                y = x.clone(); y2 = x + 1
            elif partner in ("fp32", "f16", "f16x3"):
                Bm.set_precision(partner); y = Bm.extract(fe2, w)
        with torch.cuda.stream(s1):
            outs.append(runA())
    torch.cuda.synchronize()
    worst = max(worst, max(float((o - ref).abs().max()) for o in outs))
print(name, prec, "embed" if use_embed else "extract", "partner", pname, partner, "worst", worst)
