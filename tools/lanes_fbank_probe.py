"""Open issue probe: does fbank give its serial bits while an engine of the given model / precision
runs on another stream?   python tools/lanes_fbank_probe.py ECAPA_TDNN_GLOB_c512 192 f16x3   (fp32 partner: clean)"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from fixtures import synth
from bench import device_wavs
from wespeaker_amd.engine import Frontend, NativeSpeakerModel
dev = torch.device("cuda:0")
pname, pE, pprec = sys.argv[1], int(sys.argv[2]), sys.argv[3]
Bm = NativeSpeakerModel(pname, synth.synth_state_dict(pname, 80, pE, seed=12), feat_dim=80, embed_dim=pE, max_batch=64, max_frames=198)
Bm.set_precision(pprec)
fe = Frontend(16000, 80); fe2 = Frontend(16000, 80)
w = device_wavs(64, 32000, dev, 40)
ref = fe.fbank(w, cmn=True).clone(); ref_nc = fe.fbank(w, cmn=False).clone(); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = bad_nc = 0; worst = 0.0
for rep in range(6):
    outs = []
    for i in range(6):
        with torch.cuda.stream(s2):
            y = Bm.extract(fe2, w)
        with torch.cuda.stream(s1):
            outs.append((fe.fbank(w, cmn=True), fe.fbank(w, cmn=False)))
    torch.cuda.synchronize()
    for a, b in outs:
        if not torch.equal(a, ref): bad += 1; worst = max(worst, float((a - ref).abs().max()))
        if not torch.equal(b, ref_nc): bad_nc += 1
print("partner", pname, pprec, ": fbank+cmn differs in", bad, "of 36, fbank alone in", bad_nc, "of 36; worst", worst)
torch.cuda.synchronize()
again = fe.fbank(w, cmn=False)
print("serial recompute equals ref:", torch.equal(again, ref_nc))
a = outs[-1][1]
d = (a - ref_nc).abs()
rows = torch.nonzero(d.amax(dim=(1, 2)) > 0).flatten().tolist()
print("utterances differing in last output:", rows[:20], "n =", len(rows))
if rows:
    u = rows[0]
    fr = torch.nonzero(d[u].amax(dim=1) > 0).flatten().tolist()
    print("utt", u, "frames differing:", fr[:40], "n =", len(fr))
    f0 = fr[0]
    bins = torch.nonzero(d[u, f0] > 0).flatten().tolist()
    print("frame", f0, "bins differing:", bins[:80], "n =", len(bins))
    print("got", a[u, f0, bins[:6]].tolist(), "ref", ref_nc[u, f0, bins[:6]].tolist())
