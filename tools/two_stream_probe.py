"""Experiment: does running two half-batches on two HIP streams (two engines) hide kernel tails?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wespeaker_amd import Frontend, NativeSpeakerModel
from fixtures import synth
from bench import device_wavs
dev = torch.device("cuda:0")
sd = synth.synth_ecapa_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42)
fe = Frontend(16000, 80, device=dev)
wav = device_wavs(256, 32000, dev, 0)
import itertools
MODE = sys.argv[1] if len(sys.argv) > 1 else "split"      # split: half batches per lane; full: whole batches alternate
for nstream in (1, 2, 3):
    per = 256 // nstream if MODE == "split" else 256
    models = [NativeSpeakerModel("ECAPA_TDNN_GLOB_c512", sd, device=dev, max_batch=per, max_frames=198) for _ in range(nstream)]
    streams = [torch.cuda.Stream(dev) for _ in range(nstream)]
    def step():
        outs = []
        for i, (m, s) in enumerate(zip(models, streams)):
            with torch.cuda.stream(s):
                outs.append(m.extract(fe, wav[i * per:(i + 1) * per] if MODE == "split" else wav))
        return outs
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    n_utts = 256 if MODE == "split" else 256 * nstream
    print("%s lanes %d: %.3f ms/step  %.0f emb/s" % (MODE, nstream, dt * 1e3, n_utts / dt))
    del models
