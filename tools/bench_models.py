#!/usr/bin/env python
"""Throughput of every model family of SURVEY.md section 8 (BASELINE.json configs 2-4) on one MI355X:
wav resident in HBM -> fbank -> CMN -> forward, both GEMM back-ends.  One JSON line per model.

    python tools/bench_models.py [--steps 5] > profiles/r01_models.jsonl
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wespeaker_amd import Frontend, NativeSpeakerModel
from fixtures import synth
from bench import device_wavs

CASES = [  # (model, embed_dim, batch, chunk)
    ("ECAPA_TDNN_GLOB_c512", 192, 256, 256),
    ("ECAPA_TDNN_c512", 192, 256, 256),
    ("ECAPA_TDNN_GLOB_c1024", 192, 256, 256),      # BASELINE configs[1]: batch 256 x 2 s
    ("ResNet34", 256, 512, 512),
    ("ResNet221", 256, 128, 64),
    ("CAMPPlus", 512, 512, 512),                   # its GEMMs have M = B*T/2 rows: batch 512 fills the chip
]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--batch", type=int, default=0, help="override batch and engine chunk")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    fe = Frontend(16000, 80, device=dev)
    T = fe.num_frames(32000)
    for name, ed, batch, chunk in CASES:
        if args.only and args.only not in name:
            continue
        if args.batch:
            batch = chunk = args.batch
        sd = synth.synth_state_dict(name, 80, ed, seed=42)
        model = NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=ed, device=dev, max_batch=chunk,
                                   max_frames=T)
        wav = device_wavs(batch, 32000, dev, 0)
        rec = {"model": name, "batch": batch, "engine_chunk": chunk, "frames": T,
               "gflop_per_utt": model.flops(1, T) / 1e9}
        for prec in ("fp32", "f16x3", "f16"):
            model.set_precision(prec)
            for _ in range(2):
                model.extract(fe, wav)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                emb = model.extract(fe, wav)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            assert bool(torch.isfinite(emb).all())
            rec[prec] = {"embeddings_per_s": batch / dt, "ms_per_step": dt * 1e3,
                         "model_tflops": rec["gflop_per_utt"] * batch / dt / 1e3}
        print(json.dumps(rec), flush=True)
        del model

if __name__ == "__main__":
    main()
