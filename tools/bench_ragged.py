#!/usr/bin/env python
"""Throughput of a ragged batch (ws_extract_ragged, lengths within 12 % of each other: what the driver forms
from a real whole-utterance list) against the uniform batch of the same size (ws_extract), per back-end.

    python tools/bench_ragged.py [--model ECAPA_TDNN_GLOB_c512] [--batch 256]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from wespeaker_amd import Frontend, NativeSpeakerModel
from fixtures import synth
from bench import device_wavs, EMBED_DIM


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ECAPA_TDNN_GLOB_c512")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    E = EMBED_DIM.get(args.model[:5], 256)
    fe = Frontend(16000, 80, device=dev)
    sd = synth.synth_state_dict(args.model, 80, E, seed=42)
    model = NativeSpeakerModel(args.model, sd, feat_dim=80, embed_dim=E, device=dev, max_batch=args.batch,
                               max_frames=fe.num_frames(32000))
    wav = device_wavs(args.batch, 32000, dev, 0)
    rng = np.random.Generator(np.random.PCG64(5))
    ns = rng.integers(int(32000 / 1.12), 32001, args.batch).astype(np.int32)
    ns[0] = 32000
    rec = {"model": args.model, "batch": args.batch, "mean_len_over_max": float(ns.mean() / 32000)}
    for prec in ("fp32", "f16"):
        model.set_precision(prec)
        out = {}
        for name, fn in (("uniform", lambda: model.extract(fe, wav)),
                         ("ragged", lambda: model.extract_ragged(fe, wav, ns))):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            out[name] = args.batch * args.steps / (time.perf_counter() - t0)
        out["ragged_over_uniform"] = out["ragged"] / out["uniform"]
        rec[prec] = out
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
