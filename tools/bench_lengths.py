#!/usr/bin/env python
"""Throughput against utterance length and single-utterance latency (ECAPA-TDNN-GLOB-512; the numbers DESIGN §4.2 /
§8.7 quote) -> one JSON line per measurement, `profiles/r02_lengths.jsonl`.

    python tools/bench_lengths.py > profiles/r02_lengths.jsonl

Lengths: 1.2 / 2 / 4 / 8 s at a constant 512 s of audio per batch (the fused Res2 chain runs whole-utterance
windows up to 224 frames, time tiles beyond); latency: one and four 2 s utterances, forward only and wav -> embedding.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wespeaker_amd import Frontend, NativeSpeakerModel  # noqa: E402
from fixtures import synth  # noqa: E402
from bench import device_wavs  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    name, E = "ECAPA_TDNN_GLOB_c512", 192
    sd = synth.synth_state_dict(name, 80, E, seed=42)
    fe = Frontend(16000, 80, device=dev)
    for seconds in (1.2, 2.0, 4.0, 8.0):
        batch = int(512 / seconds) if seconds >= 2 else 256
        n = int(seconds * 16000)
        model = NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=E, device=dev, max_batch=batch,
                                   max_frames=fe.num_frames(n))
        wav = device_wavs(batch, n, dev, 0)
        for prec in ("fp32", "f16"):
            model.set_precision(prec)
            for _ in range(3):
                model.extract(fe, wav)
            rates = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    model.extract(fe, wav)
                torch.cuda.synchronize()
                rates.append(batch * 10 / (time.perf_counter() - t0))
            model.check_range()
            r = sorted(rates)[1]
            print(json.dumps({"what": "throughput", "seconds": seconds, "frames": fe.num_frames(n), "batch": batch,
                              "precision": prec, "utt_per_s": round(r, 1), "audio_s_per_s": round(r * seconds, 1)}),
                  flush=True)
        del model
    model = NativeSpeakerModel(name, sd, feat_dim=80, embed_dim=E, device=dev, max_batch=8, max_frames=198)
    for prec in ("fp32", "f16"):
        model.set_precision(prec)
        for B in (1, 4):
            x = torch.randn(B, 198, 80, device=dev)
            wav = device_wavs(B, 32000, dev, 0)
            for _ in range(5):
                model.embed(x)
            out = {}
            for key, fn in (("forward_us", lambda: model.embed(x)), ("wav_to_embedding_us", lambda: model.extract(fe, wav))):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    fn()
                    torch.cuda.synchronize()
                out[key] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
            print(json.dumps({"what": "latency", "batch": B, "frames": 198, "precision": prec, **out}), flush=True)


if __name__ == "__main__":
    main()
