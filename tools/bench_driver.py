#!/usr/bin/env python
"""End-to-end rate of the batch driver (wespeaker_amd.extract: wav files -> decode threads -> pinned H2D on a
copy stream -> ws_extract / ws_extract_ragged -> D2H -> ark/scp) on synthetic PCM16 files in /dev/shm.

    python tools/bench_driver.py [--n 4096] [--model ECAPA_TDNN_GLOB_c512] [--workers 8]

Two lists: all files exactly 2 s (the uniform path) and lengths uniform in [1.5, 2.5] s (a real test set: every
length different -> 12 % length classes -> ragged batches).  Reports utterances/s per back-end and the share of
the device-resident rate (bench.py) that survives the host pipeline.
"""
import argparse, json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from wespeaker_amd import Frontend, NativeSpeakerModel
from wespeaker_amd import extract as wx
from fixtures import synth
from bench import EMBED_DIM


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--model", default="ECAPA_TDNN_GLOB_c512")
    ap.add_argument("--workers", type=int, default=0, help="0 = scaled with the host (extract.decode_threads)")
    ap.add_argument("--max_batch", type=int, default=256)
    ap.add_argument("--repeats", type=int, default=7)
    ap.add_argument("--python_loader", action="store_true", help="the Python thread-pool decode path instead of "
                    "the native loader (ws_wav_load_rows)")
    ap.add_argument("--long_n", type=int, default=0, help="instead of the two 2-s lists: this many files of 4 - 12 s "
                    "(whole utterances of a VoxCeleb-like test set) -> utt/s and audio-seconds/s")
    args = ap.parse_args()
    if args.long_n:
        return long_list(args)
    root = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        rng = np.random.Generator(np.random.PCG64(1))
        base = synth.synth_wav(0, 40000)
        lists = {"uniform_2s": [], "varied_1.5-2.5s": []}
        for i in range(args.n):
            for tag, n in (("uniform_2s", 32000), ("varied_1.5-2.5s", int(rng.integers(24000, 40001)))):
                p = os.path.join(root, "%s_%05d.wav" % (tag[:3], i))
                synth.write_wav(p, np.roll(base, i * 37)[:n])
                lists[tag].append("utt%05d %s" % (i, p))
        dev = torch.device("cuda:0")
        E = EMBED_DIM.get(args.model[:5], 256)
        fe = Frontend(16000, 80, device=dev)
        model = NativeSpeakerModel(args.model, synth.synth_state_dict(args.model, 80, E, seed=42), feat_dim=80,
                                   embed_dim=E, device=dev, max_batch=args.max_batch, max_frames=250)
        rec = {"model": args.model, "files": args.n, "decode_threads": wx.decode_threads(args.workers), "max_batch": args.max_batch,
               "loader": "python threads" if args.python_loader else "native (ws_wav_load_rows)",
               "host_cores": os.cpu_count()}
        from wespeaker_amd import SpeakerModelLanes
        lanes2 = SpeakerModelLanes(args.model, synth.synth_state_dict(args.model, 80, E, seed=42), lanes=2, feat_dim=80,
                                   embed_dim=E, device=dev, max_batch=args.max_batch, max_frames=250)
        lanes3 = SpeakerModelLanes(args.model, synth.synth_state_dict(args.model, 80, E, seed=42), lanes=3, feat_dim=80,
                                   embed_dim=E, device=dev, max_batch=args.max_batch, max_frames=250)
        for prec in ("fp32", "fp32_2lanes", "fp32_3lanes", "f16"):
            if not prec.endswith("lanes"):
                model.set_precision(prec)
            for tag, lines in lists.items():
                ex = wx.GpuExtractor({"fp32_2lanes": lanes2, "fp32_3lanes": lanes3}.get(prec, model), fe)
                run = (lambda ls: wx.extract_entries(wx.iter_entries("scp", ls), ex, batch_size=1,
                                                     max_batch=args.max_batch, num_workers=args.workers)) \
                    if args.python_loader else \
                    (lambda ls: wx.extract_list("scp", ls, ex, batch_size=1, max_batch=args.max_batch,
                                                num_workers=args.workers))
                run(lines[:512])                                                   # warm-up (page cache, capacity)
                # one pass over 8 k two-second files lasts 0.07 - 0.25 s: a single pass is mostly scheduling noise
                # of the decode threads (the same box gave 55 k and 118 k utt/s for one list), so the MEDIAN of
                # --repeats passes is reported and all of them are kept beside it
                rates = []
                for _ in range(args.repeats):
                    t0 = time.perf_counter()
                    keys, emb = run(lines)
                    wx.write_ark_scp(keys, emb, os.path.join(root, "out_%s_%s.ark" % (prec, tag[:3])))
                    rates.append(args.n / (time.perf_counter() - t0))
                    assert len(keys) == args.n and np.isfinite(emb).all()
                rec["%s/%s" % (prec, tag)] = float(np.median(rates))
                rec["%s/%s/passes" % (prec, tag)] = [round(r) for r in rates]
        print(json.dumps(rec))
    finally:
        shutil.rmtree(root, ignore_errors=True)


def long_list(args):
    """Whole utterances of 4 - 12 s (every length different): sorted-length ragged batches, time-tiled Res2 chain,
    the attentive pooling kernel on segments."""
    root = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        rng = np.random.Generator(np.random.PCG64(2))
        base = synth.synth_wav(0, 192000)
        lines, total_s = [], 0.0
        for i in range(args.long_n):
            n = int(rng.integers(64000, 192001))
            p = os.path.join(root, "long_%05d.wav" % i)
            synth.write_wav(p, np.roll(base, i * 37)[:n])
            lines.append("utt%05d %s" % (i, p))
            total_s += n / 16000.0
        dev = torch.device("cuda:0")
        E = EMBED_DIM.get(args.model[:5], 256)
        fe = Frontend(16000, 80, device=dev)
        from wespeaker_amd import SpeakerModelLanes
        sd = synth.synth_state_dict(args.model, 80, E, seed=42)
        mb = min(args.max_batch, 128)
        model = NativeSpeakerModel(args.model, sd, feat_dim=80, embed_dim=E, device=dev, max_batch=mb, max_frames=1200)
        lanes2 = SpeakerModelLanes(args.model, sd, lanes=2, feat_dim=80, embed_dim=E, device=dev, max_batch=mb,
                                   max_frames=1200)
        rec = {"model": args.model, "files": args.long_n, "seconds_of_audio": round(total_s, 1), "lengths": "4 - 12 s",
               "max_batch": mb, "decode_threads": wx.decode_threads(args.workers), "host_cores": os.cpu_count()}
        for prec in ("fp32", "fp32_2lanes", "f16"):
            if not prec.endswith("lanes"):
                model.set_precision(prec)
            ex = wx.GpuExtractor(lanes2 if prec.endswith("lanes") else model, fe)
            run = lambda ls: wx.extract_list("scp", ls, ex, batch_size=1, max_batch=mb, num_workers=args.workers)
            run(lines[:256])
            rates = []
            for _ in range(args.repeats):
                t0 = time.perf_counter()
                keys, emb = run(lines)
                wx.write_ark_scp(keys, emb, os.path.join(root, "out_%s.ark" % prec))
                rates.append(1.0 / (time.perf_counter() - t0))
                assert len(keys) == args.long_n and np.isfinite(emb).all()
            r = float(np.median(rates))
            rec["%s/utt_per_s" % prec] = round(r * args.long_n, 1)
            rec["%s/audio_s_per_s" % prec] = round(r * total_s, 1)
        print(json.dumps(rec))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
