#!/usr/bin/env python
"""Where a pass of the file-fed driver spends its time (tools/bench_driver.py's uniform list): header probe, batch
plan, the decode / upload / extract loop, the ark + scp write -- and the device-resident rate of the same batches."""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from wespeaker_amd import Frontend, SpeakerModelLanes
from wespeaker_amd import extract as wx
from fixtures import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
root = tempfile.mkdtemp(dir="/dev/shm")
try:
    base = synth.synth_wav(0, 40000)
    lines, paths = [], []
    for i in range(n):
        p = os.path.join(root, "u%05d.wav" % i)
        synth.write_wav(p, np.roll(base, i * 37)[:32000])
        lines.append("utt%05d %s" % (i, p)); paths.append(p)
    dev = torch.device("cuda:0")
    fe = Frontend(16000, 80, device=dev)
    lanes = SpeakerModelLanes("ECAPA_TDNN_GLOB_c512", synth.synth_state_dict("ECAPA_TDNN_GLOB_c512", 80, 192, seed=42),
                              lanes=2, feat_dim=80, embed_dim=192, device=dev, max_batch=256, max_frames=250)
    ex = wx.GpuExtractor(lanes, fe)
    wx.extract_list("scp", lines[:1024], ex, batch_size=1, max_batch=256)
    rec = {"files": n, "decode_threads": wx.decode_threads(0)}
    for rep in range(3):
        t0 = time.perf_counter()
        ns, sr = wx.probe_wavs(paths, wx.decode_threads(0))
        t1 = time.perf_counter()
        for k in ex.timing:
            ex.timing[k] = 0
        keys, emb = wx.extract_list("scp", lines, ex, batch_size=1, max_batch=256)
        t2 = time.perf_counter()
        tm = {k: (round(v * 1e3, 2) if isinstance(v, float) else v) for k, v in ex.timing.items()}
        wx.write_ark_scp(keys, emb, os.path.join(root, "o.ark"))
        t3 = time.perf_counter()
        # decode alone (no GPU work): the staging ring filled batch after batch
        td = time.perf_counter()
        for b0 in range(0, n, 256):
            st = ex.stage(min(256, n - b0), 32000)
            wx.load_wav_rows(paths[b0:b0 + 256], st.view, np.full(min(256, n - b0), 32000, np.int32), None, wx.decode_threads(0))
        td = time.perf_counter() - td
        rec["pass%d" % rep] = {"probe_ms": (t1 - t0) * 1e3, "extract_list_ms (incl. its own probe)": (t2 - t1) * 1e3,
                              "write_ms": (t3 - t2) * 1e3, "decode_only_ms": td * 1e3,
                              "utt_per_s_list+write": n / (t3 - t1), "submitting_thread_ms": tm}
    # device-resident: the same number of batches through the lanes, wav already on the GPU
    wav = torch.randint(-3000, 3000, (256, 32000), dtype=torch.int16, device=dev)
    for _ in range(4):
        lanes.extract(fe, wav)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // 256):
        lanes.extract(fe, wav)
    torch.cuda.synchronize()
    rec["device_resident_ms"] = (time.perf_counter() - t0) * 1e3
    print(json.dumps(rec))
finally:
    shutil.rmtree(root, ignore_errors=True)
