#!/usr/bin/env python
"""ws_wav_probe / ws_wav_load_rows over 4096 two-second files in /dev/shm against the number of C++ threads."""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wespeaker_amd import extract as wx
from fixtures import synth
n = 4096
root = tempfile.mkdtemp(dir="/dev/shm")
try:
    base = synth.synth_wav(0, 40000)
    paths = []
    for i in range(n):
        p = os.path.join(root, "u%05d.wav" % i); synth.write_wav(p, np.roll(base, i * 37)[:32000]); paths.append(p)
    pin = torch.empty((256, 32000), dtype=torch.int16).pin_memory()
    out = {}
    for th in (1, 2, 4, 8, 16, 32, 64, 128):
        wx.probe_wavs(paths, th)
        t0 = time.perf_counter(); wx.probe_wavs(paths, th); t1 = time.perf_counter()
        for b0 in range(0, n, 256):
            wx.load_wav_rows(paths[b0:b0 + 256], pin, np.full(256, 32000, np.int32), None, th)
        t2 = time.perf_counter()
        out[th] = {"probe_ms": round((t1 - t0) * 1e3, 2), "decode_ms": round((t2 - t1) * 1e3, 2)}
    print(json.dumps(out))
finally:
    shutil.rmtree(root, ignore_errors=True)
