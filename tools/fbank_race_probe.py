"""DESIGN.md 6.0 root-cause probe: fbank next to a binary16 engine on another stream.

For both builds of the fbank kernel (ws_debug_fbank_mode: 1 = the round-3 build with packed-fp32 instructions,
0 = the shipped build without them) the victim frontend runs `LAUNCHES` times on stream 1 while a partner engine
runs forwards on stream 2; every output is compared bit for bit with the build's own serial result.

    python tools/fbank_race_probe.py [out.json] [launches]

Also prints what identifies the lease (GPU unique id, RAS / ECC counters before and after).
"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

from bench import device_wavs
from fixtures import synth
from wespeaker_amd import _lib
from wespeaker_amd.engine import Frontend, NativeSpeakerModel

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fbank_race_probe.json"
LAUNCHES = int(sys.argv[2]) if len(sys.argv) > 2 else 400


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=60).stdout.strip()
    except Exception as e:  # noqa: BLE001
        return "n/a (%s)" % e


def ras():
    return {"uniqueid": sh("rocm-smi --showuniqueid | grep -i unique"),
            "ras": sh("rocm-smi --showrasinfo all 2>/dev/null | grep -v '^=' | head -40"),
            "ecc": sh("amd-smi metric --ecc 2>/dev/null | head -30")}


dev = torch.device("cuda:0")
report = {"lease": ras(), "launches_per_case": LAUNCHES, "cases": []}
print("lease:", report["lease"]["uniqueid"], flush=True)

partners = {}


def partner(name, E, prec):
    key = (name, E)
    if key not in partners:
        partners[key] = NativeSpeakerModel(name, synth.synth_state_dict(name, 80, E, seed=12), feat_dim=80,
                                           embed_dim=E, max_batch=64, max_frames=198)
    partners[key].set_precision(prec)
    return partners[key]


fe, fe2 = Frontend(16000, 80), Frontend(16000, 80)
w = device_wavs(64, 32000, dev, 40)
L = _lib.lib()
L.ws_debug_fbank_mode(0)
ref = fe.fbank(w, cmn=False).clone()
torch.cuda.synchronize()
L.ws_debug_fbank_mode(1)
ref_packed = fe.fbank(w, cmn=False).clone()
torch.cuda.synchronize()
L.ws_debug_fbank_mode(0)
refs = {0: ref, 1: ref_packed}
print("packed build alone vs shipped build alone: max |d|", float((ref - ref_packed).abs().max()), flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
feats_p = fe2.fbank(w, cmn=True)


def run_case(mode, pname, pE, pprec, how):
    P = partner(pname, pE, pprec) if pprec else None
    torch.cuda.synchronize()
    L.ws_debug_fbank_mode(mode)
    bad = n = 0
    worst = 0.0
    bins = {}
    nframes, waves, rounds, blocks_hit, frames_per_launch = 0, {}, {}, set(), []
    t0 = time.time()
    ref = refs[mode]
    while n < LAUNCHES:
        outs = []
        for _ in range(5):
            if P is not None:
                with torch.cuda.stream(s2):
                    if how == "extract":
                        P.extract(fe2, w)
                    else:
                        P.embed(feats_p)
            with torch.cuda.stream(s1):
                for _ in range(8):
                    outs.append(fe.fbank(w, cmn=False))
        torch.cuda.synchronize()
        for o in outs:
            n += 1
            if not torch.equal(o, ref):
                bad += 1
                d = (o - ref).abs()
                worst = max(worst, float(d.max()))
                for b in torch.nonzero(d.amax(dim=(0, 1)) > 0).flatten().tolist():
                    bins[b] = bins.get(b, 0) + 1
                fr = torch.nonzero(d.amax(dim=2).flatten() > 0).flatten().tolist()     # frame = b * T + f
                frames_per_launch.append(len(fr))
                for f in fr:
                    nframes += 1
                    waves[f % 4] = waves.get(f % 4, 0) + 1
                    rounds[f // 6144] = rounds.get(f // 6144, 0) + 1
                    blocks_hit.add((f % 6144) // 4)
    L.ws_debug_fbank_mode(0)
    rec = {"fbank_mode": mode, "partner": pname if P is not None else None, "partner_precision": pprec,
           "partner_runs": how, "launches": n, "launches_differing": bad, "worst_abs": worst,
           "bins_differing": dict(sorted(bins.items())), "frames_differing": nframes,
           "frames_per_differing_launch_max": max(frames_per_launch) if frames_per_launch else 0,
           "by_wave_of_block": dict(sorted(waves.items())), "by_round_of_the_persistent_loop": dict(sorted(rounds.items())),
           "distinct_blocks_hit": len(blocks_hit),
           "seconds": round(time.time() - t0, 2)}
    report["cases"].append(rec)
    print(json.dumps(rec), flush=True)


ECAPA = ("ECAPA_TDNN_GLOB_c512", 192)
run_case(1, *ECAPA, None, "none")                            # alone: two streams' worth of launches, nothing beside
for prec in ("f16x3", "f16", "fp32"):
    for mode in (1, 0):
        run_case(mode, *ECAPA, prec, "embed")
for mode in (1, 0):
    run_case(mode, *ECAPA, "f16x3", "extract")
for pm in (("ResNet34", 256), ("ResNet221", 256), ("CAMPPlus", 512)):
    for mode in (1, 0):
        run_case(mode, *pm, "f16", "embed")
report["lease_after"] = ras()
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
with open(out_path, "w") as f:
    json.dump(report, f, indent=1)
tot = {}
for c in report["cases"]:
    k = "mode%d" % c["fbank_mode"]
    tot.setdefault(k, [0, 0])
    tot[k][0] += c["launches_differing"]
    tot[k][1] += c["launches"]
print("differing / launches per fbank build:", tot)
