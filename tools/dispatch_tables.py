#!/usr/bin/env python
"""Which kernel does every conv / linear layer of the BASELINE models get?  (ws_debug_dispatch_report)

    python tools/dispatch_tables.py            # print the tables of the five models x {fp32, f16}
    python tools/dispatch_tables.py --write    # (re)write tests/golden/dispatch_<model>_<prec>.txt

The tables are pinned by tests/test_gpu_parity.py::test_dispatch_tables_are_pinned: the dispatcher picks among a
dozen tile shapes / staging forms by shape and operand type, and a layer that silently falls off the fast forms
costs speed, never correctness -- the golden tables make such a change a reviewed one.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

# (model, embed_dim, batch, engine chunk): bench.py's per-GPU workloads at 2 s (T = 198)
CASES = [("ECAPA_TDNN_GLOB_c512", 192, 256, 256), ("ECAPA_TDNN_GLOB_c1024", 192, 256, 256),
         ("ResNet34", 256, 512, 512), ("ResNet221", 256, 256, 256), ("CAMPPlus", 512, 512, 512)]
PRECISIONS = ("fp32", "f16")


def table(model, E, batch, chunk, prec, frames=198):
    from wespeaker_amd import NativeSpeakerModel
    from wespeaker_amd.engine import dispatch_log, dispatch_report
    from fixtures import synth
    dev = torch.device("cuda:0")
    sd = synth.synth_state_dict(model, 80, E, seed=42)
    m = NativeSpeakerModel(model, sd, feat_dim=80, embed_dim=E, device=dev, max_batch=chunk, max_frames=frames)
    m.set_precision(prec)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    feats = torch.randn(batch, frames, 80, device=dev, generator=g)
    m.embed(feats)                     # first-use initialisation outside the log
    torch.cuda.synchronize()
    dispatch_log(True, clear=True)
    try:
        m.embed(feats)
        torch.cuda.synchronize()
        return dispatch_report()
    finally:
        dispatch_log(False, clear=True)


def golden_path(model, prec):
    return os.path.join(ROOT, "tests", "golden", "dispatch_%s_%s.txt" % (model, prec))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    for model, E, batch, chunk in CASES:
        for prec in PRECISIONS:
            lines = table(model, E, batch, chunk, prec)
            if args.write:
                with open(golden_path(model, prec), "w") as f:
                    f.write("".join(l + "\n" for l in lines))
            print("== %s %s batch %d: %d distinct (problem, kernel) pairs" % (model, prec, batch, len(lines)))
            for l in lines:
                print("   " + l)


if __name__ == "__main__":
    main()
