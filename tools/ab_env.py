#!/usr/bin/env python
"""Same-box A/B of environment switches on the headline-only bench (one process per variant, interleaved rounds).

    python tools/ab_env.py [--rounds 2] [--bench-args "--model ResNet34"] - WS_TAIL=0 WS_TAIL=2

`-` is the shipped path (no switch).  Prints one compact line per run: windows, one-batch-in-flight value, checksum.
"""
import argparse
import json
import os
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--bench-args", default="")
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    for r in range(args.rounds):
        for v in args.variants:
            env = dict(os.environ)
            if v != "-":
                for kv in v.split(","):
                    k, val = kv.split("=")
                    env[k] = val
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--headline-only", "--windows", "3", "--sustain-s", "0"]
            cmd += shlex.split(args.bench_args)
            out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                prec = d.get("headline_backend", "fp32")
                b = d["backends"][prec]
                print("%-24s round %d  windows %s  one-lane %.0f  checksum %.6f" % (
                    v, r, b["windows_embeddings_per_s"], d.get("value_one_batch_in_flight") or 0,
                    d.get("embedding_checksum") or 0), flush=True)
            except Exception as e:  # noqa: BLE001
                print("%-24s round %d  FAILED rc=%d %s" % (v, r, out.returncode, e), flush=True)


if __name__ == "__main__":
    main()
