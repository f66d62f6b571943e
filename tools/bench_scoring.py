#!/usr/bin/env python
"""Side benchmark (not the headline): cosine scoring + AS-norm on one MI355X vs the numpy oracle.

  python tools/bench_scoring.py            # prints one JSON object

Workload: VoxCeleb1-O-sized eval set (4874 x 256 float32 embeddings, 37 611 trials -- the sizes
SURVEY.md 8(d) quotes) against a 20 000-utterance cohort, top_n = 300 (the recipes' default), plus
a 1 M-pair cosine trial list.  Inputs are resident on the GPU when the timed regions start.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from wespeaker_amd import score as wscore  # noqa: E402
from fixtures import synth


def timed(fn, reps=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    n, n_cohort, dim, top_n = 4874, 20000, 256, 300
    emb, _ = synth.synth_embeddings(n, dim, seed=31)
    cohort, _ = synth.synth_embeddings(n_cohort, dim, seed=32)
    t, c = wscore.UnitTable(emb), wscore.UnitTable(cohort)
    ia, ib = synth.synth_trial_pairs(1000000, n, n, seed=5)
    ia_d, ib_d = torch.from_numpy(ia).cuda(), torch.from_numpy(ib).cuda()
    out = {"workload": {"eval": [n, dim], "cohort": [n_cohort, dim], "top_n": top_n,
                        "pairs": int(ia.shape[0])}}
    dt = timed(lambda: wscore.cosine_pairs(t, t, ia_d, ib_d))
    out["cosine_pairs_per_s"] = ia.shape[0] / dt
    out["cosine_pairs_ms"] = dt * 1e3
    dt = timed(lambda: wscore.cohort_stats(t, c, top_n))
    out["cohort_stats_ms"] = dt * 1e3
    out["cohort_scores_per_s"] = n * n_cohort / dt
    # algorithmic bytes: the score matrix is written once and read once (4 B each way)
    out["cohort_stats_algorithmic_GBps"] = 8.0 * n * n_cohort / dt / 1e9
    dt = timed(lambda: wscore.cosine_matrix(t, c))
    out["cosine_matrix_ms"] = dt * 1e3
    out["cosine_matrix_TFLOPs"] = 2.0 * n * n_cohort * dim / dt / 1e12
    # CPU oracle on a bounded sample (first 256 eval rows), all host threads numpy gives us
    from oracle import score as oscore
    t0 = time.perf_counter()
    oscore.get_mean_std(emb[:256], cohort, top_n)
    cpu = time.perf_counter() - t0
    out["cpu_oracle_cohort_scores_per_s"] = 256 * n_cohort / cpu
    out["cpu_sample"] = "oracle get_mean_std (numpy float32: matmul + full row sort) on 256 of the 4874 rows"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
