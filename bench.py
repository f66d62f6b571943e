#!/usr/bin/env python
"""Headline benchmark: embeddings/sec on 2 s @ 16 kHz synthetic utterances through a speaker model
(wav -> Kaldi fbank -> CMN -> forward, all in the HIP library) + PLDA trials/sec.

    python bench.py [--gpus N --steps K --warmup W] [--model M] [--precision P]
    python bench.py --gpus 8 --workload vox1o            # fixed-size set, strong scaling
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no torchrun environment launches itself: the process re-executes under
`torch.distributed.run` with one rank per GPU (the reference's launcher spawns its per-GPU jobs the same way,
tools/extract_embedding.sh:46-65).

Two workload shapes:

* default (weak scaling): one "step" = one pass of the hot path over one batch of `--batch` utterances per GPU whose
  PCM16 samples are already resident in HBM; the only collective is the all_gather of the (B, E) embeddings, inside
  the timed region.  `value` = utterances of all ranks / max-over-ranks time.
* `--total-utts U` / `--workload vox1o|stream10k` (strong scaling, BASELINE.json configs 2 and 3): a FIXED set of U
  utterances is cut into contiguous shards by parallel.shard_range (tools/extract_embedding.sh:39-67's rule), every
  rank walks its shard in `--batch`-utterance batches, ONE all_gather collects the (U, E) table, and rank 0 scores
  the trial list (PLDA LLR + cosine) -- one step = the whole set, all of it inside the timed region.

Which number is the headline.  The reference path north_star names is PyTorch **fp32**; the parity-grade back-end is
WS_PREC_FP32 (exact fp32 products on v_mfma_f32_32x32x2_f32) and `value` / `dtype` / `roofline` describe THAT run
(`--precision fp32`, the default).  f16x3 and f16 are declared fast modes, timed on the same workload in the same
process and reported under `backends`, each with its own `roofline` block (dominant kernel class measured live with
HIP events on the launch stream).

What goes where.  Rank 0 prints ONE compact JSON line (< 8 KB; `compact_line`: the contract's keys, the headline
`roofline` and `cpu_baseline`, `plda`, one figure per other BASELINE config, the fixed-size sets) and writes the full
record -- per-back-end blocks with their windows, notes, shard tables, per-class event times -- to `bench_detail.json`
beside this script (`--detail-file`).  Round 5's 24-KB line was dropped by the driver's parser.  Three batches are in
flight per GPU by default (`--lanes`); the sustained leg, the 1e8-trial PLDA matrices and the batch sweep are opt-in.

WS_BENCH_STUB=1 (tests only): a host stand-in for the extractor (no GPU, gloo) so that the launch / sharding /
gather / JSON logic of THIS file can run on CPU at world size 2.
"""
import argparse
import json
import math
import os
import socket
import statistics
import sys
import time

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL / tensor sharing otherwise fail with
# hipIpcGetMemHandle: invalid argument); the launch environment exports it, keep it if it does not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from wespeaker_amd import parallel  # noqa: E402
from fixtures import synth  # noqa: E402

STUB = os.environ.get("WS_BENCH_STUB") == "1"

FP32_MFMA_PEAK_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md, dense f32 MFMA
F16_MFMA_PEAK_TFLOPS = 2500.0       # dense f16/bf16 MFMA (not the 2:1-sparse marketing figure)
HBM_PEAK_GBS = 8000.0               # HBM3E spec (6.3 TB/s is what a copy reaches)
# FP64 matrix: the local guide has no row for it.  The datasheet says 78.6 TFLOP/s; a register-only burner of
# v_mfma_f64_16x16x4_f64 (tools/f64_peak_probe.hip: 8 independent accumulator chains per wavefront, no memory traffic)
# sustains 47.0 / 47.7 / 49.4 TFLOP/s at 2 / 4 / 8 wavefronts per SIMD and 2.37 GHz on this chip
# (profiles/r06_f64_mfma_peak.jsonl: ~100 cycles per instruction and SIMD, not 64) -- the measured figure is the peak
# the PLDA GEMM is priced against, the datasheet figure is quoted beside it
F64_MFMA_PEAK_TFLOPS = 49.4
F64_MFMA_DATASHEET_TFLOPS = 78.6
BACKENDS = ("fp32", "f16x3", "f16")
DOMINANT = "gemm_main"              # profile class 0: every conv/linear GEMM launch with N > 64
METRIC = "embeddings/sec (2 s utts, ECAPA-512) + PLDA trials/sec at 1/2/4/8 MI355X"

# per-GPU batch / engine chunk defaults: ECAPA = BASELINE configs[1]'s 256 x 2 s; the 2-D families get
# what fills the chip at their row counts (ResNet221: 256-utterance chunks)
DEFAULT_BATCH = {"ECAPA": (256, 256), "ResNet34": (512, 512), "ResNet18": (512, 512),
                 "ResNe": (256, 256), "CAMPP": (512, 512)}
EMBED_DIM = {"ECAPA": 192, "ResNe": 256, "CAMPP": 512}
# BASELINE.json configs 2 / 3: (model, utterances, trials)
WORKLOADS = {"vox1o": ("ResNet34", 4874, 37611), "stream10k": ("CAMPPlus", 10000, 0)}

DTYPE_TEXT = {
    "fp32": "f32",
    "f16x3": "f16x3 split MFMA, f32 accumulate (fp32-grade: 1.7e-6 rel err vs float64; declared fast mode)",
    "f16": "f16 MFMA operands, f32 accumulate (declared fast mode: narrower than the fp32 reference; "
           "1 - cos = 2e-7 vs it in tests/, bar 1e-4)",
}


def kernel_text(model_name, prec):
    ecapa = model_name.startswith("ECAPA")
    if prec == "fp32":
        return ("gemm_f32_stream_kernel (persistent: one workgroup of eight 64x64 wavefronts per CU walks whole rounds "
                "of 256x128 tiles, operands by LDS-DMA into a 3-stage ring of 48-KB K-tiles, v_mfma_f32_32x32x2_f32, "
                "exact fp32 products: the plain 1x1 layers, ECAPA's k5 layer as an im2col GEMM, and -- CONV form, per-tap "
                "A pieces; 256x64 tile for 64 channels -- the 3x3 / stride-1 layers of the ResNets' stages 2-4) + "
                + ("astp_fused_kernel (attention linear1 -> tanh -> linear2 -> softmax pooling, one workgroup per "
                   "utterance, v_mfma_f32_16x16x4_f32) + " if ecapa else "")
                + "conv_gemm_dual_kernel / conv_gemm_kernel<..,PREC=0> "
                "(everything else and the remaining rows): every conv/linear with N > 64")
    if prec == "f16x3":
        return ("conv_gemm_dual_kernel<..,PREC=1> / conv_gemm_kernel<128,128,2,2,..,PREC=1> (3 x "
                "v_mfma_f32_32x32x16_f16 on hi/lo binary16 splits): every conv/linear with N > 64")
    if ecapa:
        return ("gemm_f16_p8_kernel (256x256 tile, two wave groups one barrier interval apart, "
                "v_mfma_f32_32x32x16_f16, both binary16 operands staged by global_load_lds_dwordx4: the "
                "N >= 512 layers) + gemm_f16_dma_kernel<128,128,64,2> (attention layers, one with the "
                "fused pooling epilogue) + their 64x64 tail launches")
    return ("gemm_f16_dma_kernel<..,CONV> / gemm_f16_p8_kernel<CONV> (implicit-GEMM convolutions on binary16 "
            "maps by LDS-DMA) + conv3x3_direct_f16_kernel launches routed through the same class")


# ------------------------------------------------------------------------------------------ launch
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch_if_needed(argv, gpus):
    """`python bench.py --gpus N` (N > 1) without a torchrun environment: become
    `python -m torch.distributed.run --nproc-per-node N ... bench.py <same args>` (one rank per GPU)."""
    if gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


# ------------------------------------------------------------------------------------------ inputs
def device_wavs(batch, num_samples, device, seed_base):
    """Synthetic PCM16 batch generated on the device (same recipe family as synth.synth_wav:
    gaussian noise sigma 3000 + 8000-amplitude tone with per-utterance f0 in [80, 400] Hz)."""
    g = torch.Generator(device=device)
    g.manual_seed(1234 + seed_base)
    t = torch.arange(num_samples, device=device, dtype=torch.float32) / 16000.0
    f0 = 80.0 + 320.0 * torch.rand(batch, 1, device=device, generator=g)
    x = 3000.0 * torch.randn(batch, num_samples, device=device, generator=g)
    x = x + 8000.0 * torch.sin(2 * np.pi * f0 * t[None, :])
    return x.round().clamp(-32768, 32767).to(torch.int16).contiguous()


class StubExtractor:
    """Host stand-in (WS_BENCH_STUB=1, tests only): a fixed random projection of the samples."""

    def __init__(self, embed_dim, num_samples):
        g = torch.Generator().manual_seed(5)
        self.w = torch.randn(num_samples, embed_dim, generator=g) / num_samples ** 0.5

    def extract(self, fe, wav):
        # (tests: WS_BENCH_STUB_SLOW="rank:seconds" makes ONE rank slow, to prove that `value` is priced on the
        # max-over-ranks time and not on rank 0's own clock)
        slow = os.environ.get("WS_BENCH_STUB_SLOW")
        if slow:
            r, sec = slow.split(":")
            if int(r) == int(os.environ.get("RANK", "0")):
                time.sleep(float(sec))
        return wav.to(torch.float32) @ self.w

    def set_precision(self, prec):
        return self

    def check_range(self):
        pass


def oracle_forward_fn(model_name, embed_dim):
    """callable feats (1, T, 80) -> emb: the CPU restatement of the reference forward."""
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in
          synth.synth_state_dict(model_name, 80, embed_dim, seed=42).items()}
    if model_name.startswith("ECAPA"):
        from oracle import ecapa as o
        return lambda f: o.ecapa_forward(sd, f)
    if model_name.startswith("ResNet"):
        from oracle import resnet as o
        return lambda f: o.resnet_forward(sd, f, model_name)
    from oracle import campplus as o
    return lambda f: o.campplus_forward(sd, f)


def cpu_baseline(model_name, embed_dim, sample_utts, budget_s=8.0):
    """Oracle on the host cores: the Speaker.extract_embedding_list loop (fbank -> CMN -> model,
    batch 1) over up to `sample_utts` synthetic utterances, `budget_s` seconds per thread count."""
    from oracle import fbank as ofbank
    forward = oracle_forward_fn(model_name, embed_dim)
    wavs = [synth.synth_wav(i) for i in range(sample_utts)]
    avail = os.cpu_count() or 1
    best = None
    # batch-1 torch ops do not scale to every core of a large host: try a few thread counts on a
    # bounded sample each and report the best one (the threads actually used are stated)
    for threads in sorted({1, min(8, avail), min(32, avail)}):
        torch.set_num_threads(threads)
        forward(ofbank.speaker_features(wavs[0])[None])      # warm-up
        n = 0
        t0 = time.perf_counter()
        for w in wavs:
            forward(ofbank.speaker_features(w)[None])
            n += 1
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
        if best is None or n / dt > best[0]:
            best = (n / dt, threads, n, dt)
    return {"value": best[0], "unit": "embeddings/s", "cores": best[1], "kind": "port",
            "host_cores": avail, "host_cpu_model": host_cpu_model(),
            "what": "oracle/ (numpy fbank + torch-fp32 functional restatement of the reference forward, "
                    "bit-identical to the reference nn.Module on the golden cases); the reference checkout "
                    "itself is not on the GPU box, so this is a port, not `reference`",
            #  the reference's OWN nn.Module, timed once where the checkout exists -- the build
            # container's 8 host cores, forward only, no fbank: BASELINE.md section 3
            "reference_module_on_the_build_container": {
                "value": 68.3, "unit": "embeddings/s", "cores": 8, "one_thread": 28.4,
                "what": "wespeaker.models.ecapa_tdnn ECAPA_TDNN_GLOB_c512 forward on (1, 198, 80) inputs, batch 1 x "
                        "100, torch CPU fp32 (BASELINE.md:73); a constant quoted beside the port, not measured in this run"},
            "sample": "%d synthetic 2 s utts in %.1f s, batch 1 (the Speaker.extract_embedding_list "
                      "loop), %s; best of 1/8/32 threads" % (best[2], best[3], model_name)}


def plda_cpu_baseline(params, enroll_t, test_t, idx_e, idx_t, budget_s=4.0):
    """The reference's own scoring on the host cores (oracle/plda.py): the per-trial Python loop of eval_sv
    (two_cov_plda.py:165-184, 246-256) on a bounded sample of the trial list, and the vectorised numpy closed
    form (a dense 1000 x 1000 block) as the fair CPU figure."""
    from oracle import plda as oplda
    e = np.asarray(enroll_t, dtype=np.float64)
    t = np.asarray(test_t, dtype=np.float64)
    n_loop, t0 = 0, time.perf_counter()
    chunk = 2000
    while time.perf_counter() - t0 < budget_s and n_loop < len(idx_e):
        sl = slice(n_loop, min(len(idx_e), n_loop + chunk))
        oplda.llr_pairs(params, e, np.ones(e.shape[0]), t, idx_e[sl], idx_t[sl])
        n_loop = sl.stop
    loop_dt = time.perf_counter() - t0
    ne, nt = min(1000, e.shape[0]), min(1000, t.shape[0])
    oplda.llr_matrix_vectorised(params, e[:ne], np.ones(ne), t[:nt])
    reps, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < budget_s / 2 or reps == 0:
        oplda.llr_matrix_vectorised(params, e[:ne], np.ones(ne), t[:nt])
        reps += 1
    mat_dt = (time.perf_counter() - t1) / reps
    return {"value": n_loop / loop_dt, "unit": "trials/s", "cores": 1, "kind": "port",
            "sample": "%d trials of the list in %.1f s through oracle/plda.py's per-trial loop (the reference's "
                      "eval_sv trial loop, two_cov_plda.py:246-256; pinned to the reference on "
                      "tests/golden/plda_ref.npz)" % (n_loop, loop_dt),
            "vectorised_numpy_trials_per_s": ne * nt / mat_dt,
            "vectorised_sample": "dense %d x %d block, numpy float64 closed form (BLAS threads as numpy chooses), "
                                 "%.1f ms per block" % (ne, nt, mat_dt * 1e3)}


def plda_leg(args, device, with_cpu_baseline):
    """1 M synthetic trial pairs over 2 x 10 k embeddings (incl. the transform) + a dense 1000 x 1000 matrix, float64;
    roofline of the pair kernel (cache-gather bound) and of the dense GEMM, CPU baseline = the reference's own
    per-trial loop (oracle/plda.py) on a bounded sample + vectorised numpy."""
    from wespeaker_amd import TwoCovPLDA
    D = 192
    p = synth.synth_plda(D, seed=7)
    plda = TwoCovPLDA.from_params(p["mu"], p["transform"], p["psi"], p["offset"], False, device=device)
    n_emb = 10000
    emb_tab, _ = synth.synth_embeddings(2 * n_emb, D, seed=11)
    emb_tab = torch.from_numpy(emb_tab).to(device)
    ie, it = synth.synth_trial_pairs(args.trials, n_emb, n_emb, seed=99)
    ie_d, it_d = torch.from_numpy(ie).to(device), torch.from_numpy(it).to(device)
    nn = 1          # multisession_avg=True: every enrollment model counts as one session

    def timed(fn, k):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(device)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t1 = time.perf_counter()
        ev0.record()
        for _ in range(k):
            fn()
        ev1.record()
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t1) / k, ev0.elapsed_time(ev1) * 1e-3 / k

    k = max(3, min(args.steps, 20))
    e_t = plda.prepare_test(emb_tab[:n_emb])
    t_t = plda.prepare_test(emb_tab[n_emb:])

    def plda_step():
        a = plda.prepare_test(emb_tab[:n_emb])
        b = plda.prepare_test(emb_tab[n_emb:])
        return plda.llr_pairs(a, nn, b, ie_d, it_d)

    pdt, _ = timed(plda_step, k)
    _, pair_dev = timed(lambda: plda.llr_pairs(e_t, nn, t_t, ie_d, it_d), k)
    e1k = plda.prepare_test(emb_tab[:1000])
    t1k = plda.prepare_test(emb_tab[n_emb:n_emb + 1000])
    mdt, mat_dev = timed(lambda: plda.llr_matrix(e1k, nn, t1k), k)
    # dense matrices at the size where the MFMA loop matters: 10 000 x 10 000 = 1e8 trials, the
    # per-rank block of parallel.llr_matrix_sharded, at D = 192 (ECAPA) and D = 512 (CAM++); 800 MB of float64 scores
    dense = {}
    for dd in ((D, 512) if args.plda_dense else ()):
        pd = synth.synth_plda(dd, seed=7)
        pl = plda if dd == D else TwoCovPLDA.from_params(pd["mu"], pd["transform"], pd["psi"], pd["offset"], False,
                                                          device=device)
        tab, _ = synth.synth_embeddings(2 * n_emb, dd, seed=11)
        tab = torch.from_numpy(tab).to(device)
        ea, tb = pl.prepare_test(tab[:n_emb]), pl.prepare_test(tab[n_emb:])
        out_holder = [None]

        def dense_step():
            out_holder[0] = None                      # (release the previous 800 MB before the next launch allocates)
            out_holder[0] = pl.llr_matrix(ea, nn, tb)

        ddt, ddev = timed(dense_step, 5)
        # the shader clock held under this kernel (f64 MFMA at full rate is the chip's most power-hungry loop): the
        # one-wavefront probe of the sustained leg on a side stream, a pair of counters every 0.2 ms, next to 20 launches
        from wespeaker_amd import _lib as _wl
        n_samp = 256
        cbuf = torch.zeros(2 * n_samp, dtype=torch.int64, device=device)
        side = torch.cuda.Stream(device=device)
        torch.cuda.synchronize(device)
        with torch.cuda.stream(side):
            _wl.check(_wl.lib().ws_debug_clock_probe(_wl.ptr(cbuf), n_samp, 20000, side.cuda_stream), "ws_debug_clock_probe")
        t_c = time.perf_counter()
        n_l = 0
        while time.perf_counter() - t_c < n_samp * 0.2e-3 * 1.05:
            dense_step()
            n_l += 1
            if n_l % 4 == 0:
                torch.cuda.current_stream(device).synchronize()
        torch.cuda.synchronize(device)
        raw = cbuf.cpu().numpy().astype(np.uint64).reshape(n_samp, 2).astype(np.float64)
        mhz = np.diff(raw[:, 0]) / np.maximum(np.diff(raw[:, 1]), 1.0) * 100.0      # (counter: nominal 100 MHz)
        mhz = mhz[8:-8]
        tf = 2.0 * n_emb * n_emb * dd / ddev / 1e12
        out_gbs = n_emb * n_emb * 8 / ddev / 1e9
        dense["D%d" % dd] = {
            "workload": "dense %d x %d LLR matrix, D = %d, float64 (1e8 trials per launch)" % (n_emb, n_emb, dd),
            "trials_per_s": n_emb * n_emb / ddt, "ms": ddt * 1e3, "kernel_only_ms": ddev * 1e3,
            "roofline": {"kernel": "plda_gemm_f64_big (v_mfma_f64_16x16x4_f64, 128x128 tiles, 64x64 per wavefront)",
                         "bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / F64_MFMA_PEAK_TFLOPS,
                         "peak_source": "measured: tools/f64_peak_probe.hip, profiles/r06_f64_mfma_peak.jsonl",
                         "frac_of_datasheet_78.6": tf / F64_MFMA_DATASHEET_TFLOPS,
                         "algorithmic_flops_per_launch": 2.0 * n_emb * n_emb * dd,
                         "output_write_gbs": out_gbs, "output_write_frac_of_hbm": out_gbs / HBM_PEAK_GBS,
                         "shader_clock_mhz_under_this_kernel": {"mean": float(mhz.mean()), "min": float(mhz.min()),
                                                                "max": float(mhz.max()), "samples": int(mhz.size)},
                         # 256 CUs x 4 SIMDs x 16 f64 FMAs per cycle x 2: 78.6 TF at 2.4 GHz
                         "peak_at_held_clock": 256 * 4 * 16 * 2 * float(mhz.mean()) * 1e6 / 1e12,
                         "frac_at_held_clock": tf / (256 * 4 * 16 * 2 * float(mhz.mean()) * 1e6 / 1e12)},
            "finite": bool(torch.isfinite(out_holder[0][::997, ::991]).all())}
        out_holder[0] = None
        del ea, tb, tab
    torch.cuda.empty_cache()
    # the same trials as a trial FILE lists them -- grouped by enrollment model (two_cov_plda.py:246-256 walks the file
    # line by line; eval_sv orders its index list this way): 16 consecutive trials of a workgroup then share the
    # enrollment row, which comes from L1 -- half the cache-gather bytes
    order = np.argsort(ie, kind="stable")
    ie_g, it_g = torch.from_numpy(ie[order]).to(device), torch.from_numpy(it[order]).to(device)

    def plda_step_grouped():
        a = plda.prepare_test(emb_tab[:n_emb])
        b = plda.prepare_test(emb_tab[n_emb:])
        return plda.llr_pairs(a, nn, b, ie_g, it_g)

    gdt, _ = timed(plda_step_grouped, k)
    _, grouped_dev = timed(lambda: plda.llr_pairs(e_t, nn, t_t, ie_g, it_g), k)
    # yardstick: what the row-gather path itself delivers (16 lanes read one 1536-B row per index from the same
    # cache-resident table and reduce it; no second operand): the measured peak of a cache gather on this GPU
    from wespeaker_amd import _lib
    probe_out = torch.empty(args.trials, dtype=torch.float64, device=device)
    tt64 = t_t if t_t.dtype == torch.float64 else t_t.double()

    def probe():
        _lib.check(_lib.lib().ws_debug_row_gather(_lib.ptr(tt64), D, _lib.ptr(it_d), args.trials, _lib.ptr(probe_out),
                                                  _lib.current_stream_ptr(device)), "ws_debug_row_gather")

    _, probe_dev = timed(probe, k)
    gather_peak_gbs = args.trials * (D * 8 + 4 + 8) / probe_dev / 1e9
    # a trial gathers one enrollment and one test row (uniform n: D doubles each), reads its two
    # int32 indices and writes one double -- the tables (2 x 10 k x 192 x 8 B = 30.7 MB) live in L2 / Infinity
    # Cache: a CACHE gather (HBM only sees the 8 MB index list and the 8 MB score vector), priced against the
    # measured row-gather yardstick above; a grouped list needs one row per trial
    bpt = 2 * D * 8 + 8 + 8
    bpt_g = D * 8 + 8 + 8
    pair_gbs = args.trials * bpt / pair_dev / 1e9
    grouped_gbs = args.trials * bpt_g / grouped_dev / 1e9
    mat_tf = 2.0 * 1000 * 1000 * D / mat_dev / 1e12
    plda_info = {
        "pairs_trials_per_s": args.trials / pdt, "pairs_ms": pdt * 1e3,
        "pairs_workload": "%d index pairs over 2x%d embeddings D=%d incl. transform, list in random order"
                          % (args.trials, n_emb, D),
        "pairs_kernel_only_trials_per_s": args.trials / pair_dev, "pairs_kernel_only_ms": pair_dev * 1e3,
        "pairs_grouped_trials_per_s": args.trials / gdt, "pairs_grouped_ms": gdt * 1e3,
        "pairs_grouped_workload": "the same trials grouped by enrollment model, as a trial file lists them "
                                  "(incl. transform)",
        "pairs_grouped_kernel_only_trials_per_s": args.trials / grouped_dev,
        "pairs_grouped_kernel_only_ms": grouped_dev * 1e3,
        "roofline": {"kernel": "plda_llr_pairs (16 lanes per trial, double2 gathers of the [g*e] and [t] rows, "
                               "16-lane shuffle reduce)",
                     "bound": "cache-gather", "achieved": pair_gbs, "peak": gather_peak_gbs, "unit": "GB/s",
                     "frac": pair_gbs / gather_peak_gbs, "traffic": None,
                     "algorithmic_bytes_per_trial": bpt,
                     "peak_source": "ws_debug_row_gather: %d random 1536-B rows of the same table in %.1f us "
                                    "(measured in this run)" % (args.trials, probe_dev * 1e6),
                     "note": "gather rows come from L2 / Infinity Cache (30.7 MB of tables), not HBM (%.0f GB/s peak): "
                             "HBM itself only sees the 8 MB index list and the 8 MB score vector per launch"
                             % HBM_PEAK_GBS},
        "roofline_grouped": {"kernel": "plda_llr_pairs on the grouped list (the enrollment row of 16 consecutive trials "
                                       "comes from L1: one test-row gather per trial)",
                             "bound": "cache-gather", "achieved": grouped_gbs, "peak": gather_peak_gbs, "unit": "GB/s",
                             "frac": grouped_gbs / gather_peak_gbs, "algorithmic_bytes_per_trial": bpt_g},
        "dense_1e8": dense or None,
        "matrix_trials_per_s": 1e6 / mdt, "matrix_ms": mdt * 1e3,
        "matrix_workload": "dense 1000x1000 LLR matrix D=%d" % D,
        "matrix_roofline": {"kernel": "plda_gemm_f64 (v_mfma_f64_16x16x4_f64, 64x64 tiles)", "bound": "mfma",
                            "achieved": mat_tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": mat_tf / F64_MFMA_PEAK_TFLOPS,
                            "peak_source": "measured: tools/f64_peak_probe.hip, profiles/r06_f64_mfma_peak.jsonl",
                            "frac_of_datasheet_78.6": mat_tf / F64_MFMA_DATASHEET_TFLOPS,
                            "note": "0.38 GFLOP + 8 MB of output in one %.0f-us launch: latency-bound at this "
                                    "size, not MFMA-bound" % (mat_dev * 1e6)},
        "dtype": "f64"}
    # memory-side counters of the pair kernel (separate rocprofv3 --pmc passes of `bench.py --plda-only`, committed
    # summary: tools/pmc_cmd.sh + tools/pmc_plda_summary.py).  The guide's counters see what leaves the L2 -- Infinity
    # Cache hits included -- so `traffic` is fabric-side bytes per launch, and the rate is quoted against the HBM peak
    # next to the measured gather yardstick
    pmc_path = next((q for q in (os.path.join(ROOT, "profiles", "%s_pmc_plda.json" % r) for r in ("r06", "r05"))
                     if os.path.exists(q)), "")
    if pmc_path and args.trials == 1000000:
        with open(pmc_path) as fp:
            pmc = json.load(fp)
        for key, roof, dev_s, algo in (("plda_llr_pairs_random_list", "roofline", pair_dev, bpt),
                                       ("plda_llr_pairs_grouped_list", "roofline_grouped", grouped_dev, bpt_g)):
            c = pmc[key]
            r = plda_info[roof]
            r["traffic"] = c["traffic_bytes_per_launch"]
            r["traffic_unit"] = "bytes per launch that left the L2 (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC; Infinity-Cache hits are counted)"
            r["traffic_source"] = "profiles/" + os.path.basename(pmc_path)
            r["l2_hit_rate"] = c["l2_hit_rate"]
            r["l2_request_bytes_per_launch"] = c["l2_requests_x128B_bytes"]
            r["fabric_gbs"] = c["traffic_bytes_per_launch"] / dev_s / 1e9
            r["algorithmic_gbs_frac_of_hbm_peak"] = args.trials * algo / dev_s / 1e9 / HBM_PEAK_GBS
            r["fabric_gbs_frac_of_hbm_peak"] = r["fabric_gbs"] / HBM_PEAK_GBS
        plda_info["roofline"]["note"] = (
            "PMC: %.0f %% of the row gathers hit the 4-MB L2 of their XCD; %.2f GB per launch leave it (algorithmic "
            "%.2f GB) = %.1f TB/s in this run -- served by the 256-MB Infinity Cache, which holds both 15-MB tables "
            "whole, and HBM behind it (the memory-side counters count Infinity-Cache hits too, so they cannot split "
            "the two); against HBM's %.0f GB/s spec peak that is %.2f, against the measured gather yardstick `frac`"
            % (100 * plda_info["roofline"]["l2_hit_rate"], plda_info["roofline"]["traffic"] / 1e9,
               args.trials * bpt / 1e9, plda_info["roofline"]["fabric_gbs"] / 1e3, HBM_PEAK_GBS,
               plda_info["roofline"]["fabric_gbs_frac_of_hbm_peak"]))
    if with_cpu_baseline:
        plda_info["cpu_baseline"] = plda_cpu_baseline(
            {"mu": p["mu"], "transform": p["transform"], "psi": p["psi"], "offset": p["offset"],
             "normalize_length": False},
            e_t.cpu().numpy(), t_t.cpu().numpy(), ie, it)
    return plda_info



# ------------------------------------------------------------------------------------------ the one line
LINE_LIMIT = 8192            # the contract reads ONE short stdout line; everything else goes to bench_detail.json


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as fp:
            for ln in fp:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def _r(x, nd=6):
    """floats to `nd` significant digits (the full-precision figure stays in the detail file)."""
    if isinstance(x, float) and math.isfinite(x) and x != 0.0:
        return float("%.*g" % (nd, x))
    return x


def _round_tree(o):
    if isinstance(o, dict):
        return {k: _round_tree(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_round_tree(v) for v in o]
    return _r(o)


ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "launches",
             "avg_launch_ms", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch",
             "whole_step_frac_of_peak", "pmc_mfma_busy_fraction_of_cycles", "kernel_time_share")


def compact_roofline(roof):
    if not roof:
        return roof
    out = _pick(roof, ROOF_KEYS)
    if isinstance(out.get("kernel"), str) and len(out["kernel"]) > 200:
        out["kernel"] = out["kernel"][:197] + "..."
    return out


def compact_line(full, detail_name):
    """The driver's line from the full record: the contract's keys, the headline `roofline` and `cpu_baseline`,
    one figure per other BASELINE config -- like runtime/core/bin/extract_emb_main.cc:100-117 prints ONE RTF figure.
    Everything dropped here is in `detail_name` (written beside this script) unchanged."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                        "scaling", "vs_baseline", "dtype", "data", "config", "headline_backend",
                        "value_one_batch_in_flight", "value_median_over_windows", "value_spread_rel",
                        "value_sustained", "embedding_checksum", "collective", "collective_backend"))
    if full.get("roofline") is not None:
        line["roofline"] = compact_roofline(full["roofline"])
    sc = full.get("self_check")
    if sc:
        line["self_check"] = _pick(sc, ("ok", "max_rel_l2_vs_small_batch_run", "tolerance"))
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "host_cores", "host_cpu_model", "sample"))
    fm = full.get("fast_mode")
    if fm:
        line["fast_mode"] = fm
    pl = full.get("plda")
    if pl:
        cp = _pick(pl, ("pairs_trials_per_s", "pairs_ms", "pairs_kernel_only_ms", "pairs_grouped_trials_per_s",
                        "matrix_trials_per_s", "matrix_ms", "dtype"))
        if pl.get("roofline"):
            cp["roofline"] = _pick(pl["roofline"], ("bound", "achieved", "peak", "unit", "frac", "traffic",
                                                    "algorithmic_bytes_per_trial",
                                                    "algorithmic_gbs_frac_of_hbm_peak", "l2_hit_rate"))
        if pl.get("matrix_roofline"):
            cp["matrix_roofline"] = _pick(pl["matrix_roofline"], ("bound", "achieved", "peak", "unit", "frac"))
        if pl.get("cpu_baseline"):
            cp["cpu_baseline"] = _pick(pl["cpu_baseline"], ("value", "unit", "cores", "kind",
                                                            "vectorised_numpy_trials_per_s"))
        line["plda_trials_per_s"] = full.get("plda_trials_per_s")
        line["plda"] = cp
    cf = full.get("configs")
    if cf:
        cc = {}
        for mname, leg in cf.items():
            if mname == "fixed_size_sets_n1_fp32":
                continue
            e = {"batch": leg.get("batch")}
            for prec in ("fp32", "f16"):
                if prec in leg:
                    b = leg[prec]
                    e[prec] = {"value": b["value"], "ms_per_step": b["ms_per_step"],
                               "whole_step_frac": b["roofline"]["whole_step_frac_of_peak"],
                               "frac": b["roofline"]["frac"]}
            cc[mname] = e
        line["configs"] = cc
        sets = cf.get("fixed_size_sets_n1_fp32")
        if sets:
            line["fixed_size_sets"] = {k: _pick(v, ("value", "ms_per_step", "total_utts", "trials_scored_per_step",
                                                    "embedding_checksum")) for k, v in sets.items()}
    if full.get("set") is not None:
        line["set"] = full["set"]
    line["detail"] = detail_name
    line = _round_tree(line)
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:          # never lose the contract's keys to size again: shed the optional blocks
        for key in ("fixed_size_sets", "configs", "fast_mode", "self_check", "plda"):
            line.pop(key, None)
            text = json.dumps(line)
            if len(text) < LINE_LIMIT:
                break
    assert len(text) < LINE_LIMIT, len(text)
    return text


def emit(full, detail_path):
    """Full record -> the detail file (and stderr's last line names it); compact line -> stdout, once."""
    try:
        with open(detail_path, "w") as fp:
            json.dump(full, fp, indent=1)
            fp.write("\n")
        name = os.path.relpath(detail_path, ROOT)
    except OSError as err:                       # read-only checkout: the line still goes out
        print("bench.py: could not write %s (%s)" % (detail_path, err), file=sys.stderr)
        name = None
    print("bench.py: full record in %s" % name, file=sys.stderr, flush=True)
    print(compact_line(full, name), flush=True)


# ------------------------------------------------------------------------------------------ main
def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU per step (0 = model default)")
    ap.add_argument("--chunk", type=int, default=0, help="engine forward chunk (0 = model default)")
    ap.add_argument("--model", default=None,
                    help="reference constructor name: ECAPA_TDNN[_GLOB]_c{512,1024}, ResNet{18,34,50,221,...}, "
                         "CAMPPlus (default: ECAPA_TDNN_GLOB_c512, or the workload's model)")
    ap.add_argument("--workload", default="default", choices=["default"] + sorted(WORKLOADS),
                    help="vox1o: a VoxCeleb1-O-sized set (4874 utts, 37611 trials), ResNet34 unless --model says "
                         "otherwise; stream10k: 10000 utts through CAM++ -- fixed total size, strong scaling")
    ap.add_argument("--total-utts", type=int, default=0,
                    help="fixed-size set of this many utterances sharded over the ranks (strong scaling)")
    ap.add_argument("--set-trials", type=int, default=-1, help="trial pairs rank 0 scores per step of a fixed-size set")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--trials", type=int, default=1000000)
    ap.add_argument("--cpu-utts", type=int, default=1500)
    ap.add_argument("--windows", type=int, default=5,
                    help="timed windows of --steps steps per back-end (the first one of the headline back-end "
                         "is the contract's timed region = `value`; all of them give median / spread)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plda-only", action="store_true",
                    help="only the PLDA leg (for rocprofv3: the kernel table then holds the plda_* kernels alone)")
    ap.add_argument("--no-configs", action="store_true", help="skip the compact legs of the other BASELINE configs")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the --precision back-end: no other back-ends, no config legs, no PLDA, no CPU "
                         "baseline (for rocprofv3 runs: the kernel statistics then describe one workload)")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes (per-back-end blocks, windows, shard tables, notes); stdout "
                         "carries only the compact line (< 8 KB)")
    ap.add_argument("--plda-dense", action="store_true",
                    help="also time the 1e8-trial dense LLR matrices (D = 192 / 512) with the shader clock sampled "
                         "next to them (opt-in: 1.6 GB of scores per launch)")
    ap.add_argument("--batch-sweep", action="store_true",
                    help="also time the headline workload at per-GPU batches of 512 and 1024 (opt-in)")
    ap.add_argument("--sustain-s", type=float, default=0.0,
                    help="length of the sustained window of the headline back-end (same step function, one timed "
                         "region of this many seconds, the shader clock sampled next to it): `value_sustained`; 0 = off")
    ap.add_argument("--lanes", type=int, default=3,
                    help="batches in flight per GPU (wespeaker_amd.SpeakerModelLanes: one engine + HIP stream per "
                         "lane, step i runs on lane i %% lanes); 1 = one stream, every launch behind the previous one")
    ap.add_argument("--precision", default="fp32", choices=list(BACKENDS),
                    help="back-end of the headline (`value`): fp32 = the reference's arithmetic (default); "
                         "f16x3 / f16 are the declared fast modes, always reported under `backends`")
    return ap.parse_args(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    self_launch_if_needed(argv, args.gpus)

    rank, world, local_rank = parallel.init_distributed("gloo" if STUB else None)
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d (launch with --nproc-per-node equal to --gpus, "
                         "or just `python bench.py --gpus N`)" % (world, args.gpus))
    # WS_SHARE_GPU=1 (debug): all ranks use GPU 0 (with WS_DIST_BACKEND=gloo) so that the N > 1 control
    # flow can be exercised on a single-GPU box
    if STUB:
        device = torch.device("cpu")
    else:
        device = torch.device("cuda", 0 if os.environ.get("WS_SHARE_GPU") else local_rank)
        torch.cuda.set_device(device)

    if args.plda_only:
        info = plda_leg(args, device, not args.no_cpu_baseline)
        emit({"metric": METRIC, "unit": "trials/s", "value": info["pairs_trials_per_s"], "n_gpus": 1,
              "plda_trials_per_s": info["pairs_trials_per_s"], "plda": info}, args.detail_file)
        return

    set_mode = args.workload != "default" or args.total_utts > 0
    w_model, w_utts, w_trials = WORKLOADS.get(args.workload, (None, 0, 0))
    name = args.model or w_model or "ECAPA_TDNN_GLOB_c512"
    total_utts = args.total_utts or w_utts
    set_trials = args.set_trials if args.set_trials >= 0 else (w_trials if args.workload != "default" else 37611)
    fam = name[:5]
    E = EMBED_DIM.get(fam, 256)
    dbatch, dchunk = DEFAULT_BATCH.get(name, DEFAULT_BATCH.get(fam, (256, 256)))
    batch = args.batch or dbatch
    chunk = args.chunk or min(dchunk, batch)
    num_samples = int(args.seconds * 16000)

    def sync():
        if not STUB:
            torch.cuda.synchronize(device)

    active = parallel.collectives_active()            # several ranks (or WS_DIST_FORCE_GROUP=1: a lone rank on RCCL)
    nccl = active and dist.get_backend() == "nccl"

    def fence():
        if active:
            if nccl:
                dist.barrier(device_ids=[device.index])
            else:
                dist.barrier()
        sync()

    def max_over_ranks(x):
        if not active:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device if nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def collective_info():
        """What the data path's collectives ran on, for the driver's SCALE record: backend ("nccl" = RCCL on ROCm),
        the number of ranks the process group holds, and the library version torch reports for it."""
        if not active:
            return {"backend": None, "ranks": 1, "library_version": None}
        ver = None
        if nccl:
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                ver = None
        return {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "library_version": ver}

    def make_model(model_name, embed_dim, max_batch, frames):
        if STUB:
            return StubExtractor(embed_dim, num_samples)
        from wespeaker_amd import NativeSpeakerModel
        sd = synth.synth_state_dict(model_name, 80, embed_dim, seed=42)
        return NativeSpeakerModel(model_name, sd, feat_dim=80, embed_dim=embed_dim, device=device,
                                  max_batch=max_batch, max_frames=frames)

    if STUB:
        fe, T = None, 198
    else:
        from wespeaker_amd import Frontend
        fe = Frontend(16000, 80, device=device)
        T = fe.num_frames(num_samples)

    # ======================================================================= fixed-size set (strong scaling)
    def run_set(model_name, embed_dim, n_utts, n_trials, prec, steps, warmup, per_batch, per_chunk, model=None,
                lanes_obj=None):
        """U utterances cut into contiguous shards (parallel.shard_range), every rank walks its shard in batches,
        one all_gather, rank 0 scores `n_trials` (PLDA LLR + cosine).  One step = the whole set."""
        lo, hi = parallel.shard_range(n_utts, rank, world)
        n_local = hi - lo
        m = model or make_model(model_name, embed_dim, min(per_chunk, max(1, n_local)), T)
        m.set_precision(prec)
        # equal batches (VoxCeleb1-O over 8 ranks: 610 utterances = 2 x 305, not 512 + 98), alternating between the
        # lanes like the steps of the default mode
        n_b = max(1, -(-n_local // per_batch))
        if not STUB and max(1, args.lanes) > 1 and n_b < args.lanes and n_local >= 128 * args.lanes:
            n_b = args.lanes
        cuts = [(n_local * i) // n_b for i in range(n_b + 1)]
        set_lanes = lanes_obj if n_b > 1 else None
        if set_lanes is not None:
            set_lanes.set_precision(prec)
        elif not STUB and max(1, args.lanes) > 1 and n_b > 1:
            from wespeaker_amd import SpeakerModelLanes
            set_lanes = SpeakerModelLanes(model_name, synth.synth_state_dict(model_name, 80, embed_dim, seed=42),
                                          lanes=args.lanes, feat_dim=80, embed_dim=embed_dim, device=device,
                                          max_batch=min(per_chunk, max(1, n_local)), max_frames=T)
            set_lanes.set_precision(prec)
        if STUB:
            g = torch.Generator().manual_seed(99)
            allw = (3000.0 * torch.randn(n_utts, num_samples, generator=g)).round().to(torch.int16)
            wav = allw[lo:hi].contiguous()
        else:
            wav = device_wavs(max(1, n_local), num_samples, device, seed_base=1000 + rank)[:n_local]
        scorer = None
        if rank == 0 and n_trials > 0 and STUB:
            # host stand-in of rank 0's scoring step: the same trial list, cosine in numpy (control flow only)
            ie_s, it_s = synth.synth_trial_pairs(n_trials, n_utts, n_utts, seed=99)

            def scorer(emb):
                # (tests: WS_BENCH_STUB_SCORE_SLEEP seconds inside rank 0's scoring step -- it must show in ms_per_step)
                time.sleep(float(os.environ.get("WS_BENCH_STUB_SCORE_SLEEP", "0")))
                u = torch.nn.functional.normalize(emb.double(), dim=1)
                cos = (u[torch.from_numpy(ie_s).long()] * u[torch.from_numpy(it_s).long()]).sum(1)
                return cos, cos
        if rank == 0 and n_trials > 0 and not STUB:
            from wespeaker_amd import TwoCovPLDA
            from wespeaker_amd import score as wscore
            pp = synth.synth_plda(embed_dim, seed=7)
            plda = TwoCovPLDA.from_params(pp["mu"], pp["transform"], pp["psi"], pp["offset"], False, device=device)
            ie, it = synth.synth_trial_pairs(n_trials, n_utts, n_utts, seed=99)
            ie_d, it_d = torch.from_numpy(ie).to(device), torch.from_numpy(it).to(device)

            def scorer(emb):
                tt = plda.prepare_test(emb)
                llr = plda.llr_pairs(tt, 1, tt, ie_d, it_d)
                tab = wscore.UnitTable(emb, device=device)
                cos = wscore.cosine_pairs(tab, tab, ie_d, it_d)
                return llr, cos

        def one_pass():
            if set_lanes is not None:
                pend = [set_lanes.extract(fe, wav[cuts[i]:cuts[i + 1]]) for i in range(n_b) if cuts[i + 1] > cuts[i]]
                outs = [p_.wait() for p_ in pend]
            else:
                outs = [m.extract(fe, wav[cuts[i]:cuts[i + 1]]) for i in range(n_b) if cuts[i + 1] > cuts[i]]
            local = torch.cat(outs, 0) if outs else torch.zeros((0, embed_dim), dtype=torch.float32, device=device)
            emb = parallel.gather_rows(local, n_utts)
            sc = scorer(emb) if scorer is not None else None
            return emb, sc

        for _ in range(warmup):
            one_pass()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            emb, sc = one_pass()
        fence()
        dt = max_over_ranks(time.perf_counter() - t0)
        m.check_range()
        res = {"model": model_name, "precision": prec, "total_utts": n_utts, "trials_scored_per_step": n_trials,
               "value": n_utts * steps / dt, "unit": "embeddings/s", "ms_per_step": dt / steps * 1e3,
               "steps": steps, "shard": "rank r of %d takes utterances [r*ceil(U/G), (r+1)*ceil(U/G)) "
                                         "(parallel.shard_range = tools/extract_embedding.sh's split rule)" % world,
               "per_rank_utts": parallel.shard_size(n_utts, world), "batch": per_batch,
               "shards": [list(parallel.shard_range(n_utts, r, world)) for r in range(world)],
               "batches_per_rank": [cuts[i + 1] - cuts[i] for i in range(n_b)],
               "batches_in_flight": set_lanes.lanes if set_lanes is not None else 1}
        if sc is not None:
            res["trials_scored_on_rank0"] = int(sc[0].shape[0])
            res["trials_per_s_inside_the_step"] = n_trials * steps / dt
            res["scores_finite"] = bool(torch.isfinite(sc[0]).all()) and bool(torch.isfinite(sc[1]).all())
        if rank == 0:
            assert emb.shape == (n_utts, embed_dim) and bool(torch.isfinite(emb).all())
            res["embedding_checksum"] = float(emb.double().abs().sum().item())
            # order-sensitive: row i weighted by i + 1 (a gather that permuted the shards keeps the plain checksum)
            wts = torch.arange(1, n_utts + 1, dtype=torch.float64, device=emb.device)
            res["embedding_order_checksum"] = float((emb.double().abs().sum(1) * wts).sum().item())
        return res, m

    if set_mode:
        res, _ = run_set(name, E, total_utts, set_trials, args.precision, args.steps, args.warmup, batch, chunk)
        if rank == 0:
            line = {
                "metric": METRIC, "value": res["value"], "unit": "embeddings/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": DTYPE_TEXT[args.precision], "data": "synthetic",
                "config": {"workload": "%s fbank80 E=%d, FIXED set of %d x %.0f s @16 kHz PCM16 utts sharded over %d "
                                       "GPU(s) (wav resident in HBM -> fbank -> CMN -> forward -> all_gather -> rank 0 "
                                       "scores %d trials: PLDA LLR + cosine), one step = the whole set"
                                       % (name, E, total_utts, args.seconds, world, set_trials),
                           "total_utts": total_utts, "trials": set_trials, "per_gpu_batch": batch,
                           "engine_chunk": chunk, "frames": T, "parallelism": "utterance-sharded x%d" % world},
                "set": res,
                "collective_backend": dist.get_backend() if active else None,
                "collective": collective_info(),
            }
            emit(line, args.detail_file)
        if active:
            fence()
            dist.destroy_process_group()
        return

    # ======================================================================= default: per-step batches (weak scaling)
    n_lanes = 1 if STUB else max(1, args.lanes)
    lm = None
    if n_lanes > 1:
        from wespeaker_amd import SpeakerModelLanes
        lm = SpeakerModelLanes(name, synth.synth_state_dict(name, 80, E, seed=42), lanes=n_lanes, feat_dim=80,
                               embed_dim=E, device=device, max_batch=chunk, max_frames=T)
        model = lm.engines[0]
    else:
        model = make_model(name, E, chunk, T)
    if STUB:
        g = torch.Generator().manual_seed(1234 + rank)
        wav = (3000.0 * torch.randn(batch, num_samples, generator=g)).round().to(torch.int16)
    else:
        wav = device_wavs(batch, num_samples, device, seed_base=rank)
    n_total = batch * world

    # N > 1: the all_gather of step k is issued asynchronously (RCCL's own stream) and joined two steps later, so it
    # overlaps the next forward and the ranks are not re-synchronised at every step; everything is joined
    # before the closing fence of a timed window (`drain`), i.e. all K gathers lie inside the timed region
    in_flight = []

    use_lanes = [lm is not None]                                   # (the single-lane windows switch it off)

    def step():
        if use_lanes[0]:
            # step i on lane i % lanes: two (three) batches in flight; a result is joined `lanes` steps later
            res = lm.extract(fe, wav)
            if not active:
                in_flight.append(res)
            else:
                with torch.cuda.stream(lm.streams[res.lane]):     # the gather waits for this lane only
                    in_flight.append(parallel.gather_rows_async(res.tensor, n_total))
            return in_flight.pop(0).wait() if len(in_flight) > max(2, n_lanes) else None
        emb = model.extract(fe, wav)                              # (B, E) on this GPU
        if not active:
            return emb
        in_flight.append(parallel.gather_rows_async(emb, n_total))
        return in_flight.pop(0).wait() if len(in_flight) > 2 else None

    def drain(last):
        while in_flight:
            last = in_flight.pop(0).wait()
        return last

    def timed_window(steps):
        """EXACTLY `steps` steps between two fences; max over ranks; HIP events bracket only the dominant
        kernel class inside it (events around all ~45 launches per chunk cost ~12 %)."""
        fence()
        events = not STUB and not use_lanes[0]      # with several batches in flight the kernels of the lanes share
        if events:                                  # the CUs: a launch's begin -> end is not its own duration
            model.profile(1)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        out = drain(out)
        fence()
        dt = max_over_ranks(time.perf_counter() - t0)
        prof = None
        if events:
            prof = model.profile_read()[DOMINANT]
            model.profile(False)
        return dt, prof, out

    def roofline_block(m, model_name, prec, steps, dt0, profs, breakdown, bsteps, per_batch):
        g = profs[0]
        achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        ach_all = [p["flops"] / (p["ms"] * 1e-3) / 1e12 for p in profs if p["ms"] > 0]
        peak = FP32_MFMA_PEAK_TFLOPS if prec == "fp32" else F16_MFMA_PEAK_TFLOPS
        total_ms = sum(breakdown[c]["ms"] for c in breakdown)
        gemm_ms = sum(breakdown[c]["ms"] for c in breakdown if c.startswith("gemm"))
        flops_utt = m.flops(1, T)
        whole = flops_utt * per_batch * steps / dt0 / 1e12       # per GPU
        roof = {
            "kernel": kernel_text(model_name, prec), "bound": "mfma",
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": None,
            "achieved_median_over_windows": statistics.median(ach_all) if ach_all else None,
            "launches": g["launches"], "avg_launch_ms": g["ms"] / max(1, g["launches"]),
            "algorithmic_flops_per_launch": g["flops"] / max(1, g["launches"]),
            "algorithmic_bytes_per_launch": g["bytes"] / max(1, g["launches"]),
            "kernel_time_share": breakdown[DOMINANT]["ms"] / total_ms if total_ms else None,
            "all_gemm_time_share": gemm_ms / total_ms if total_ms else None,
            "forward_flops_per_utt": flops_utt,
            "whole_step_tflops_per_gpu": whole, "whole_step_frac_of_peak": whole / peak,
            "event_ms_per_step_by_class_untimed_pass":
                {c: round(breakdown[c]["ms"] / bsteps, 4) for c in breakdown},
        }
        if prec == "f16x3":
            roof["note"] = ("achieved counts ALGORITHMIC flops (2MNK); this back-end issues 3 MFMA passes per "
                            "product, i.e. %.0f TFLOP/s of f16 MFMA work = %.3f of the dense f16 peak"
                            % (3 * achieved, 3 * achieved / peak))
        # HBM traffic of the dominant kernel class: separate rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE and WRITE_SIZE cannot share a pass); the committed aggregate of the newest round
        tag = "" if model_name == "ECAPA_TDNN_GLOB_c512" else "_" + model_name
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            pmc_path = os.path.join(ROOT, "profiles", "%s_pmc_dominant_kernel_%s%s.json" % (rnd, prec, tag))
            if os.path.exists(pmc_path):
                with open(pmc_path) as fpmc:
                    pmc = json.load(fpmc)
                # per launch of THIS line's class: the counted kernels' bytes per forward / the launches per step
                per_fwd = pmc.get("dominant_bytes_per_forward")
                lps = g["launches"] / float(steps)             # launches of the class per step (= per forward)
                roof["traffic"] = per_fwd / lps if per_fwd and lps else pmc.get("traffic_bytes_per_launch")
                roof["traffic_unit"] = "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC)"
                if "whole_forward" in pmc:
                    roof["whole_forward_hbm_bytes"] = pmc["whole_forward"]["traffic_bytes"]
                    roof["whole_forward_kernel_launches"] = pmc["whole_forward"]["kernel_launches_per_forward"]
                roof["traffic_source"] = "profiles/" + os.path.basename(pmc_path)
                if "mfma_busy_fraction_of_cycles" in pmc:
                    roof["pmc_mfma_busy_fraction_of_cycles"] = pmc["mfma_busy_fraction_of_cycles"]
                break
        return roof

    def run_backend(prec, steps, warmup, windows):
        # several batches in flight on every back-end (round 3 had to keep the binary16 ones on one stream:
        # DESIGN.md 6.0, closed in round 4)
        lanes_here = lm is not None
        if lanes_here:
            lm.set_precision(prec)
        else:
            model.set_precision(prec)
        use_lanes[0] = lanes_here
        for _ in range(warmup):
            step()
        drain(None)
        dts, profs, out = [], [], None
        for _ in range(windows):
            dt, prof, out = timed_window(steps)
            dts.append(dt)
            profs.append(prof)
        (lm if lanes_here else model).check_range()            # binary16 back-ends: loud on overflow
        vals = [n_total * steps / d for d in dts]
        block = {"precision": prec, "dtype": DTYPE_TEXT[prec], "value": vals[0], "unit": "embeddings/s",
                 "ms_per_step": dts[0] / steps * 1e3, "steps": steps, "warmup": warmup,
                 "batches_in_flight": n_lanes if lanes_here else 1,
                 "windows_embeddings_per_s": [round(v, 1) for v in vals],
                 "median": statistics.median(vals), "min": min(vals), "max": max(vals),
                 "spread_rel": (max(vals) - min(vals)) / statistics.median(vals)}
        if lanes_here:
            # the same steps on ONE lane (every launch behind the previous one): the per-kernel durations of the
            # roofline block come from these windows -- with several batches in flight the lanes' kernels overlap
            use_lanes[0] = False
            for _ in range(2):
                step()
            drain(None)
            sdts, profs = [], []
            for _ in range(max(1, min(windows, 3))):
                dt, prof, _o = timed_window(steps)
                sdts.append(dt)
                profs.append(prof)
            svals = [n_total * steps / d for d in sdts]
            block["one_batch_in_flight"] = {"value": svals[0], "ms_per_step": sdts[0] / steps * 1e3,
                                            "windows_embeddings_per_s": [round(v, 1) for v in svals]}
        if not STUB:
            # untimed pass with every kernel class bracketed, for the per-class breakdown only
            use_lanes[0] = False
            model.profile(True)
            bsteps = min(steps, 5)
            for _ in range(bsteps):
                step()
            drain(None)
            fence()
            breakdown = model.profile_read()
            model.profile(False)
            block["roofline"] = roofline_block(model, name, prec, steps, dts[0], profs, breakdown, bsteps, batch)
            if lanes_here:
                block["roofline"]["note_lanes"] = (
                    "`achieved` / `frac` / `avg_launch_ms`: HIP events around the class's launches in timed windows "
                    "of the same run with ONE batch in flight (`one_batch_in_flight`), where a launch's begin -> end "
                    "is its own duration; `value` and `whole_step_*` are the windows with %d batches in flight "
                    "(the lanes' kernels share the CUs there)" % n_lanes)
            use_lanes[0] = lanes_here
        return block, out

    # ---- headline back-end first: W warmup steps, then the contract's timed region (window 0)
    blocks = {}
    blocks[args.precision], all_emb = run_backend(args.precision, args.steps, args.warmup, max(1, args.windows))

    # ---- sustained window: the contract's region is K steps (80 ms at the default K); the same
    # step function held for --sustain-s seconds in ONE timed region, with the shader clock sampled every 10 ms by a
    # one-wavefront probe on a side stream (csrc/probes.hip), so that the peak the fractions are priced against can be
    # compared with the clock the chip held under this very load
    sustained = None
    if rank == 0 and world == 1 and not STUB and args.sustain_s > 0:
        from wespeaker_amd import _lib as _wl
        use_lanes[0] = lm is not None
        (lm if lm is not None else model).set_precision(args.precision)
        est_ms = blocks[args.precision]["ms_per_step"]
        n_steps = max(args.steps, int(math.ceil(args.sustain_s * 1e3 / est_ms)))
        period_ticks = 1000000                                          # 10 ms at the counter's nominal 100 MHz
        n_samp = min(4096, int(n_steps * est_ms / 10.0) + 8)
        buf = torch.zeros(2 * n_samp, dtype=torch.int64, device=device)
        side = torch.cuda.Stream(device=device)
        for _ in range(2):
            step()
        drain(None)
        fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            e0.record(side)
            _wl.check(_wl.lib().ws_debug_clock_probe(_wl.ptr(buf), n_samp, period_ticks, side.cuda_stream),
                      "ws_debug_clock_probe")
            e1.record(side)
        sdt, _p, _o = timed_window(n_steps)
        side.synchronize()
        raw = buf.cpu().numpy().astype(np.uint64).reshape(n_samp, 2)
        cyc, rt = raw[:, 0].astype(np.float64), raw[:, 1].astype(np.float64)
        probe_s = e0.elapsed_time(e1) * 1e-3
        rt_hz = (rt[-1] - rt[0]) / probe_s * (n_samp / max(1.0, n_samp - 1.0)) if probe_s > 0 else 1e8
        inside = (rt - rt[0]) / max(rt_hz, 1.0) <= sdt                    # samples taken while the window ran
        d_c, d_r = np.diff(cyc), np.diff(rt)
        mhz = (d_c / np.maximum(d_r, 1.0)) * rt_hz / 1e6
        mhz_in = mhz[inside[1:]] if inside[1:].any() else mhz
        sval = n_total * n_steps / sdt
        sustained = {"seconds": sdt, "steps": n_steps, "value": sval, "unit": "embeddings/s",
                     "ms_per_step": sdt / n_steps * 1e3,
                     "rel_to_value": sval / blocks[args.precision]["value"] - 1.0,
                     "batches_in_flight": n_lanes if lm is not None else 1,
                     "shader_clock_mhz": {"mean": float(mhz_in.mean()), "min": float(mhz_in.min()),
                                          "max": float(mhz_in.max()), "samples": int(mhz_in.size),
                                          "sample_period_ms": 10.0,
                                          "first_100ms_mean": float(mhz[:10].mean()),
                                          "constant_counter_hz": rt_hz},
                     "whole_step_frac_of_peak": (model.flops(1, T) * batch * n_steps / sdt / 1e12) /
                                                (FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else F16_MFMA_PEAK_TFLOPS),
                     "how": "one timed region (barrier + synchronize on both sides) of the same step function as "
                            "`value`; clock = d(s_memtime) / d(s_memrealtime) of a one-wavefront probe kernel "
                            "running on a side stream next to the window (ws_debug_clock_probe), the constant "
                            "counter's rate calibrated against HIP events around the probe"}
        # the peak of the guide (157.3 TF fp32) is quoted at the boost clock: the same fraction against the clock held
        if args.precision == "fp32":
            # 256 CUs x 4 SIMDs x (32x32x2 MACs per 64-cycle MFMA = 32 MACs per cycle) x 2 flops: 157.3 TF at 2.4 GHz
            held_peak = 256 * 4 * 32 * 2 * sustained["shader_clock_mhz"]["mean"] * 1e6 / 1e12
            sustained["fp32_mfma_peak_at_held_clock_tflops"] = held_peak
            sustained["whole_step_frac_of_peak_at_held_clock"] = (model.flops(1, T) * batch * n_steps / sdt / 1e12) / held_peak
        use_lanes[0] = False

    # ---- rows of the TIMED output against a small-batch run of the same utterances (the timed kernels did the work)
    self_check = None
    if rank == 0 and not STUB:
        rows = sorted({0, 1, batch // 2, batch - 1})
        small = model.extract(fe, wav[rows])
        big = all_emb[rows].to(small.device)
        rel = ((big - small).norm(dim=1) / small.norm(dim=1)).max().item()
        tol = 2e-4 if args.precision != "f16" else 5e-3
        self_check = {"rows": rows, "max_rel_l2_vs_small_batch_run": rel, "tolerance": tol, "ok": bool(rel <= tol),
                      "what": "rows of the last timed step's output vs the same utterances extracted as a batch of "
                              "%d (other tile shapes / kernels; not bit-equal: a row's position in the 64-row tiles "
                              "moves the fp32 summation order of the SE / context statistics)" % len(rows)}
        assert self_check["ok"], self_check

    if not args.headline_only and not STUB:
        for other in [m for m in BACKENDS if m != args.precision]:
            blocks[other], _ = run_backend(other, max(3, min(args.steps, 10)), 2, max(1, min(args.windows, 3)))
    model.set_precision(args.precision)
    if lm is not None and args.precision == "fp32":
        lm.set_precision("fp32")
    use_lanes[0] = False

    # ---- the other BASELINE.json configs, compact: per model one fp32 leg with its dominant-class fraction (HIP
    # events) + the f16 rate; and the two fixed-size sets (configs 2 / 3) at this N
    configs = None
    if rank == 0 and world == 1 and not args.headline_only and not args.no_configs and not STUB \
            and name == "ECAPA_TDNN_GLOB_c512":
        configs = {}
        keep = {}
        for cname in ("ECAPA_TDNN_GLOB_c1024", "ResNet34", "ResNet221", "CAMPPlus"):
            cfam = cname[:5]
            cE = EMBED_DIM.get(cfam, 256)
            cb, cc = DEFAULT_BATCH.get(cname, DEFAULT_BATCH.get(cfam, (256, 256)))
            cl = None
            if n_lanes > 1:
                cl = SpeakerModelLanes(cname, synth.synth_state_dict(cname, 80, cE, seed=42), lanes=n_lanes,
                                       feat_dim=80, embed_dim=cE, device=device, max_batch=cc, max_frames=T)
                cm = cl.engines[0]
            else:
                cm = make_model(cname, cE, cc, T)
            cw = wav if cb == batch else device_wavs(cb, num_samples, device, seed_base=77)
            leg = {"model": cname, "batch": cb, "unit": "embeddings/s"}
            for prec in ("fp32", "f16"):
                cm.set_precision(prec)
                ks = 3 if cname == "ResNet221" else 6
                for _ in range(2):
                    cm.extract(fe, cw)
                sync()
                cm.profile(1)
                tb = time.perf_counter()
                for _ in range(ks):
                    cm.extract(fe, cw)
                sync()
                cdt = time.perf_counter() - tb
                pr = cm.profile_read()[DOMINANT]
                cm.profile(False)
                cm.check_range()
                peak = FP32_MFMA_PEAK_TFLOPS if prec == "fp32" else F16_MFMA_PEAK_TFLOPS
                ach = pr["flops"] / (pr["ms"] * 1e-3) / 1e12 if pr["ms"] > 0 else 0.0
                whole = cm.flops(1, T) * cb * ks / cdt / 1e12
                leg[prec] = {"value": cb * ks / cdt, "ms_per_step": cdt / ks * 1e3, "steps": ks,
                             "batches_in_flight": 1,
                             "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                                          "frac": ach / peak, "kernel": "dominant class (every conv/linear with "
                                          "N > 64), HIP events inside this window",
                                          "whole_step_frac_of_peak": whole / peak}}
                if cl is not None:
                    # the headline's mode: n_lanes batches in flight (the roofline above stays the one-lane window's)
                    cl.set_precision(prec)
                    for _ in range(2 * n_lanes):
                        cl.extract(fe, cw)
                    sync()
                    tb = time.perf_counter()
                    for _ in range(ks * n_lanes):
                        cl.extract(fe, cw)
                    sync()
                    cdt2 = time.perf_counter() - tb
                    leg[prec]["value_one_batch_in_flight"] = leg[prec]["value"]
                    leg[prec]["value"] = cb * ks * n_lanes / cdt2
                    leg[prec]["ms_per_step"] = cdt2 / (ks * n_lanes) * 1e3
                    leg[prec]["batches_in_flight"] = n_lanes
                    leg[prec]["roofline"]["whole_step_frac_of_peak"] = \
                        cm.flops(1, T) * cb * ks * n_lanes / cdt2 / 1e12 / peak
            configs[cname] = leg
            keep[cname] = (cm, cl)
        # configs 2 / 3 as fixed-size sets on this one GPU (N > 1: `bench.py --gpus N --workload vox1o|stream10k`)
        sets = {}
        for wl, mname in (("vox1o", "ResNet34"), ("vox1o", "ResNet221"), ("stream10k", "CAMPPlus")):
            _, u, tr = WORKLOADS[wl]
            cfam = mname[:5]
            cb, cc = DEFAULT_BATCH.get(mname, DEFAULT_BATCH.get(cfam, (256, 256)))
            r, _ = run_set(mname, EMBED_DIM.get(cfam, 256), u, tr, "fp32", 1 if mname == "ResNet221" else 2, 1,
                           cb, cc, model=keep[mname][0], lanes_obj=keep[mname][1])
            sets["%s_%s" % (wl, mname)] = r
        configs["fixed_size_sets_n1_fp32"] = sets
        del keep

    # ---- the same workload at larger per-GPU batches (fp32 headline back-end): what the fixed per-launch costs (~45
    # launches, the latency-bound SE / pooling kernels, the partial last rounds of tiles) take at batch 256
    batch_sweep = None
    if rank == 0 and world == 1 and not args.headline_only and not STUB and not args.batch and args.batch_sweep:
        batch_sweep = {}
        for b in (512, 1024):
            bm = make_model(name, E, b, T)
            bm.set_precision(args.precision)
            bw = device_wavs(b, num_samples, device, seed_base=500 + b)
            for _ in range(2):
                bm.extract(fe, bw)
            sync()
            bm.profile(1)
            ks = 5
            tb = time.perf_counter()
            for _ in range(ks):
                bm.extract(fe, bw)
            sync()
            bdt = time.perf_counter() - tb
            pr = bm.profile_read()[DOMINANT]
            bm.profile(False)
            peak = FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else F16_MFMA_PEAK_TFLOPS
            batch_sweep[str(b)] = {"value": b * ks / bdt, "unit": "embeddings/s", "ms_per_step": bdt / ks * 1e3,
                                   "steps": ks, "engine_chunk": b, "batches_in_flight": 1,
                                   "dominant_frac": (pr["flops"] / (pr["ms"] * 1e-3) / 1e12 / peak) if pr["ms"] else None}
            del bm
            if n_lanes > 1 and args.precision == "fp32":
                # ... and with the headline's number of batches in flight
                bl = SpeakerModelLanes(name, synth.synth_state_dict(name, 80, E, seed=42), lanes=n_lanes, feat_dim=80,
                                       embed_dim=E, device=device, max_batch=b, max_frames=T)
                bl.set_precision(args.precision)
                for _ in range(2 * n_lanes):
                    bl.extract(fe, bw)
                sync()
                tb = time.perf_counter()
                for _ in range(ks * n_lanes):
                    bl.extract(fe, bw)
                sync()
                bdt = time.perf_counter() - tb
                batch_sweep[str(b)]["value_%d_batches_in_flight" % n_lanes] = b * ks * n_lanes / bdt
                del bl
            del bw

    # ---- PLDA leg (rank 0 scores after the gather; 1 M synthetic trial pairs over 10 k embeddings)
    plda_info = None
    if rank == 0 and not args.headline_only and not STUB:
        plda_info = plda_leg(args, device, world == 1 and not args.no_cpu_baseline)

    if rank == 0:
        head = blocks[args.precision]
        line = {
            "metric": METRIC,
            "value": head["value"],
            "unit": "embeddings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": head["dtype"],
            "data": "synthetic",
            "config": {"workload": "%s fbank80 E=%d, %d x %.0f s @16 kHz PCM16 utts per GPU per step "
                                   "(wav resident in HBM -> fbank -> CMN -> forward -> all_gather)"
                                   % (name, E, batch, args.seconds),
                       "per_gpu_batch": batch, "global_batch": n_total, "frames": T,
                       "engine_chunk": chunk, "parallelism": "utterance-sharded x%d" % world,
                       "batches_in_flight_per_gpu": head.get("batches_in_flight", 1)},
            "headline_backend": args.precision,
            "headline_is_parity_grade": args.precision == "fp32",
            "headline_note": "value/dtype/roofline describe the %s back-end; fp32 = the arithmetic of the "
                             "reference path (parity-grade); f16x3 and f16 are declared fast modes, see "
                             "`backends`" % args.precision,
            "value_one_batch_in_flight": (head.get("one_batch_in_flight") or {}).get("value"),
            "value_median_over_windows": head["median"],
            "value_sustained": sustained["value"] if sustained else None,
            "sustained": sustained,
            "value_spread_rel": head["spread_rel"],
            "roofline": head.get("roofline"),
            "self_check": self_check,
            "backends": blocks,
            "fast_mode": ({"precision": "f16", "value": blocks["f16"]["value"],
                           "median": blocks["f16"]["median"],
                           "roofline_frac": blocks["f16"]["roofline"]["frac"]} if "f16" in blocks else None),
            "plda_trials_per_s": plda_info["pairs_trials_per_s"] if plda_info else None,
            "plda": plda_info,
        }
        if configs:
            line["configs"] = configs
        if batch_sweep:
            line["throughput_vs_per_gpu_batch"] = batch_sweep
        line["embedding_checksum"] = float(all_emb.double().abs().sum().item())
        line["collective_backend"] = dist.get_backend() if active else None
        line["collective"] = collective_info()
        if world == 1 and not args.no_cpu_baseline and not args.headline_only and not STUB:
            heavy = name.startswith("ResNet") and name not in ("ResNet18", "ResNet34")
            line["cpu_baseline"] = cpu_baseline(name, E, args.cpu_utts, budget_s=6.0 if heavy else 8.0)
        assert all_emb.shape == (n_total, E) and bool(torch.isfinite(all_emb).all())
        emit(line, args.detail_file)
    if active:
        fence()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
