"""Cosine scoring and AS-norm / S-norm on the MI355X: the file-level tools of the reference
(`wespeaker/bin/score.py`, `wespeaker/bin/score_norm.py`) with their numpy / sklearn back-end
replaced by the C-ABI kernels `ws_cos_prepare / ws_cos_pairs / ws_cohort_stats / ws_asnorm_pairs`.

Same function names, arguments, file formats and score-file columns as the reference:
  calculate_mean_from_kaldi_vec   bin/score.py:25-36
  trials_cosine_score             bin/score.py:38-72
  main                            bin/score.py:75-93
  get_mean_std                    bin/score_norm.py:26-36
  score_norm (= its `main`)       bin/score_norm.py:54-115

There is no CPU fallback: everything numeric runs through libwespeaker_amd.so.
"""
import os
from pathlib import Path

import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream_ptr, ptr
from .kaldi_io import read_vec_scp

_SCRATCH_BYTES = 1 << 30     # score rows in flight for the cohort statistics (<= 1 GiB of 288)


def _device(device=None):
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def _f32(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).to(dev)


def _i32(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=torch.int32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.int32))).to(dev)


class UnitTable:
    """Mean-subtracted, L2-normalised embeddings in the padded device layout of the C-ABI
    (include/wespeaker_amd.h: (ws_cos_table_rows(n), ws_cos_table_ld(dim)) float32)."""

    def __init__(self, emb, mean_vec=None, device=None):
        dev = _device(device)
        L = _lib.lib()
        emb = _f32(emb, dev)
        if emb.dim() != 2:
            raise ValueError("embeddings must be (n, dim)")
        self.n, self.dim = int(emb.shape[0]), int(emb.shape[1])
        self.device = dev
        rows, ld = L.ws_cos_table_rows(self.n), L.ws_cos_table_ld(self.dim)
        self.unit = torch.empty((max(rows, 4), ld), dtype=torch.float32, device=dev)
        self.mag = torch.empty((max(self.n, 1),), dtype=torch.float32, device=dev)
        mv = None
        if mean_vec is not None and not (np.isscalar(mean_vec) and float(mean_vec) == 0.0):
            mv = _f32(np.array(np.broadcast_to(np.asarray(mean_vec, dtype=np.float32), (self.dim,))), dev)
        with torch.cuda.device(dev):
            check(L.ws_cos_prepare(ptr(emb), ptr(mv) if mv is not None else None, self.n, self.dim,
                                   ptr(self.unit), ptr(self.mag), current_stream_ptr(dev)),
                  "ws_cos_prepare")
        self.mag = self.mag[:self.n]


def cosine_pairs(table_a: UnitTable, table_b: UnitTable, idx_a, idx_b) -> torch.Tensor:
    """cosine of (table_a[idx_a[p]], table_b[idx_b[p]]) for every trial p -> float32 GPU tensor."""
    dev = table_a.device
    ia, ib = _i32(idx_a, dev), _i32(idx_b, dev)
    if ia.numel() != ib.numel():
        raise ValueError("index lists differ in length")
    out = torch.empty((ia.numel(),), dtype=torch.float32, device=dev)
    if ia.numel() == 0:
        return out
    if int(ia.min()) < 0 or int(ia.max()) >= table_a.n or int(ib.min()) < 0 or int(ib.max()) >= table_b.n:
        raise IndexError("trial index out of range")
    with torch.cuda.device(dev):
        check(_lib.lib().ws_cos_pairs(ptr(table_a.unit), ptr(table_b.unit), table_a.dim, ptr(ia),
                                      ptr(ib), ia.numel(), ptr(out), current_stream_ptr(dev)),
              "ws_cos_pairs")
    return out


def cosine_matrix(table_a: UnitTable, table_b: UnitTable) -> torch.Tensor:
    """(n_a, n_b) cosine matrix (a view of the padded GEMM output)."""
    dev = table_a.device
    L = _lib.lib()
    ldo = max(4, L.ws_cos_table_rows(table_b.n))
    out = torch.empty((max(table_a.n, 1), ldo), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(L.ws_cos_matrix(ptr(table_a.unit), table_a.n, ptr(table_b.unit), table_b.n, table_a.dim,
                              ptr(out), ldo, current_stream_ptr(dev)), "ws_cos_matrix")
    return out[:table_a.n, :table_b.n]


def cohort_stats(table: UnitTable, cohort: UnitTable, top_n: int, scratch_bytes=_SCRATCH_BYTES):
    """get_mean_std on prepared tables -> (mean, std) float32 GPU tensors of length table.n."""
    dev = table.device
    L = _lib.lib()
    lds = L.ws_cos_table_rows(cohort.n)
    rows = max(128, min(table.n, scratch_bytes // 4 // max(lds, 1)))
    rows = rows if rows >= table.n else (rows // 128) * 128
    scratch = torch.empty((rows * lds,), dtype=torch.float32, device=dev)
    mean = torch.empty((max(table.n, 1),), dtype=torch.float32, device=dev)
    sd = torch.empty((max(table.n, 1),), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(L.ws_cohort_stats(ptr(table.unit), table.n, ptr(cohort.unit), cohort.n, table.dim,
                                int(top_n), ptr(scratch), scratch.numel(), ptr(mean), ptr(sd),
                                current_stream_ptr(dev)), "ws_cohort_stats")
        torch.cuda.current_stream(dev).synchronize()     # scratch is freed on return
    return mean[:table.n], sd[:table.n]


def get_mean_std(emb, cohort, top_n):
    """bin/score_norm.py:26-36 (numpy in, numpy out; computed on the GPU)."""
    t, c = UnitTable(emb), UnitTable(cohort)
    m, s = cohort_stats(t, c, top_n)
    return m.cpu().numpy(), s.cpu().numpy()


def asnorm_pairs(scores, idx_e, idx_t, e_mean, e_std, t_mean, t_std) -> torch.Tensor:
    dev = e_mean.device if isinstance(e_mean, torch.Tensor) else _device()
    s, ie, it = _f32(scores, dev), _i32(idx_e, dev), _i32(idx_t, dev)
    out = torch.empty_like(s)
    if s.numel() == 0:
        return out
    with torch.cuda.device(dev):
        check(_lib.lib().ws_asnorm_pairs(ptr(s), ptr(ie), ptr(it), ptr(_f32(e_mean, dev)),
                                         ptr(_f32(e_std, dev)), ptr(_f32(t_mean, dev)),
                                         ptr(_f32(t_std, dev)), s.numel(), ptr(out),
                                         current_stream_ptr(dev)), "ws_asnorm_pairs")
        torch.cuda.current_stream(dev).synchronize()     # temporaries above are freed on return
    return out


# ----------------------------------------------------------------------------- bin/score.py
def calculate_mean_from_kaldi_vec(scp_path):
    vecs = read_vec_scp(scp_path)
    mean_vec = None
    for vec in vecs.values():
        if mean_vec is None:
            mean_vec = np.zeros_like(vec)
        mean_vec += vec
    return mean_vec / len(vecs)


def _read_table(path):
    rows = []
    with open(path, "r", encoding="utf8") as fin:
        for line in fin:
            tokens = line.strip().split()
            if tokens:
                rows.append(tokens)
    return rows


def trials_cosine_score(eval_scp_path="", store_dir="", mean_vec=None, trials=()):
    if mean_vec is None or not os.path.exists(mean_vec):
        mean = None
    else:
        mean = np.load(mean_vec)
    emb_dict = read_vec_scp(eval_scp_path)
    names = list(emb_dict.keys())
    index = {k: i for i, k in enumerate(names)}
    table = UnitTable(np.vstack([emb_dict[k] for k in names]), mean) if names else None
    for trial in trials:
        store_path = os.path.join(store_dir, os.path.basename(trial) + ".score")
        rows = _read_table(trial)
        ia = np.fromiter((index[r[0]] for r in rows), dtype=np.int32, count=len(rows))
        ib = np.fromiter((index[r[1]] for r in rows), dtype=np.int32, count=len(rows))
        scores = cosine_pairs(table, table, ia, ib).cpu().numpy() if rows else []
        with open(store_path, "w") as w_f:
            for segs, cos_score in zip(rows, scores):
                if len(segs) == 3:   # enroll_name test_name target/nontarget
                    w_f.write("{} {} {:.5f} {}\n".format(segs[0], segs[1], cos_score, segs[2]))
                else:                # enroll_name test_name
                    w_f.write("{} {} {:.5f}\n".format(segs[0], segs[1], cos_score))


def main(exp_dir, eval_scp_path, cal_mean, cal_mean_dir, *trials):
    if not cal_mean:
        print("Do not do mean normalization for evaluation embeddings.")
        mean_vec_path = None
    else:
        scp_path = os.path.join(cal_mean_dir, "xvector.scp")
        print("Calculate mean statistics from {}.".format(scp_path))
        mean_vec = calculate_mean_from_kaldi_vec(scp_path)
        mean_vec_path = os.path.join(cal_mean_dir, "mean_vec.npy")
        np.save(mean_vec_path, mean_vec)
    store_score_dir = os.path.join(exp_dir, "scores")
    Path(store_score_dir).mkdir(parents=True, exist_ok=True)
    trials_cosine_score(eval_scp_path, store_score_dir, mean_vec_path, trials)


# ------------------------------------------------------------------------ bin/score_norm.py
def score_norm(score_norm_method, top_n, trial_score_file, score_norm_file, cohort_emb_scp,
               eval_emb_scp, mean_vec_path=None):
    """bin/score_norm.py:54-115 (`main`).  Output line: enroll test normed_score label
    enroll_mag test_mag enroll_mean test_mean."""
    if score_norm_method not in ("asnorm", "snorm"):
        raise ValueError(score_norm_method)
    if not mean_vec_path:
        print("Do not do mean normalization for evaluation embeddings.")
        mean_vec = None
    else:
        assert os.path.exists(mean_vec_path), "mean_vec file ({}) does not exist !!!".format(
            mean_vec_path)
        mean_vec = np.load(mean_vec_path)

    rows = _read_table(trial_score_file)
    enroll_list = sorted(set(r[0] for r in rows))      # remove overlap and sort (:77-79)
    test_list = sorted(set(r[1] for r in rows))
    eval_emb = read_vec_scp(eval_emb_scp)
    enroll = UnitTable(np.vstack([eval_emb[u] for u in enroll_list]), mean_vec)
    test = UnitTable(np.vstack([eval_emb[u] for u in test_list]), mean_vec)
    cohort_emb = read_vec_scp(cohort_emb_scp)
    cohort = UnitTable(np.vstack(list(cohort_emb.values())), mean_vec)

    n_top = int(top_n) if score_norm_method == "asnorm" else cohort.n     # score_norm.py:84-89
    e_mean, e_std = cohort_stats(enroll, cohort, n_top)
    t_mean, t_std = cohort_stats(test, cohort, n_top)

    e_index = {u: i for i, u in enumerate(enroll_list)}
    t_index = {u: i for i, u in enumerate(test_list)}
    ie = np.fromiter((e_index[r[0]] for r in rows), dtype=np.int32, count=len(rows))
    it = np.fromiter((t_index[r[1]] for r in rows), dtype=np.int32, count=len(rows))
    raw = np.array([float(r[2]) for r in rows], dtype=np.float32)
    normed = asnorm_pairs(raw, ie, it, e_mean, e_std, t_mean, t_std).cpu().numpy()
    e_mag, t_mag = enroll.mag.cpu().numpy(), test.mag.cpu().numpy()
    e_mean_h, t_mean_h = e_mean.cpu().numpy(), t_mean.cpu().numpy()
    with open(score_norm_file, "w", encoding="utf-8") as fout:
        for r, s, a, b in zip(rows, normed, ie, it):
            fout.write("{} {} {:.5f} {} {:.4f} {:.4f} {:.4f} {:.4f}\n".format(
                r[0], r[1], s, r[3], e_mag[a], t_mag[b], e_mean_h[a], t_mean_h[b]))
