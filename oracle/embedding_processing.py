"""ORACLE (test infrastructure, never shipped): numpy restatement of the reference's
embedding-processing chain (mean-subtract / length-norm / lda).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows wespeaker/utils/embedding_processing.py in /root/reference:
  :72-131   Lda.compute_mean_and_lda_scatter_matrices (speakers with one utterance skipped; count-weighted
            mean, np.cov(..., bias=True, fweights=counts) between-class, count-weighted within-class)
  :133-175  Lda.__init__ (eigh of WC, eigenvalue floor max(E) * eps, whitening T1, eigh of T1 BC T1',
            last `dim` eigenvectors)
  :177-195  Lda.__call__, Length_norm.__call__;  :204-216 MeanSubtraction

PINNED: tests/golden/embd_proc_ref.npz holds outputs of the reference's own
EmbeddingProcessingChain("mean-subtract | length-norm | lda | length-norm") built from ark/scp files
(oracle/make_golden.py:make_embd_proc); tests/test_oracle_golden.py checks this file against it (up
to the per-column sign of the LDA eigenvectors, which eigh leaves arbitrary).
"""
import numpy as np
import scipy.linalg as spl


def length_norm(e):
    e = e.copy()
    e /= np.sqrt((e ** 2).sum(axis=1)[:, np.newaxis])
    return e


def lda_fit(class_mats, dim, eps=1e-6):
    """class_mats: list of (n_s, D) arrays already passed through the preceding links."""
    counts, means, covs = [], [], []
    for m in class_mats:
        if m.shape[0] > 1:
            counts.append(m.shape[0])
            means.append(np.mean(m, axis=0))
            covs.append(np.cov(m, rowvar=False, bias=True))
    counts, means, covs = np.array(counts), np.vstack(means), np.array(covs)
    mean = np.sum(counts[:, np.newaxis] * means, axis=0) / np.sum(counts)
    BC = np.cov(means, rowvar=False, bias=True, fweights=counts)
    WC = np.sum(counts[:, np.newaxis, np.newaxis] * covs, axis=0) / np.sum(counts)
    E, M = spl.eigh(WC)
    floor = np.max(E) * eps
    E[E < floor] = floor
    T1 = np.dot(np.diag(1 / np.sqrt(E)), M.T)
    BC = np.dot(np.dot(T1, BC), T1.T)
    _, lda = spl.eigh(BC)
    return mean, np.dot(T1.T, lda[:, -dim:])


def chain_fit_apply(train_rows, train_spk, dim, probe):
    """mean-subtract(train) | length-norm | lda(train, utt2spk, dim) | length-norm applied to probe.
    Returns (probe output, (mean1, lda_mean, lda))."""
    mean1 = np.mean(train_rows, axis=0)
    t = length_norm(train_rows - mean1)
    mats = {}
    for row, s in zip(t, train_spk):
        mats.setdefault(s, []).append(row)
    m, lda = lda_fit([np.vstack(v) for v in mats.values()], dim)
    out = length_norm((length_norm(probe - mean1) - m).dot(lda))
    return out, (mean1, m, lda)
