"""Embedding-processing chain (mean-subtract | length-norm | lda | whitening) behind the reference's
interface, `wespeaker/utils/embedding_processing.py`:

    chain_string_to_dict            :23-67
    Lda                             :70-178   (scatter statistics + eigen-decomposition, `__call__`)
    Length_norm / Whitening / MeanSubtraction   :181-216
    EmbeddingProcessingChain        :219-271  (`__call__`, save / load (pickle), update_link)
and the tools bin/prep_embd_proc.py, bin/apply_embd_proc.py (`prep_embd_proc`, `apply_embd_proc`).

Everything that touches the N embeddings runs on the GPU: a link is applied to all rows by
`ws_rows_affine` (y = (x - sub) M, optional length normalisation, float64), and the LDA / mean
statistics (per-speaker means, within-class scatter) come from `ws_plda_stats`.  The D x D algebra
(eigh, whitening) is numpy / scipy float64 exactly like the reference.
"""
import pickle
import re

import numpy as np
import torch

from . import _lib, plda_train
from .kaldi_io import VectorWriter, read_vec_scp


def chain_string_to_dict(chain_string=None):
    """embedding_processing.py:23-67."""
    links = chain_string.split('|') if chain_string is not None else []
    a = []
    for link in links:
        x = link.split('--')
        method = x.pop(0).strip(' ')
        args_and_values = {}
        for xx in x:
            xx = re.sub("=", " ", xx)
            xx = re.sub(" +", " ", xx).strip(' ').split(' ')
            assert len(xx) == 2
            args_and_values[xx[0]] = xx[1]
        a.append([method, args_and_values])
    return a


def _apply_link(embd, sub=None, M=None, normalize=False):
    """rows (n, d_in) numpy float32/float64 -> (n, d_out) numpy float64 through ws_rows_affine."""
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    x = np.ascontiguousarray(embd)
    is64 = x.dtype == np.float64
    if not is64:
        x = x.astype(np.float32, copy=False)
    xt = torch.from_numpy(x).to(dev)
    n, d_in = int(xt.shape[0]), int(xt.shape[1])
    st = torch.from_numpy(np.ascontiguousarray(sub, dtype=np.float64)).to(dev) if sub is not None else None
    mt = torch.from_numpy(np.ascontiguousarray(M, dtype=np.float64)).to(dev) if M is not None else None
    d_out = int(mt.shape[1]) if mt is not None else d_in
    out = torch.empty((n, d_out), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().ws_rows_affine(
            _lib.ptr(xt), int(is64), n, d_in, _lib.ptr(st) if st is not None else None,
            _lib.ptr(mt) if mt is not None else None, d_out, int(bool(normalize)), _lib.ptr(out),
            _lib.current_stream_ptr(dev)), "ws_rows_affine")
        res = out.cpu().numpy()
    return res


class Lda:

    def compute_mean_and_lda_scatter_matrices(self, scp_file, utt2spk_file, equal_speaker_weight=False,
                                              current_chain=None):
        """embedding_processing.py:72-131.  Speakers with a single utterance are skipped like there;
        per-speaker means and the within-class scatter come from one ws_plda_stats call."""
        if current_chain is None:
            current_chain = lambda e: e  # noqa: E731
        _, embeddings_dict = plda_train.get_data_for_plda(scp_file, utt2spk_file)
        mats = [np.vstack(v) for v in embeddings_dict.values() if len(v) > 1]
        n_skipped = len(embeddings_dict) - len(mats)
        counts = np.array([m.shape[0] for m in mats])
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        rows = current_chain(np.vstack(mats))
        means, scatter = plda_train.gpu_stats(rows.astype(np.float32), offs)
        print("  #speakers: {}, #used {}, #skipped {} (only having one utterances)".format(
            len(embeddings_dict), len(mats), n_skipped))
        if equal_speaker_weight:
            mean = means.mean(0)
            between = np.cov(means, rowvar=False, bias=True)
            # sum of the per-speaker (biased) covariances / #speakers: needs the per-speaker split
            covs = [np.cov(rows[a:b], rowvar=False, bias=True) for a, b in zip(offs[:-1], offs[1:])]
            within = np.sum(covs, axis=0) / len(embeddings_dict)
        else:
            mean = np.sum(counts[:, None] * means, axis=0) / counts.sum()
            between = np.cov(means, rowvar=False, bias=True, fweights=counts)
            within = scatter / counts.sum()          # sum_s n_s cov_s(bias) / sum_s n_s
        return mean, between, within

    def __init__(self, args, current_chain=None):
        import scipy.linalg as spl
        print(" LDA")
        dim = int(args['dim'])
        eps = float(args['eps']) if 'eps' in args else 1e-6
        self.m, BC, WC = self.compute_mean_and_lda_scatter_matrices(args['scp'], args['utt2spk'],
                                                                    current_chain=current_chain)
        E, M = spl.eigh(WC)
        E_floor = np.max(E) * eps                    # floor like Kaldi (:147-150)
        E[E < E_floor] = E_floor
        T1 = np.dot(np.diag(1 / np.sqrt(E)), M.T)
        BC = np.dot(np.dot(T1, BC), T1.T)
        D, lda = spl.eigh(BC)
        self.lda = np.dot(T1.T, lda[:, -dim:])
        print("  Input dimension: {}, output dimension: {}, sum of all eigenvalues {:.2f}, sum of kept "
              "eigenvalues {:.2f}".format(len(D), dim, np.sum(D), np.sum(D[-dim:])))

    def __call__(self, embd):
        return _apply_link(embd, sub=self.m, M=self.lda)


class Length_norm:

    def __init__(self, args=None, current_chain=None):
        pass

    def __call__(self, embd):
        return _apply_link(embd, normalize=True)


class Whitening:
    """Declared but empty in the reference (:198-201): constructing it works, calling it does not."""

    def __init__(self, args, current_chain):
        pass


class MeanSubtraction:

    def __init__(self, args, current_chain=None):
        if current_chain is None:
            current_chain = lambda e: e  # noqa: E731
        e = np.vstack(list(read_vec_scp(args['scp']).values()))
        rows = np.asarray(current_chain(e))
        means, _ = plda_train.gpu_stats(rows.astype(np.float32), [0, rows.shape[0]])
        self.mean = means[0]

    def __call__(self, embd):
        return _apply_link(embd, sub=self.mean)


class EmbeddingProcessingChain:
    string2class = {'lda': Lda, 'length-norm': Length_norm, 'whitening': Whitening,
                    'mean-subtract': MeanSubtraction}

    def __init__(self, chain=None):
        self.chain_of_classes = []
        for m, a in chain_string_to_dict(chain):
            print("Method: {}".format(m))
            print("Argument: {}".format(a))
            self.chain_of_classes.append(self.string2class[m](a, self))

    def __call__(self, embd):
        for c in self.chain_of_classes:
            embd = c(embd)
        return embd

    def save(self, path, data_format='pickle'):
        print("Saving embedding processing chain to {}".format(path))
        with open(path, 'wb') as f:
            pickle.dump(self.chain_of_classes, f)

    def load(self, path, data_format='pickle'):
        print("Loading embedding processing chain from {}".format(path))
        with open(path, 'rb') as f:
            self.chain_of_classes = pickle.load(f)

    def update_link(self, link_no_to_replace, new_link):
        nl = chain_string_to_dict(new_link)
        assert len(nl) == 1, "Length of new chain must be one."
        m, a = nl[0]
        old, self.chain_of_classes = self.chain_of_classes, []
        for i, ol in enumerate(old):
            if i != link_no_to_replace:
                self.chain_of_classes.append(ol)
            else:
                print("Replacing link number {} ({}) with".format(i, ol))
                self.chain_of_classes.append(self.string2class[m](a, self))


# --------------------------------------------------- bin/prep_embd_proc.py, bin/apply_embd_proc.py
def prep_embd_proc(chain='whitening | length-norm ', path=None):
    c = EmbeddingProcessingChain(chain=chain)
    if path:
        c.save(path)
    return c


def apply_embd_proc(path, input, output):  # noqa: A002 (reference flag names)
    """bin/apply_embd_proc.py: --path --input (scp) --output ('x.ark,scp' | 'x.ark')."""
    chain = EmbeddingProcessingChain()
    chain.load(path)
    d = read_vec_scp(input)
    utt = list(d.keys())
    embd = chain(np.array(list(d.values())))
    print("Read {} embeddings of dimension {}.".format(len(utt), embd.shape[1] if len(utt) else 0))
    if output.endswith('ark,scp') or output.endswith('scp,ark'):
        stem = output[:-len('ark,scp')]
        with VectorWriter(stem + "ark", stem + "scp") as w:
            for u, e in zip(utt, embd):
                w(u, e)
    elif output.endswith('ark'):
        with VectorWriter(output) as w:
            for u, e in zip(utt, embd):
                w(u, e)
    else:
        raise ValueError("output must end in 'ark,scp', 'scp,ark' or 'ark'")
    return embd
