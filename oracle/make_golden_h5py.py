#!/opt/conda/bin/python3.9
"""TEST INFRASTRUCTURE (generator).  A PLDA model file written by the real h5py, exactly the way the reference's
TwoCovPLDA.save_model writes it (wespeaker/utils/plda/two_cov_plda.py:311-339: six create_dataset calls, the four
arrays with maxshape=None-per-axis, compression="gzip", fletcher32=True, the two flags as Python ints).

The product's interpreter has no h5py; the Anaconda interpreter of this image has (h5py 3.3.0), so this script is
run with THAT interpreter as a generator only:

    /opt/conda/bin/python3.9 oracle/make_golden_h5py.py          # writes tests/golden/plda_h5py.h5 + _expected.npz
    /opt/conda/bin/python3.9 oracle/make_golden_h5py.py read F    # prints what h5py reads from file F as JSON

(the `read` form lets tests/test_host_logic.py check that h5py reads what wespeaker_amd/hdf5_io.py wrote).  The
reference module itself cannot be imported there (it needs kaldiio), hence the six calls are repeated here.
"""
import json
import os
import sys

import h5py
import numpy as np

KEYS = ("mu", "transform", "psi", "offset", "normalize_length", "subtract_train_set_mean")


def write_like_the_reference(path, mu, transform, psi, offset, normalize_length, subtract_train_set_mean):
    with h5py.File(path, "w") as f:                                   # two_cov_plda.py:313-339
        f.create_dataset("mu", data=mu, maxshape=(None), compression="gzip", fletcher32=True)
        f.create_dataset("transform", data=transform, maxshape=(None, None), compression="gzip", fletcher32=True)
        f.create_dataset("psi", data=psi, maxshape=(None), compression="gzip", fletcher32=True)
        f.create_dataset("offset", data=offset, maxshape=(None), compression="gzip", fletcher32=True)
        f.create_dataset("normalize_length", data=int(normalize_length), maxshape=(None))
        f.create_dataset("subtract_train_set_mean", data=int(subtract_train_set_mean), maxshape=(None))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "read":
        with h5py.File(sys.argv[2], "r") as f:                        # two_cov_plda.py:348-355
            out = {k: np.asarray(f.get(k)[()]).tolist() for k in KEYS}
            out["_dtypes"] = {k: str(f.get(k).dtype) for k in KEYS}
            out["_shapes"] = {k: list(f.get(k).shape) for k in KEYS}
        print(json.dumps(out))
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.RandomState(20260924)
    D = 12
    mu, transform = rng.randn(D), rng.randn(D, D)
    psi = rng.rand(D) * 3 + 0.1
    offset = -1.0 * np.matmul(transform, mu)
    gold = os.path.join(root, "tests", "golden")
    write_like_the_reference(os.path.join(gold, "plda_h5py.h5"), mu, transform, psi, offset, True, False)
    np.savez(os.path.join(gold, "plda_h5py_expected.npz"), mu=mu, transform=transform, psi=psi, offset=offset,
             normalize_length=1, subtract_train_set_mean=0)
    print("wrote plda_h5py.h5 with h5py", h5py.__version__)


if __name__ == "__main__":
    main()
