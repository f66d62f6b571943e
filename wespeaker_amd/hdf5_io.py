"""HDF5 datasets through the HDF5 C library itself (ctypes), for the PLDA model files.

The reference stores a trained PLDA with h5py (wespeaker/utils/plda/two_cov_plda.py:311-339: six datasets in the
root group -- `mu`, `transform`, `psi`, `offset` as float64 arrays with gzip + fletcher32 and unlimited maxshape,
`normalize_length` / `subtract_train_set_mean` as integer scalars) and reads it back with `f.get(name)[()]`
(:348-355).  h5py is not part of this image, libhdf5 is: this module binds the dozen C entry points needed to read
and write exactly that kind of file -- any layout, filter or integer / float width the library can convert --
so `TwoCovPLDA.save_model / load_model` exchange files with the reference tools.  (h5py, when importable, is not
needed; the `.npz` container and Kaldi `<Plda>` files stay available, wespeaker_amd/plda.py.)

Only what PLDA files need: numeric scalar / n-D datasets directly under the root group.
"""
import ctypes
import ctypes.util
import os

import numpy as np

_hid = ctypes.c_int64          # hid_t since HDF5 1.10
_hsize = ctypes.c_uint64
_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5P_DEFAULT, _H5S_ALL = 0, 0
_H5S_UNLIMITED = 0xFFFFFFFFFFFFFFFF
_H5T_INTEGER, _H5T_FLOAT = 0, 1

_LIB = None


class Hdf5Error(RuntimeError):
    pass


def _candidates():
    env = os.environ.get("WS_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for d in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib/x86_64-linux-gnu/hdf5/serial", "/usr/lib64",
              "/usr/local/lib"):
        for n in ("libhdf5.so", "libhdf5_serial.so"):
            yield os.path.join(d, n)


def available():
    try:
        _lib()
        return True
    except Hdf5Error:
        return False


def _lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    last = None
    for path in _candidates():
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:
            last = e
            continue
        sig = {
            "H5open": (ctypes.c_int, []),
            "H5Fopen": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid]),
            "H5Fcreate": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid, _hid]),
            "H5Fclose": (ctypes.c_int, [_hid]),
            "H5Dopen2": (_hid, [_hid, ctypes.c_char_p, _hid]),
            "H5Dcreate2": (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid, _hid]),
            "H5Dget_space": (_hid, [_hid]),
            "H5Dget_type": (_hid, [_hid]),
            "H5Dread": (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
            "H5Dwrite": (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
            "H5Dclose": (ctypes.c_int, [_hid]),
            "H5Sget_simple_extent_ndims": (ctypes.c_int, [_hid]),
            "H5Sget_simple_extent_dims": (ctypes.c_int, [_hid, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
            "H5Screate_simple": (_hid, [ctypes.c_int, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
            "H5Screate": (_hid, [ctypes.c_int]),
            "H5Sclose": (ctypes.c_int, [_hid]),
            "H5Tget_class": (ctypes.c_int, [_hid]),
            "H5Tclose": (ctypes.c_int, [_hid]),
            "H5Pcreate": (_hid, [_hid]),
            "H5Pset_chunk": (ctypes.c_int, [_hid, ctypes.c_int, ctypes.POINTER(_hsize)]),
            "H5Pset_deflate": (ctypes.c_int, [_hid, ctypes.c_uint]),
            "H5Pset_fletcher32": (ctypes.c_int, [_hid]),
            "H5Pclose": (ctypes.c_int, [_hid]),
            "H5Lexists": (ctypes.c_int, [_hid, ctypes.c_char_p, _hid]),
            "H5Eset_auto2": (ctypes.c_int, [_hid, ctypes.c_void_p, ctypes.c_void_p]),
        }
        try:
            for name, (res, args) in sig.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            if lib.H5open() < 0:
                raise Hdf5Error("H5open failed")
            ver = (ctypes.c_uint * 3)()
            lib.H5get_libversion.restype = ctypes.c_int
            lib.H5get_libversion(ctypes.byref(ver, 0), ctypes.byref(ver, 4), ctypes.byref(ver, 8))
            if (ver[0], ver[1]) < (1, 10):                # hid_t is a 32-bit int before 1.10: not this binding
                raise ValueError("HDF5 %d.%d.%d is older than 1.10" % (ver[0], ver[1], ver[2]))
            lib.H5Eset_auto2(0, None, None)           # errors are reported by return values, not printed
            lib._f64 = _hid.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value
            lib._i64 = _hid.in_dll(lib, "H5T_NATIVE_INT64_g").value
            lib._dcpl = _hid.in_dll(lib, "H5P_CLS_DATASET_CREATE_ID_g").value
        except (AttributeError, ValueError, Hdf5Error) as e:
            last = e
            continue
        _LIB = lib
        return lib
    raise Hdf5Error("no usable HDF5 C library found (set WS_HDF5_LIB to libhdf5.so): %s" % (last,))


def _check(status, what):
    if status < 0:
        raise Hdf5Error("HDF5: %s failed" % what)
    return status


def read_datasets(path, names):
    """{name: numpy array} for the named datasets of the root group: float classes come back as float64, integer
    classes as int64 (the library converts whatever width / byte order / layout / filter the file uses); scalar
    datasets as 0-d arrays, like h5py's `dset[()]`."""
    L = _lib()
    f = L.H5Fopen(os.fsencode(path), _H5F_ACC_RDONLY, _H5P_DEFAULT)
    if f < 0:
        raise Hdf5Error("cannot open %s as HDF5" % path)
    out = {}
    try:
        for name in names:
            key = name.encode()
            if L.H5Lexists(f, key, _H5P_DEFAULT) <= 0:
                raise KeyError("%s has no dataset '%s'" % (path, name))
            d = _check(L.H5Dopen2(f, key, _H5P_DEFAULT), "open dataset '%s'" % name)
            try:
                sp = _check(L.H5Dget_space(d), "dataspace of '%s'" % name)
                ty = _check(L.H5Dget_type(d), "datatype of '%s'" % name)
                try:
                    nd = _check(L.H5Sget_simple_extent_ndims(sp), "rank of '%s'" % name)
                    dims = (_hsize * max(nd, 1))()
                    if nd:
                        _check(L.H5Sget_simple_extent_dims(sp, dims, None), "extent of '%s'" % name)
                    shape = tuple(int(dims[i]) for i in range(nd))
                    cls = L.H5Tget_class(ty)
                    if cls == _H5T_FLOAT:
                        arr, mem = np.empty(shape, np.float64), L._f64
                    elif cls == _H5T_INTEGER:
                        arr, mem = np.empty(shape, np.int64), L._i64
                    else:
                        raise Hdf5Error("dataset '%s' of %s is neither integer nor float" % (name, path))
                    if arr.size:
                        _check(L.H5Dread(d, mem, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, arr.ctypes.data_as(ctypes.c_void_p)),
                               "read '%s'" % name)
                    out[name] = arr
                finally:
                    L.H5Tclose(ty)
                    L.H5Sclose(sp)
            finally:
                L.H5Dclose(d)
    finally:
        L.H5Fclose(f)
    return out


def write_datasets(path, items, compress=True):
    """Writes {name: value} into a NEW file `path`.  Arrays become float64 (floating input) or int64 datasets, one
    chunk = the whole array, gzip(4) then fletcher32 when `compress`; 1-D arrays with fixed extent, n-D arrays with
    unlimited maxshape; Python / numpy scalars become contiguous scalar datasets.  This is the layout h5py 3.3
    produces for the reference's calls (tests/golden/plda_h5py.h5, oracle/make_golden_h5py.py): the reference passes
    `maxshape=(None)` -- a plain None, not a tuple -- for its vectors and `(None, None)` for the matrix."""
    L = _lib()
    f = L.H5Fcreate(os.fsencode(path), _H5F_ACC_TRUNC, _H5P_DEFAULT, _H5P_DEFAULT)
    if f < 0:
        raise Hdf5Error("cannot create %s" % path)
    try:
        for name, value in items.items():
            a = np.asarray(value)
            if a.dtype.kind == "f":
                a, mem = np.asarray(a, dtype=np.float64, order="C"), L._f64
            elif a.dtype.kind in "iub":
                a, mem = np.asarray(a, dtype=np.int64, order="C"), L._i64
            else:
                raise Hdf5Error("'%s': only integer / float data" % name)
            plist = _H5P_DEFAULT
            if a.ndim == 0:
                sp = _check(L.H5Screate(0), "scalar dataspace")              # H5S_SCALAR
            else:
                dims = (_hsize * a.ndim)(*a.shape)
                maxd = (_hsize * a.ndim)(*([_H5S_UNLIMITED] * a.ndim)) if a.ndim > 1 else None
                sp = _check(L.H5Screate_simple(a.ndim, dims, maxd), "dataspace of '%s'" % name)
                plist = _check(L.H5Pcreate(L._dcpl), "dataset creation properties")
                chunk = (_hsize * a.ndim)(*[max(1, s) for s in a.shape])     # (filters / unlimited dims need chunks)
                _check(L.H5Pset_chunk(plist, a.ndim, chunk), "chunk shape of '%s'" % name)
                if compress and a.size:
                    _check(L.H5Pset_deflate(plist, 4), "gzip")
                    _check(L.H5Pset_fletcher32(plist), "fletcher32")
            try:
                d = _check(L.H5Dcreate2(f, name.encode(), mem, sp, _H5P_DEFAULT, plist, _H5P_DEFAULT),
                           "create dataset '%s'" % name)
                try:
                    if a.size:
                        _check(L.H5Dwrite(d, mem, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT,
                                          a.ctypes.data_as(ctypes.c_void_p)), "write '%s'" % name)
                finally:
                    L.H5Dclose(d)
            finally:
                L.H5Sclose(sp)
                if plist != _H5P_DEFAULT:
                    L.H5Pclose(plist)
    finally:
        if L.H5Fclose(f) < 0:
            raise Hdf5Error("closing %s failed" % path)
