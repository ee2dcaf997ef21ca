"""Minimal h5py stand-in (TEST INFRASTRUCTURE, see tests/refscripts/README.md): one .npz archive per "HDF5 file".

Covers what the reference's scripts and readers call: ``File(path, mode)`` as a context manager or kept open,
``create_dataset(name, shape, maxshape=, dtype=, compression=, data=)``, ``Dataset.resize(n, axis=0)``, ``.shape``,
``len()``, basic / negative-slice indexing for reads and writes, ``np.array(dataset)``, group-style names
(``'/value/img'``) and object references (a reference is the NAME of its target, stored in a unicode array, so
``f[f['/value/img'][i, 0]]`` dereferences as h5py's does for MATLAB v7.3 files)."""
import os

import numpy as np

__version__ = "0.0-stub"


def _key(name):
    if isinstance(name, bytes):
        name = name.decode()
    return str(name).lstrip("/")


class Dataset:
    def __init__(self, owner, name, arr):
        self._owner, self.name, self._a = owner, "/" + name, arr

    shape = property(lambda self: self._a.shape)
    dtype = property(lambda self: self._a.dtype)
    ndim = property(lambda self: self._a.ndim)

    def __len__(self):
        return self._a.shape[0]

    def __getitem__(self, idx):
        out = self._a[idx]
        if isinstance(out, np.ndarray):
            return out.copy()
        return out.item() if self._a.dtype.kind == "U" else out

    def __setitem__(self, idx, value):
        self._owner._writable()
        self._a[idx] = value
        self._owner._dirty = True

    def __array__(self, dtype=None, copy=None):
        return self._a.astype(dtype) if dtype is not None else self._a.copy()

    def resize(self, size, axis=None):
        self._owner._writable()
        shape = list(self._a.shape)
        if axis is None:
            shape = list(size)
        else:
            shape[axis] = int(size)
        new = np.zeros(shape, dtype=self._a.dtype)
        sl = tuple(slice(0, min(a, b)) for a, b in zip(self._a.shape, shape))
        new[sl] = self._a[sl]
        self._a = new
        self._owner._dirty = True


class File:
    def __init__(self, name, mode="r", **kwargs):
        self.filename, self.mode = os.fspath(name), mode
        self._sets, self._dirty, self._open = {}, False, True
        exists = os.path.exists(self.filename)
        if mode in ("r", "r+") and not exists:
            raise OSError(f"Unable to open file (stub h5py): {self.filename} does not exist")
        if mode in ("w-", "x") and exists:
            raise OSError(f"Unable to create file (stub h5py): {self.filename} exists")
        if exists and mode in ("r", "r+", "a"):
            with np.load(self.filename, allow_pickle=False) as z:
                for k in z.files:
                    self._sets[k] = Dataset(self, k, z[k])
        if mode in ("w", "w-", "x") or (mode == "a" and not exists):
            self._dirty = True

    def _writable(self):
        if self.mode == "r":
            raise OSError("stub h5py: file is open read-only")

    def create_dataset(self, name, shape=None, dtype=None, data=None, maxshape=None, compression=None, **kwargs):
        self._writable()
        k = _key(name)
        if k in self._sets:
            raise ValueError(f"Unable to create dataset (name already exists): {name}")
        if data is not None:
            arr = np.array(data, dtype=dtype)
            if shape is not None:
                arr = arr.reshape(shape)
        else:
            arr = np.zeros(shape if shape is not None else (), dtype=dtype or np.float32)
        self._sets[k] = Dataset(self, k, arr)
        self._dirty = True
        return self._sets[k]

    def __getitem__(self, name):
        k = _key(name)
        if k in self._sets:
            return self._sets[k]
        prefix = k + "/"
        if any(n.startswith(prefix) for n in self._sets):
            return _Group(self, prefix)
        raise KeyError(f"Unable to open object (object '{name}' doesn't exist)")

    def __contains__(self, name):
        k = _key(name)
        return k in self._sets or any(n.startswith(k + "/") for n in self._sets)

    def keys(self):
        return sorted({n.split("/")[0] for n in self._sets})

    def flush(self):
        if self._dirty and self.mode != "r":
            tmp = self.filename + ".tmp.npz"
            np.savez(tmp, **{k: d._a for k, d in self._sets.items()})
            os.replace(tmp, self.filename)
            self._dirty = False

    def close(self):
        if self._open:
            self.flush()
            self._open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Group:
    def __init__(self, f, prefix):
        self._f, self._p = f, prefix

    def __getitem__(self, name):
        return self._f[self._p + _key(name)]

    def keys(self):
        return sorted({n[len(self._p):].split("/")[0] for n in self._f._sets if n.startswith(self._p)})
