"""DenseVec: host-side mirror of the reference's ``DenseVec<T>`` (densevec.rs:5-140) and of the
``Vector`` trait's default algorithms (vector.rs:40-63), backed by a device-resident ``smh_vec``.

Same names, argument meaning and failure behaviour as the reference; every operation runs in
the HIP library (no host arithmetic).  ``T`` is f32 or f64.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


class DenseVec:
    __slots__ = ("_h", "_dtype", "_keep")

    def __init__(self, handle, dtype, keep=None):
        self._h = handle
        self._dtype = np.dtype(dtype)
        self._keep = keep  # object owning borrowed device memory (e.g. a torch tensor)

    # ---- constructors: Vector::with_capacity / from_vec (densevec.rs:24-34) -------------------
    @classmethod
    def from_vec(cls, vec, dtype=None):
        a = np.ascontiguousarray(vec, dtype=dtype if dtype is not None else None)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64 if dtype is None else dtype)
        h = C.c_void_p()
        check(lib().smh_vec_from_host(_lib.dtype_code(a.dtype), a.size, a.ctypes.data, C.byref(h)))
        return cls(h, a.dtype)

    @classmethod
    def zeros(cls, n, dtype=np.float32):
        h = C.c_void_p()
        check(lib().smh_vec_create(_lib.dtype_code(dtype), n, C.byref(h)))
        return cls(h, dtype)

    @classmethod
    def from_device_ptr(cls, ptr, n, dtype, keep=None):
        """Borrow n elements of device memory (e.g. ``tensor.data_ptr()``); ``keep`` is held alive."""
        h = C.c_void_p()
        check(lib().smh_vec_wrap_dev(_lib.dtype_code(dtype), n, C.c_void_p(ptr), C.byref(h)))
        return cls(h, dtype, keep)

    @classmethod
    def from_torch(cls, t):
        import torch
        assert t.is_cuda and t.is_contiguous() and t.dim() == 1
        dt = {torch.float32: np.float32, torch.float64: np.float64}[t.dtype]
        return cls.from_device_ptr(t.data_ptr(), t.numel(), dt, keep=t)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().smh_vec_destroy(h)
            except Exception:
                pass

    # ---- accessors ----------------------------------------------------------------------------
    @property
    def dtype(self):
        return self._dtype

    def dim(self):  # densevec.rs:36-38
        return lib().smh_vec_dim(self._h)

    def __len__(self):
        return self.dim()

    def data_ptr(self):
        return lib().smh_vec_data(self._h)

    def to_numpy(self):  # iter() (densevec.rs:20-22) collected
        out = np.empty(self.dim(), dtype=self._dtype)
        check(lib().smh_vec_download(self._h, out.ctypes.data))
        return out

    def get(self, i):  # densevec.rs:40-42: bounds-checked
        if i >= self.dim():
            raise IndexError("index out of bounds: the len is %d but the index is %d" % (self.dim(), i))
        return self.to_numpy()[i]

    def clone(self):
        out = DenseVec.zeros(self.dim(), self._dtype)
        check(lib().smh_vec_copy(out._h, self._h))
        return out

    # ---- Vector::add / sub / scale (densevec.rs:51-73) ------------------------------------------
    def add(self, rhs):
        check(lib().smh_vec_add(self._h, rhs._h))

    def sub(self, rhs):
        check(lib().smh_vec_sub(self._h, rhs._h))

    def scale(self, a):
        check(lib().smh_vec_scale(self._h, float(a)))

    def axpy(self, a, x):
        """self += round(a*x): the composite `*x += p.clone() * alpha` (linearsolver.rs:47)."""
        check(lib().smh_vec_axpy(self._h, float(a), x._h))

    def xpby(self, b, r):
        """self = round(b*self) + r: `p.scale(beta); p.add(&r)` (linearsolver.rs:58-59)."""
        check(lib().smh_vec_xpby(self._h, float(b), r._h))

    # ---- operator sugar (densevec.rs:76-140) ---------------------------------------------------
    def __iadd__(self, rhs):
        self.add(rhs)
        return self

    def __isub__(self, rhs):
        self.sub(rhs)
        return self

    def __imul__(self, a):
        self.scale(a)
        return self

    def __add__(self, rhs):
        ret = self.clone()
        ret += rhs
        return ret

    def __sub__(self, rhs):
        ret = self.clone()
        ret -= rhs
        return ret

    def __mul__(self, rhs):
        if isinstance(rhs, DenseVec):  # inner product (densevec.rs:133-140)
            return self.inner_prod(rhs)
        ret = self.clone()
        ret *= rhs
        return ret

    # ---- Vector defaults (vector.rs:50-63) -------------------------------------------------------
    def inner_prod(self, rhs):
        out = C.c_double()
        check(lib().smh_vec_dot(self._h, rhs._h, C.byref(out)))
        return self._dtype.type(out.value)

    def norm_squared(self):
        out = C.c_double()
        check(lib().smh_vec_norm_squared(self._h, C.byref(out)))
        return self._dtype.type(out.value)

    def norm(self):
        out = C.c_double()
        check(lib().smh_vec_norm(self._h, C.byref(out)))
        return out.value
