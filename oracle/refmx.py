"""numpy driver for oracle/_ref/libref_<op>.so -- the reference's own operator_cxx sources compiled
against oracle/mxshim/ (see build_ref_cxx.py).  TEST INFRASTRUCTURE ONLY: may be imported by
tests/ and tests/golden/make_golden_cxx.py, never by anything under simpledet_amd/.

    op = RefOp("roi_align_v2", "_contrib_ROIAlign_v2", pooled_size=(7, 7), spatial_scale=0.25)
    out, ax, ay = op.forward([data, rois], ctx="cpu")       # shapes come from the reference's
    dx, drois = op.backward([dy], [data, rois], [out, ax, ay], ctx="gpu")   # own FInferShape

ctx="gpu" runs the reference's CUDA operator through the host emulation in mxshim/cuemu.h.
Parameters are passed as strings exactly as MXNet's Python front end does (str(value)).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")

_DTYPE_FLAG = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.uint8): 3,
               np.dtype(np.int32): 4, np.dtype(np.int64): 6}
_FLAG_DTYPE = {v: k for k, v in _DTYPE_FLAG.items()}


class _Arr(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("ndim", ctypes.c_int), ("shape", ctypes.c_int64 * 8),
                ("dtype", ctypes.c_int)]


def available(lib):
    return os.path.exists(os.path.join(_REF, "libref_%s.so" % lib))


_libs = {}


def _load(lib):
    if lib not in _libs:
        path = os.path.join(_REF, "libref_%s.so" % lib)
        if not os.path.exists(path):
            raise RuntimeError("%s missing: run `python oracle/build_ref_cxx.py` where /root/reference exists"
                               % path)
        d = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        d.mxref_last_error.restype = ctypes.c_char_p
        d.mxref_create.restype = ctypes.c_void_p
        d.mxref_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                   ctypes.POINTER(ctypes.c_char_p)]
        d.mxref_free.argtypes = [ctypes.c_void_p]
        for fn in ("mxref_num_outputs", "mxref_num_visible_outputs", "mxref_is_legacy"):
            getattr(d, fn).argtypes = [ctypes.c_void_p]
        for fn in ("mxref_list_arguments", "mxref_list_outputs"):
            getattr(d, fn).argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        d.mxref_infer_shape.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(_Arr), ctypes.c_int,
                                        ctypes.POINTER(_Arr)]
        d.mxref_infer_type.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                       ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        d.mxref_forward.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.POINTER(_Arr), ctypes.c_int, ctypes.POINTER(_Arr),
                                    ctypes.POINTER(ctypes.c_int)]
        d.mxref_backward.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int, ctypes.POINTER(_Arr)] * 4 \
            + [ctypes.POINTER(ctypes.c_int)]
        d.mxref_backward_dependency.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                                ctypes.c_int]
        d.mxref_gradient_inputs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                            ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        d.mxref_srand.argtypes = [ctypes.c_uint]
        _libs[lib] = d
    return _libs[lib]


def srand(seed):
    """seed libc's rand() -- the generator std::random_shuffle draws from in the reference."""
    ctypes.CDLL(None).srand(ctypes.c_uint(seed))


def _pystr(v):
    """what MXNet's Python front end sends for a keyword argument: str(value)."""
    if isinstance(v, (list, tuple)):
        return "(" + ", ".join(_pystr(x) for x in v) + ("," if len(v) == 1 else "") + ")"
    if isinstance(v, (bool, np.bool_)):
        return "True" if v else "False"
    if isinstance(v, (float, np.floating)):
        return repr(float(v))
    return str(v)


def _arrs(arrays):
    a = (_Arr * max(1, len(arrays)))()
    for i, x in enumerate(arrays):
        assert x.flags["C_CONTIGUOUS"]
        a[i].data = x.ctypes.data
        a[i].ndim = x.ndim
        for j, s in enumerate(x.shape):
            a[i].shape[j] = s
        a[i].dtype = _DTYPE_FLAG[x.dtype]
    return a


_REQ = {"null": 0, "write": 1, "inplace": 2, "add": 3}
_DEV = {"cpu": 1, "gpu": 2}


class RefError(RuntimeError):
    pass


class RefOp(object):
    def __init__(self, lib, op_name, **kwargs):
        self.lib = _load(lib)
        self.libname = lib
        self.name = op_name
        self.kwargs = kwargs
        keys = [k.encode() for k in kwargs]
        vals = [_pystr(v).encode() for v in kwargs.values()]
        n = len(keys)
        self.h = self.lib.mxref_create(op_name.encode(), n, (ctypes.c_char_p * max(1, n))(*keys),
                                       (ctypes.c_char_p * max(1, n))(*vals))
        if not self.h:
            raise RefError(self.lib.mxref_last_error().decode())
        self.legacy = bool(self.lib.mxref_is_legacy(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.mxref_free(self.h)
            self.h = None

    def _chk(self, rc):
        if rc < 0:
            raise RefError(self.lib.mxref_last_error().decode())
        return rc

    def _names(self, fn):
        buf = ctypes.create_string_buffer(4096)
        n = self._chk(fn(self.h, buf, 4096))
        return buf.value.decode().split(",") if n else []

    def list_arguments(self):
        return self._names(self.lib.mxref_list_arguments)

    def list_outputs(self):
        return self._names(self.lib.mxref_list_outputs)

    def num_outputs(self):
        return self._chk(self.lib.mxref_num_outputs(self.h))

    def num_visible_outputs(self):
        return self._chk(self.lib.mxref_num_visible_outputs(self.h))

    def infer_shape(self, in_shapes):
        ins = _arrs([np.empty(s, np.uint8) for s in in_shapes])
        outs = (_Arr * 16)()
        n = self._chk(self.lib.mxref_infer_shape(self.h, len(in_shapes), ins, 16, outs))
        return [tuple(outs[i].shape[j] for j in range(outs[i].ndim)) for i in range(n)]

    def infer_type(self, in_dtypes):
        it = (ctypes.c_int * len(in_dtypes))(*[_DTYPE_FLAG[np.dtype(t)] for t in in_dtypes])
        ot = (ctypes.c_int * 16)()
        n = self._chk(self.lib.mxref_infer_type(self.h, len(in_dtypes), it, 16, ot))
        return [_FLAG_DTYPE[ot[i]] for i in range(n)]

    def backward_dependency(self, nin):
        dep = (ctypes.c_int * 64)()
        n = self._chk(self.lib.mxref_backward_dependency(self.h, nin, dep, 64))
        return [dep[i] for i in range(n)]

    def gradient_inputs(self, nin):
        """(backward op name, [codes]); code 0.. = out_grad i, 100+i = input i, 200+i = output i."""
        enc = (ctypes.c_int * 64)()
        name = ctypes.create_string_buffer(256)
        n = self._chk(self.lib.mxref_gradient_inputs(self.h, nin, enc, 64, name, 256))
        return name.value.decode(), [enc[i] for i in range(n)]

    def forward(self, inputs, ctx="cpu", is_train=True, req=None, outputs=None):
        inputs = [np.ascontiguousarray(x) for x in inputs]
        if outputs is None:
            shapes = self.infer_shape([x.shape for x in inputs])
            try:
                types = self.infer_type([x.dtype for x in inputs])
            except RefError:
                types = [inputs[0].dtype] * len(shapes)
            # poison: an output the reference never writes must not look like a zero
            outputs = [np.full(s, np.nan, t) if np.dtype(t).kind == "f" else np.zeros(s, t)
                       for s, t in zip(shapes, types)]
        rq = (ctypes.c_int * len(outputs))(*[_REQ[r] for r in (req or ["write"] * len(outputs))])
        self._chk(self.lib.mxref_forward(self.h, _DEV[ctx], int(is_train), len(inputs), _arrs(inputs),
                                         len(outputs), _arrs(outputs), rq))
        return outputs

    def backward(self, out_grads, inputs, outputs, ctx="cpu", req=None, in_grads=None):
        """in_grads for `inputs`; NNVM ops are routed through their _backward_* op the way the
        reference's FGradient wires it."""
        inputs = [np.ascontiguousarray(x) for x in inputs]
        outputs = [np.ascontiguousarray(x) for x in outputs]
        out_grads = [np.ascontiguousarray(x) for x in out_grads]
        if in_grads is None:
            in_grads = [np.full(x.shape, np.nan, x.dtype) for x in inputs]
        req = req or ["write"] * len(in_grads)
        if self.legacy:
            # the legacy interface always receives one out_grad slot per output
            og = list(out_grads) + [np.zeros_like(o) for o in outputs[len(out_grads):]]
            rq = (ctypes.c_int * len(in_grads))(*[_REQ[r] for r in req])
            self._chk(self.lib.mxref_backward(self.h, _DEV[ctx], len(og), _arrs(og), len(inputs),
                                              _arrs(inputs), len(outputs), _arrs(outputs),
                                              len(in_grads), _arrs(in_grads), rq))
            return in_grads
        bname, codes = self.gradient_inputs(len(inputs))
        pool = {}
        for i, g in enumerate(out_grads):
            pool[i] = g
        for i, x in enumerate(inputs):
            pool[100 + i] = x
        for i, x in enumerate(outputs):
            pool[200 + i] = x
        bop = RefOp(self.libname, bname, **self.kwargs)
        bop.forward([pool[c] for c in codes], ctx=ctx, req=req, outputs=in_grads)
        return in_grads
