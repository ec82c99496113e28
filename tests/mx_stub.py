"""A minimal stand-in for the slice of MXNet the CustomOp adapter touches (mx.operator.CustomOp /
CustomOpProp / register, mx.sym.Custom, mx.nd.empty/array).  NDArrays wrap torch tensors so the
adapter's raw-pointer path (`data_ptr()`) works on a GPU box; on CPU only registration, parameter
parsing and shape inference are exercised.  MXNet itself is not installed in this environment."""
import types


class _NDArray:
    def __init__(self, t):
        self.t = t

    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def size(self):
        return int(self.t.numel())

    @property
    def context(self):
        return str(self.t.device)

    def data_ptr(self):
        return self.t.data_ptr()

    def wait_to_read(self):
        pass

    def __getitem__(self, i):
        return _NDArray(self.t[i])


class CustomOp:
    def __init__(self):
        pass

    def assign(self, dst, req, src):
        import torch
        if req in ("null", 0):
            return
        s = src.t if isinstance(src, _NDArray) else src
        if req in ("add", 3):
            dst.t += s
        else:
            if isinstance(s, (int, float)):
                dst.t.fill_(s)
            else:
                dst.t.copy_(s)


class CustomOpProp:
    def __init__(self, need_top_grad=True):
        self.need_top_grad_ = need_top_grad


def make_stub():
    import torch
    registry = {}

    def register(name):
        def deco(cls):
            registry[name] = cls
            return cls
        return deco

    class Symbol:
        def __init__(self, op_type, inputs, params, nout):
            self.op_type, self.inputs, self.params, self.nout = op_type, inputs, params, nout

        def __getitem__(self, i):
            return ("out", self, i)

    def Custom(*args, op_type=None, name=None, **kwargs):
        prop_cls = registry[op_type]
        params = {k: v for k, v in kwargs.items() if not isinstance(v, Symbol)}
        inputs = list(args) + [v for v in kwargs.values() if isinstance(v, Symbol)]
        return Symbol(op_type, inputs, params, len(prop_cls(**params).list_outputs()))

    def Variable(name):
        return Symbol("var:" + name, [], {}, 1)

    dev = "cuda" if torch.cuda.is_available() else "cpu"
    nd = types.SimpleNamespace(
        NDArray=_NDArray,
        empty=lambda shape, ctx=None, dtype="float32": _NDArray(
            torch.empty(shape, device=dev, dtype=getattr(torch, dtype))),
        array=lambda v, ctx=None, dtype="float32": _NDArray(
            torch.tensor(v, device=dev, dtype=getattr(torch, dtype))))
    sym = types.SimpleNamespace(Symbol=Symbol, Custom=Custom, Variable=Variable,
                                Group=lambda syms: list(syms), contrib=types.SimpleNamespace(),
                                reshape=lambda data, shape, name=None: ("reshape", data, tuple(shape)),
                                Cast=lambda data, dtype, name=None: ("cast", data, dtype))
    mx = types.SimpleNamespace(operator=types.SimpleNamespace(CustomOp=CustomOp,
                                                              CustomOpProp=CustomOpProp,
                                                              register=register),
                               nd=nd, sym=sym, registry=registry)
    return mx


def wrap(t):
    return _NDArray(t)
