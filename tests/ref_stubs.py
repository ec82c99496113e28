"""TEST INFRASTRUCTURE: graph-recording stand-ins for `mxnet` and `mxnext`, rich enough to import the
reference's builder modules UNMODIFIED from /root/reference and let them build their symbols
(tests/test_reference_builders.py; BASELINE configs[0] "plumbing").

* `mxnet`: every `mx.sym.<op>(...)` / `mx.symbol.<op>(...)` / `mx.sym.contrib.<op>(...)` call returns a
  `Symbol` node that records op name, symbol inputs, keyword parameters and the node name -- nothing
  is computed.  `mx.operator.{CustomOp, CustomOpProp, register}` and `mx.sym.Custom` behave like
  MXNet's (the prop is instantiated with str() kwargs to learn the output count), so the reference's
  own CustomOps (models/FPN/assign_layer_fpn.py, get_top_proposal.py, operator_py/*) register too.
* `mxnext` (github.com/RogerChern/mxnext, NOT vendored in the reference tree -- SURVEY 8(c)): the
  wrappers the hot path goes through, written from their call sites in the reference and SURVEY's
  inferred contracts.  They look `mx.sym.*` up AT CALL TIME, the way a thin wrapper library does, so
  that the plugin's aliases take effect:
      X.roi_align(feat, rois, out_size, stride)   -> mx.sym.contrib.ROIAlign_v2(pooled_size, spatial_scale)
      X.proposal_target(**kw)                      -> mx.sym.ProposalTarget(**kw)
      X.proposal(...)                              -> mx.sym.contrib.Proposal_v3 / Proposal
      X.decode_bbox(...)                           -> mx.sym.contrib.DecodeBBox
      mxnext.tvm.fpn_roi_assign / get_top_proposal -> the in-tree CustomOps they are twins of
        (models/FPN/assign_layer_fpn.py, models/FPN/get_top_proposal.py)
"""
import sys
import types


class Symbol:
    _count = 0

    def __init__(self, op, inputs=(), params=None, name=None, nout=1, parent=None, index=None):
        self.op_type, self.inputs, self.params = op, list(inputs), dict(params or {})
        self.name, self.nout, self.parent, self.index = name, nout, parent, index
        Symbol._count += 1

    # multi-output symbols index / unpack into their outputs
    def __getitem__(self, i):
        if self.op_type == "Group" and isinstance(i, int):
            return self.inputs[i]      # a Group unpacks into its members, not into selectors
        if isinstance(i, int):
            if i >= self.nout:
                raise IndexError(i)
            return Symbol("_output", [self], {"index": i}, None, 1, parent=self, index=i)
        return Symbol("_slice", [self], {"key": repr(i)})

    def __iter__(self):
        return iter([self[i] for i in range(self.nout)])

    def __len__(self):
        return self.nout

    def _bin(self, other, op):
        ins = [self] + ([other] if isinstance(other, Symbol) else [])
        return Symbol(op, ins, {} if isinstance(other, Symbol) else {"scalar": other})

    __add__ = __radd__ = lambda s, o: s._bin(o, "_plus")
    __sub__ = lambda s, o: s._bin(o, "_minus")
    __rsub__ = lambda s, o: s._bin(o, "_rminus")
    __mul__ = __rmul__ = lambda s, o: s._bin(o, "_mul")
    __truediv__ = lambda s, o: s._bin(o, "_div")
    __rtruediv__ = lambda s, o: s._bin(o, "_rdiv")
    __neg__ = lambda s: Symbol("_neg", [s])
    __pow__ = lambda s, o: s._bin(o, "_power")
    __rpow__ = lambda s, o: s._bin(o, "_rpower")
    __mod__ = lambda s, o: s._bin(o, "_mod")
    __abs__ = lambda s: Symbol("abs", [s])
    __le__ = lambda s, o: s._bin(o, "_lesser_equal")
    __gt__ = lambda s, o: s._bin(o, "_greater")
    __ge__ = lambda s, o: s._bin(o, "_greater_equal")
    __lt__ = lambda s, o: s._bin(o, "_lesser")
    __hash__ = object.__hash__

    def __eq__(self, o):     # `sym == 4` builds a node (models/TSD/poolings.py:39); symbol-vs-symbol stays identity,
        if isinstance(o, (int, float)) and not isinstance(o, bool):     # which is what the tests' list compares need
            return self._bin(o, "_equal_scalar")
        return self is o

    def __ne__(self, o):
        if isinstance(o, (int, float)) and not isinstance(o, bool):
            return self._bin(o, "_not_equal_scalar")
        return self is not o

    def __getattr__(self, k):          # sym.astype(...), sym.reshape(...), sym.get_internals() ...
        if k.startswith("__"):
            raise AttributeError(k)
        return lambda *a, **kw: _make("method_" + k, (self,) + a, kw)

    def __repr__(self):
        return "<Symbol %s %s>" % (self.op_type, self.name or "")


def source(sym):
    """the node behind an output selector"""
    return sym.parent if sym.op_type == "_output" else sym


def walk(sym, seen=None):
    """every node reachable from sym (inputs first)"""
    seen = seen if seen is not None else {}
    if id(sym) in seen:
        return seen
    seen[id(sym)] = sym
    for i in sym.inputs:
        walk(i, seen)
    return seen


def find(sym, op_type):
    roots = sym if isinstance(sym, (list, tuple)) else [sym]
    seen = {}
    for r in roots:
        walk(r, seen)
    return [n for n in seen.values() if n.op_type == op_type]


def _flatten_syms(args):
    out = []
    for a in args:
        if isinstance(a, Symbol):
            out.append(a)
        elif isinstance(a, (list, tuple)):
            out.extend(_flatten_syms(a))
        elif isinstance(a, dict):
            out.extend(_flatten_syms(list(a.values())))
    return out


def _single_output_inputs(op, ins):
    """MXNet refuses a multi-output symbol as ONE operator argument ("Keyword Argument rois is a tuple,
    single value is required" / composition error): so does the stand-in (ADVICE r4 -- a two-output
    get_top_proposal passed as `rois=` went unnoticed)."""
    for i in ins:
        if i.nout != 1:
            raise TypeError("%s: argument symbol %r has %d outputs, a single-output symbol is required"
                            % (op, i, i.nout))


def _make(op, args, kwargs):
    kwargs = dict(kwargs)
    name = kwargs.pop("name", None)
    ins = _flatten_syms(args) + _flatten_syms([v for v in kwargs.values()])
    if op not in ("Group", "method_get_internals", "method_get_children", "method_list_outputs"):
        _single_output_inputs(op, ins)
    params = {k: v for k, v in kwargs.items() if not _flatten_syms([v])}
    params.update({"_arg%d" % i: a for i, a in enumerate(args) if not _flatten_syms([a])})
    return Symbol(op, ins, params, name, _native_nout(op, params))


def _truthy(v):
    return str(v).strip().lower() in ("true", "1")


def _native_nout(op, params):
    """visible outputs of the NATIVE operators whose results the reference's builders unpack
    (NumVisibleOutputs of operator_cxx/*-inl.h; SliceChannel / split from upstream MXNet)"""
    if op in ("SliceChannel", "split"):
        return int(params.get("num_outputs", 1))
    if op in ("Proposal", "Proposal_v2", "Proposal_v3", "MultiProposal"):     # proposal_v3-inl.h:254-260
        return 2 if _truthy(params.get("output_score", False)) else 1
    if op == "GenProposalRetina":                                            # generate_proposal_retina-inl.h:161-171
        return 2
    if op in ("ProposalTarget", "ProposalTarget_v2"):                        # proposal_target-inl.h: 4 visible
        return 4
    if op == "ProposalMaskTarget":                                           # proposal_mask_target-inl.h:387-396
        return 5 + int(_truthy(params.get("output_iou", False))) + int(_truthy(params.get("output_ratio", False)))
    return 1


class _OpNamespace(types.ModuleType):
    """mx.sym / mx.symbol / mx.sym.contrib: unknown attributes are generic op constructors"""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return lambda *a, **kw: _make(k, a, kw)


class CustomOp:
    def __init__(self):
        pass

    def assign(self, dst, req, src):
        raise RuntimeError("graph-recording stub: nothing executes")


class CustomOpProp:
    def __init__(self, need_top_grad=True):
        self.need_top_grad_ = need_top_grad

    def list_arguments(self):
        return ["data"]

    def list_outputs(self):
        return ["output"]

    def list_auxiliary_states(self):
        return []


CONTRIB_OPS = ("Axpy", "BBoxNorm", "BroadcastScale", "DecodeBBox", "FocalLoss", "GAP", "GenAnchor", "GenProposal",
               "GenProposalRetina", "GroupNorm", "NMS", "Proposal", "Proposal_v2", "Proposal_v3",
               "Quantization_int8", "ROIAlign_v2", "SigmoidCrossEntropy", "SyncBatchNorm", "SyncInplaceABN",
               "DeformableConvolution", "DeformablePSROIPooling", "ROIAlign", "MultiProposal", "box_nms")


class _Prefix:
    """mx.name.Prefix: a context manager; node names are not what these tests look at"""

    def __init__(self, prefix):
        self.prefix = prefix

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def make_mx():
    mx = types.ModuleType("mxnet")
    registry = {}

    def register(name):
        def deco(cls):
            registry[name] = cls
            return cls
        return deco

    def Custom(*args, op_type=None, name=None, **kwargs):
        prop_cls = registry[op_type]
        if args and any(isinstance(v, Symbol) for v in kwargs.values()):
            # mxnet/symbol/symbol.py, Symbol._compose: a variadic operator takes its input Symbols
            raise TypeError("compose only accept input Symbols either as positional or keyword arguments, not both")
        ins = _flatten_syms(args) + [v for v in kwargs.values() if isinstance(v, Symbol)]
        _single_output_inputs(op_type, ins)
        params = {k: v for k, v in kwargs.items() if not isinstance(v, Symbol)}
        prop = prop_cls(**{k: str(v) for k, v in params.items()})   # MXNet hands CustomOpProp strings
        return Symbol(op_type, ins, params, name, len(prop.list_outputs()))

    sym = _OpNamespace("mxnet.symbol")
    sym.Symbol = Symbol
    sym.Custom = Custom
    sym.Variable = sym.var = lambda name, **kw: Symbol("var", [], dict(kw), name)
    sym.Group = lambda syms: Symbol("Group", list(syms), {}, None, len(syms))
    sym.contrib = _OpNamespace("mxnet.symbol.contrib")
    # what a SimpleDet build of MXNet registers under contrib (operator_cxx/contrib/*.cc: NNVM_REGISTER_OP /
    # MXNET_REGISTER_OP_PROPERTY names without the _contrib_ prefix) + the upstream ops the builders use;
    # models/retinanet/builder.py:358 tests membership
    sym.contrib.__all__ = list(CONTRIB_OPS)
    # `mx.contrib.symbol` / `mx.contrib.sym`: in MXNet a SEPARATE module that star-imports the contrib
    # namespace at import time (mxnet/contrib/symbol.py) -- models/tridentnet/resnet_v1.py:85 builds its
    # DeformableConvolution through it
    csym = _OpNamespace("mxnet.contrib.symbol")
    csym.__all__ = list(CONTRIB_OPS)
    mx.contrib = types.SimpleNamespace(symbol=csym, sym=csym)

    mx.sym = mx.symbol = sym
    mx.operator = types.SimpleNamespace(CustomOp=CustomOp, CustomOpProp=CustomOpProp, register=register)
    mx.nd = _OpNamespace("mxnet.ndarray")
    mx.ndarray = mx.nd
    mx.registry = registry
    mx.init = _OpNamespace("mxnet.init")
    mx.initializer = mx.init
    mx.context = types.SimpleNamespace(Context=object)
    mx.cpu = lambda i=0: "cpu(%d)" % i
    mx.gpu = lambda i=0: "gpu(%d)" % i
    mx.__version__ = "1.6.0-graph-stub"
    mx.base = types.SimpleNamespace(MXNetError=RuntimeError)
    mx.attribute = types.SimpleNamespace(AttrScope=_AttrScope)
    mx.AttrScope = _AttrScope
    mx.io = types.SimpleNamespace(DataIter=object, DataBatch=object, DataDesc=object)
    mx.metric = types.SimpleNamespace(EvalMetric=_EvalMetric)
    mx.name = types.SimpleNamespace(Prefix=_Prefix, NameManager=_Prefix)
    mx.lr_scheduler = types.SimpleNamespace(LRScheduler=object)
    return mx


class _EvalMetric:
    def __init__(self, name=None, output_names=None, label_names=None, **kw):
        self.name, self.output_names, self.label_names = name, output_names, label_names


class _AttrScope:
    def __init__(self, **kw):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def make_mxnext(mx, late_binding=True):
    """the slice of mxnext the hot path's builders touch (see the module docstring).
    late_binding=False: the wrappers use the constructors mx.sym had when mxnext was "imported"
    (a frozen copy) -- aliases installed on mx.sym afterwards do not reach them."""
    X = types.ModuleType("mxnext")
    if late_binding:
        S = lambda: mx.sym   # looked up at call time
    else:
        frozen = _OpNamespace("mxnet.symbol(frozen)")
        frozen.contrib = _OpNamespace("mxnet.symbol.contrib(frozen)")
        frozen.var, frozen.Group, frozen.Custom = mx.sym.var, mx.sym.Group, mx.sym.Custom
        S = lambda: frozen
    sys_mods_extra = {}

    X.var = lambda name, **kw: S().var(name, **kw)
    X.conv = lambda data, name=None, filter=None, kernel=1, stride=1, pad=None, dilate=1, **kw: S().Convolution(
        data=data, name=name, num_filter=filter, kernel=(kernel, kernel), stride=(stride, stride),
        dilate=(dilate, dilate), **kw)
    X.relu = lambda data, name=None: S().Activation(data=data, act_type="relu", name=name)
    X.add = lambda a, b, name=None: S().elemwise_add(a, b, name=name)
    X.fc = lambda data, name=None, filter=None, **kw: S().FullyConnected(data=data, num_hidden=filter, name=name, **kw)
    X.reshape = lambda data, shape=None, name=None, **kw: S().reshape(data=data, shape=shape, name=name)
    X.concat = lambda syms, axis=1, name=None: S().concat(*syms, dim=axis, name=name)
    X.add_n = lambda *syms, **kw: S().add_n(*syms, **kw)
    X.to_fp32 = lambda data, name=None: S().Cast(data=data, dtype="float32", name=name)
    X.to_fp16 = lambda data, name=None: S().Cast(data=data, dtype="float16", name=name)
    X.stop_grad = lambda data, name=None: S().BlockGrad(data, name=name)
    X.smooth_l1 = lambda data, scalar=1.0, name=None: S().smooth_l1(data=data, scalar=scalar, name=name)
    X.loss = lambda data, grad_scale=1.0, name=None: S().MakeLoss(data, grad_scale=grad_scale, name=name)
    X.softmax_output = lambda **kw: S().SoftmaxOutput(**kw)
    X.softmax = lambda data, axis=-1, name=None: S().softmax(data=data, axis=axis, name=name)
    X.block_grad = X.stop_grad
    X.group = lambda syms, **kw: S().Group(list(syms))

    def roi_align(feat, rois, out_size, stride, name=None):
        # the only RoIAlign in the tree whose contract matches the call (3-D rois, 5-D output)
        return S().contrib.ROIAlign_v2(data=feat, rois=rois, pooled_size=(out_size, out_size),
                                       spatial_scale=1.0 / stride, name=name)
    X.roi_align = roi_align
    X.proposal_target = lambda **kw: S().ProposalTarget(**kw)

    def proposal(cls_prob, bbox_pred, im_info, name=None, **kw):
        return S().contrib.Proposal_v3(cls_prob=cls_prob, bbox_pred=bbox_pred, im_info=im_info, name=name, **kw)
    X.proposal = proposal
    X.decode_bbox = lambda **kw: S().contrib.DecodeBBox(**kw)

    def __getattr__(k):   # every other wrapper: a generic node named after it
        if k.startswith("__"):
            raise AttributeError(k)
        return lambda *a, **kw: _make("X." + k, a, kw)
    X.__getattr__ = __getattr__

    tvm = types.ModuleType("mxnext.tvm")
    m_assign = types.ModuleType("mxnext.tvm.fpn_roi_assign")

    def fpn_roi_assign(F, rois, rcnn_stride, roi_canonical_scale, roi_canonical_level, name="fpn_roi_assign"):
        # twin of models/FPN/assign_layer_fpn.py (the in-tree CustomOp the class docstring names)
        return F.Custom(rois=rois, op_type="assign_layer_fpn", rcnn_stride=tuple(rcnn_stride),
                        roi_canonical_scale=roi_canonical_scale, roi_canonical_level=roi_canonical_level,
                        name=name)
    m_assign.fpn_roi_assign = fpn_roi_assign
    m_top = types.ModuleType("mxnext.tvm.get_top_proposal")

    def get_top_proposal(F, bbox, score, top_n, batch_size=None, name="get_top_proposal"):
        return F.Custom(bbox=bbox, score=score, op_type="get_top_proposal", top_n=top_n, name=name)
    m_top.get_top_proposal = get_top_proposal
    tvm.fpn_roi_assign, tvm.get_top_proposal = m_assign, m_top
    X.tvm = tvm

    X.__path__ = []   # a package: `from mxnext.complicate import ...`
    complicate = types.ModuleType("mxnext.complicate")

    def normalizer_factory(type="local", **kw):
        def fix_bn(data=None, name=None, **k):
            extra = {a: v for a, v in k.items() if isinstance(v, Symbol)}   # shared gamma / beta / moving stats
            return S().BatchNorm(data=data, name=name, use_global_stats=True, **extra)
        fix_bn.__name__ = {"fixbn": "fix_bn", "syncbn": "sync_bn", "localbn": "local_bn", "gn": "gn"}.get(type, str(type))
        return fix_bn
    complicate.normalizer_factory = normalizer_factory
    X.normalizer_factory = normalizer_factory
    X.complicate = complicate
    m_prop = types.ModuleType("mxnext.tvm.proposal")
    m_prop.proposal = lambda **kw: Symbol("mxnext.tvm.proposal", _flatten_syms(list(kw.values())),
                                          {k: v for k, v in kw.items() if not _flatten_syms([v])}, kw.get("name"), 2)
    tvm.proposal = m_prop
    tvm.__path__ = []
    backbone = types.ModuleType("mxnext.backbone")
    backbone.__path__ = []
    depth_config = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3),
                    152: (3, 8, 36, 3), 200: (3, 24, 36, 3)}

    class _GenericClassMethods(type):
        """any class attribute the stand-in Builder does not define (resnet_c1, resnet_unit, ... --
        models/tridentnet/resnet_v*.py subclasses mxnext's Builder and calls its classmethods) is a
        generic graph node constructor named after it"""

        def __getattr__(cls, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return lambda *a, **kw: _make("Builder." + k, a, kw)

    for fam in ("resnet_v1", "resnet_v1b", "resnet_v1d", "resnet_v2", "resnext"):
        m = types.ModuleType("mxnext.backbone." + fam)

        class Builder(metaclass=_GenericClassMethods):
            def get_backbone(self, variant, depth, endpoint, normalizer, fp16, **kw):
                data = S().var("data")
                stages = [Symbol("backbone_c%d" % i, [data], {"depth": depth, "fp16": fp16}, "c%d" % i)
                          for i in (2, 3, 4, 5)]
                return {"fpn": stages, "c4": stages[2], "c5": stages[3], "c4c5": (stages[2], stages[3])}[endpoint]

            def __getattr__(self, k):
                return getattr(type(self), k)

            @classmethod
            def resnet_stage(cls, data, name=None, **kw):
                return _make("resnet_stage", (data,), dict(kw, name=name))
        Builder.depth_config = depth_config
        m.Builder = Builder
        setattr(backbone, fam, m)
        sys_mods_extra["mxnext.backbone." + fam] = m
    for hname in ("resnet_v1_helper", "resnet_v1b_helper"):
        helper = types.ModuleType("mxnext.backbone." + hname)
        helper.depth_config = depth_config
        helper.resnet_unit = lambda data, name, filter, stride, dilate, proj, norm, **kw: _make(
            "resnet_unit", (data,), dict(name=name, filter=filter, stride=stride, dilate=dilate, proj=proj))
        helper.resnet_c1 = lambda data, norm: _make("resnet_c1", (data,), {})
        for st in ("resnet_c2", "resnet_c3", "resnet_c4", "resnet_c5"):
            setattr(helper, st, (lambda st: lambda data, n, stride, dilate, norm, **kw: _make(
                st, (data,), {"num_block": n, "stride": stride, "dilate": dilate}))(st))
        setattr(backbone, hname, helper)
        sys_mods_extra["mxnext.backbone." + hname] = helper
    m_dec = types.ModuleType("mxnext.tvm.decode_bbox")
    m_dec.decode_bbox = lambda *a, **kw: _make("mxnext.tvm.decode_bbox", a, kw)
    tvm.decode_bbox = m_dec
    sys_mods_extra["mxnext.tvm.decode_bbox"] = m_dec
    X.backbone = backbone
    mods = {"mxnext": X, "mxnext.tvm": tvm, "mxnext.tvm.fpn_roi_assign": m_assign,
            "mxnext.tvm.get_top_proposal": m_top, "mxnext.backbone": backbone,
            "mxnext.complicate": complicate,
            "mxnext.tvm.proposal": m_prop}
    mods.update(sys_mods_extra)
    return X, mods


class reference_modules:
    """context manager: `mxnet` / `mxnext` stubs in sys.modules, /root/reference on sys.path, and every
    module imported from the reference tree dropped again on exit"""

    def __init__(self, root="/root/reference", late_binding=True):
        self.root, self.late_binding = root, late_binding

    def __enter__(self):
        self.mx = make_mx()
        self.X, mods = make_mxnext(self.mx, self.late_binding)
        mods = dict(mods, mxnet=self.mx)
        mods["mxnet.symbol"] = self.mx.sym
        # operator_py/nms.py imports the tree's Cython extensions, which the reference builds in place
        # (`make` in operator_py/cython); here they are built out of tree into oracle/_ref
        # (oracle/build_ref.py) -- hand those over under the names the reference imports
        try:
            import importlib
            for m in ("cpu_nms", "bbox"):
                mods["operator_py.cython." + m] = importlib.import_module("oracle._ref." + m)
            # models/crowdhuman/input.py:9 (loader side, not on the path): name only
            ph = types.ModuleType("operator_py.cython.bbox_self")
            ph.bbox_selfoverlaps_cython = None
            mods["operator_py.cython.bbox_self"] = ph
        except Exception:   # not built: placeholders (the builders only import the names)
            for m in ("cpu_nms", "bbox", "bbox_self"):
                ph = types.ModuleType("operator_py.cython." + m)
                ph.greedy_nms = ph.soft_nms = ph.bbox_overlaps_cython = ph.bbox_selfoverlaps_cython = None
                mods["operator_py.cython." + m] = ph
        # `import operator_py.cython.bbox as x` walks the attribute chain operator_py -> cython -> bbox (with
        # sys.modules as the fall-back): the package itself has to be there too
        pkg = types.ModuleType("operator_py.cython")
        pkg.__path__ = []
        for m in ("cpu_nms", "bbox", "bbox_self"):
            setattr(pkg, m, mods["operator_py.cython." + m])
        mods["operator_py.cython"] = pkg
        # models/TSD/bbox_head.py:9 imports a shape-printing debug helper that is in no package index
        # (`from shape_tool import infer_shape`, never called)
        st = types.ModuleType("shape_tool")
        st.infer_shape = lambda *a, **kw: None
        st.enter_shape_infer_context = lambda *a, **kw: _Prefix("")      # config/TSD/tsd_r50_rpn_1x.py:142
        mods["shape_tool"] = st
        if "cv2" not in sys.modules:
            try:
                import cv2  # noqa: F401
            except Exception:
                mods["cv2"] = types.ModuleType("cv2")   # core/detection_input.py imports it at module scope
        if "pycocotools" not in sys.modules:
            try:
                import pycocotools  # noqa: F401
            except Exception:       # models/maskrcnn/input.py: `import pycocotools.mask as mask_util`
                pc = types.ModuleType("pycocotools")
                pc.__path__ = []
                pc.mask = types.ModuleType("pycocotools.mask")
                mods["pycocotools"], mods["pycocotools.mask"] = pc, pc.mask
        self.before = dict(sys.modules)
        sys.modules.update(mods)
        sys.path.insert(0, self.root)
        return self

    def __exit__(self, *a):
        sys.path.remove(self.root)
        for k in list(sys.modules):
            if k not in self.before:
                del sys.modules[k]
        for k, v in self.before.items():
            sys.modules[k] = v
        return False
