// runtime.cc -- the operator registry behind mxshim.h plus a C ABI to drive registered operators
// the way MXNet's executor does (Init(kwargs) -> InferShape/InferType -> CreateOperatorEx ->
// Forward/Backward for the legacy OperatorProperty interface; attr_parser -> FInferShape ->
// FCompute<cpu|gpu> for NNVM ops).  One copy is linked into every oracle/_ref/libref_<op>.so.
//
// TEST INFRASTRUCTURE ONLY (see mxshim.h).  Python side: oracle/refmx.py.
#include "mxshim.h"

namespace mxshim {

static std::map<std::string, PropEntry> &Props() {
  static std::map<std::string, PropEntry> m;
  return m;
}
static std::map<std::string, nnvm::Op> &Ops() {
  static std::map<std::string, nnvm::Op> m;
  return m;
}
PropEntry &RegisterProp(const char *name, std::function<mxnet::OperatorProperty *()> body) {
  PropEntry &e = Props()[name];
  e.name = name;
  e.body = body;
  return e;
}
nnvm::Op &RegisterOp(const char *name) {
  nnvm::Op &o = Ops()[name];
  o.name = name;
  return o;
}

}  // namespace mxshim

using mxnet::OpContext;
using mxnet::OpReqType;
using mxnet::TBlob;
using mxnet::TShape;

extern "C" {

typedef struct {
  void *data;
  int ndim;
  int64_t shape[8];
  int dtype;  // mshadow type flag: 0 f32, 1 f64, 3 u8, 4 i32
} mxref_array;

struct mxref_handle {
  bool legacy;
  std::unique_ptr<mxnet::OperatorProperty> prop;
  const nnvm::Op *op;
  nnvm::NodeAttrs attrs;
  mxref_handle() : legacy(true), op(NULL) {}
};

static thread_local std::string g_err;

const char *mxref_last_error(void) { return g_err.c_str(); }

#define MXREF_TRY try {
#define MXREF_CATCH(ret)                \
  }                                     \
  catch (const std::exception &e) {     \
    g_err = e.what();                   \
    return ret;                         \
  }

static int copy_names(const std::vector<std::string> &v, char *buf, int len) {
  std::string s;
  for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + v[i];
  if (static_cast<int>(s.size()) + 1 > len) { g_err = "buffer too small"; return -1; }
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return static_cast<int>(v.size());
}

int mxref_list_ops(char *buf, int len) {
  std::vector<std::string> v;
  for (auto &kv : mxshim::Props()) v.push_back(kv.first);
  for (auto &kv : mxshim::Ops()) v.push_back(kv.first);
  return copy_names(v, buf, len);
}

void mxref_srand(unsigned seed) { srand(seed); }
int mxref_rand(void) { return rand(); }

void *mxref_create(const char *op_name, int nkw, const char **keys, const char **vals) {
  MXREF_TRY
  std::unique_ptr<mxref_handle> h(new mxref_handle());
  auto pit = mxshim::Props().find(op_name);
  if (pit != mxshim::Props().end()) {
    h->legacy = true;
    h->prop.reset(pit->second.body());
    std::vector<std::pair<std::string, std::string> > kw;
    for (int i = 0; i < nkw; ++i) kw.push_back(std::make_pair(std::string(keys[i]), std::string(vals[i])));
    h->prop->Init(kw);
    return h.release();
  }
  auto oit = mxshim::Ops().find(op_name);
  if (oit == mxshim::Ops().end()) throw dmlc::Error(std::string("operator not registered: ") + op_name);
  h->legacy = false;
  h->op = &oit->second;
  h->attrs.name = op_name;
  for (int i = 0; i < nkw; ++i) h->attrs.dict[keys[i]] = vals[i];
  if (h->op->attr_parser) h->op->attr_parser(&h->attrs);
  return h.release();
  MXREF_CATCH(NULL)
}

void mxref_free(void *p) { delete static_cast<mxref_handle *>(p); }

int mxref_is_legacy(void *p) { return static_cast<mxref_handle *>(p)->legacy ? 1 : 0; }

int mxref_num_outputs(void *p) {
  mxref_handle *h = static_cast<mxref_handle *>(p);
  MXREF_TRY
  return h->legacy ? h->prop->NumOutputs() : h->op->num_outputs;
  MXREF_CATCH(-1)
}

int mxref_num_visible_outputs(void *p) {
  mxref_handle *h = static_cast<mxref_handle *>(p);
  MXREF_TRY
  if (h->legacy) return h->prop->NumVisibleOutputs();
  const nnvm::FNumVisibleOutputs *f = h->op->get_attr<nnvm::FNumVisibleOutputs>("FNumVisibleOutputs");
  return f ? static_cast<int>((*f)(h->attrs)) : h->op->num_outputs;
  MXREF_CATCH(-1)
}

int mxref_list_arguments(void *p, char *buf, int len) {
  mxref_handle *h = static_cast<mxref_handle *>(p);
  MXREF_TRY
  if (h->legacy) return copy_names(h->prop->ListArguments(), buf, len);
  const nnvm::FListInputNames *f = h->op->get_attr<nnvm::FListInputNames>("FListInputNames");
  return copy_names(f ? (*f)(h->attrs) : std::vector<std::string>(), buf, len);
  MXREF_CATCH(-1)
}

int mxref_list_outputs(void *p, char *buf, int len) {
  mxref_handle *h = static_cast<mxref_handle *>(p);
  MXREF_TRY
  if (h->legacy) return copy_names(h->prop->ListOutputs(), buf, len);
  const nnvm::FListOutputNames *f = h->op->get_attr<nnvm::FListOutputNames>("FListOutputNames");
  return copy_names(f ? (*f)(h->attrs) : std::vector<std::string>(), buf, len);
  MXREF_CATCH(-1)
}

// which arrays the backward pass needs: indices into [out_grad..., in_data..., out_data...]
int mxref_backward_dependency(void *p, int nin, int *dep, int maxdep) {
  mxref_handle *h = static_cast<mxref_handle *>(p);
  MXREF_TRY
  if (!h->legacy) throw dmlc::Error("backward_dependency: legacy operators only");
  const int nout = h->prop->NumOutputs();
  std::vector<int> og(nout), id(nin), od(nout);
  int k = 0;
  for (int i = 0; i < nout; ++i) og[i] = k++;
  for (int i = 0; i < nin; ++i) id[i] = k++;
  for (int i = 0; i < nout; ++i) od[i] = k++;
  std::vector<int> d = h->prop->DeclareBackwardDependency(og, id, od);
  if (static_cast<int>(d.size()) > maxdep) throw dmlc::Error("dep buffer too small");
  for (size_t i = 0; i < d.size(); ++i) dep[i] = d[i];
  return static_cast<int>(d.size());
  MXREF_CATCH(-1)
}

// for an NNVM op: which entries its FGradient hands to the backward node, encoded as
// 0..nout-1 = out_grad[i], 100+i = forward input i, 200+i = forward output i
int mxref_gradient_inputs(void *p, int nin, int *enc, int maxenc, char *bwd_name, int len) {
  mxref_handle *h = static_cast<mxref_handle *>(p);
  MXREF_TRY
  if (h->legacy) throw dmlc::Error("gradient_inputs: NNVM operators only");
  const nnvm::FGradient *f = h->op->get_attr<nnvm::FGradient>("FGradient");
  if (!f) throw dmlc::Error("operator has no FGradient");
  nnvm::NodePtr n = std::make_shared<nnvm::Node>();
  n->attrs = h->attrs;
  std::vector<nnvm::NodePtr> in_nodes, og_nodes;
  for (int i = 0; i < nin; ++i) {
    in_nodes.push_back(std::make_shared<nnvm::Node>());
    nnvm::NodeEntry e; e.node = in_nodes.back(); e.index = 0; e.version = 0;
    n->inputs.push_back(e);
  }
  std::vector<nnvm::NodeEntry> ograds;
  for (int i = 0; i < h->op->num_outputs; ++i) {
    og_nodes.push_back(std::make_shared<nnvm::Node>());
    nnvm::NodeEntry e; e.node = og_nodes.back(); e.index = 0; e.version = 0;
    ograds.push_back(e);
  }
  std::vector<nnvm::NodeEntry> g = (*f)(n, ograds);
  if (g.empty()) throw dmlc::Error("FGradient returned nothing");
  const nnvm::NodePtr &b = g[0].node;
  if (static_cast<int>(b->inputs.size()) > maxenc) throw dmlc::Error("enc buffer too small");
  for (size_t i = 0; i < b->inputs.size(); ++i) {
    const nnvm::NodeEntry &e = b->inputs[i];
    int code = -1;
    if (e.node == n) code = 200 + static_cast<int>(e.index);
    for (int j = 0; j < nin && code < 0; ++j) if (e.node == in_nodes[j]) code = 100 + j;
    for (size_t j = 0; j < og_nodes.size() && code < 0; ++j) if (e.node == og_nodes[j]) code = static_cast<int>(j);
    enc[i] = code;
  }
  if (static_cast<int>(b->attrs.name.size()) + 1 > len) throw dmlc::Error("name buffer too small");
  std::memcpy(bwd_name, b->attrs.name.c_str(), b->attrs.name.size() + 1);
  return static_cast<int>(b->inputs.size());
  MXREF_CATCH(-1)
}

static TShape to_tshape(const mxref_array &a) { return TShape(a.shape, a.shape + a.ndim); }
static void from_tshape(const TShape &s, mxref_array *a) {
  if (s.ndim() > 8) throw dmlc::Error("ndim > 8");
  a->ndim = s.ndim();
  for (int i = 0; i < s.ndim(); ++i) a->shape[i] = s[i];
}

int mxref_infer_shape(void *p, int nin, const mxref_array *in, int maxout, mxref_array *out) {
  mxref_handle *h = static_cast<mxref_handle *>(p);
  MXREF_TRY
  std::vector<TShape> is, os, as;
  for (int i = 0; i < nin; ++i) is.push_back(to_tshape(in[i]));
  bool ok;
  if (h->legacy) {
    ok = h->prop->InferShape(&is, &os, &as);
  } else {
    const mxnet::FInferShape *f = h->op->get_attr<mxnet::FInferShape>("FInferShape");
    if (!f) throw dmlc::Error("operator has no FInferShape");
    os.resize(h->op->num_outputs);
    ok = (*f)(h->attrs, &is, &os);
  }
  if (!ok) throw dmlc::Error("InferShape returned false");
  if (static_cast<int>(os.size()) > maxout) throw dmlc::Error("out buffer too small");
  for (size_t i = 0; i < os.size(); ++i) from_tshape(os[i], &out[i]);
  return static_cast<int>(os.size());
  MXREF_CATCH(-1)
}

int mxref_infer_type(void *p, int nin, const int *in_types, int maxout, int *out_types) {
  mxref_handle *h = static_cast<mxref_handle *>(p);
  MXREF_TRY
  std::vector<int> it(in_types, in_types + nin), ot, at;
  bool ok;
  if (h->legacy) {
    ok = h->prop->InferType(&it, &ot, &at);
  } else {
    const nnvm::FInferType *f = h->op->get_attr<nnvm::FInferType>("FInferType");
    if (!f) throw dmlc::Error("operator has no FInferType");
    ot.resize(h->op->num_outputs, -1);
    ok = (*f)(h->attrs, &it, &ot);
  }
  if (!ok) throw dmlc::Error("InferType returned false");
  if (static_cast<int>(ot.size()) > maxout) throw dmlc::Error("out buffer too small");
  for (size_t i = 0; i < ot.size(); ++i) out_types[i] = ot[i];
  return static_cast<int>(ot.size());
  MXREF_CATCH(-1)
}

static std::vector<TBlob> blobs(int n, const mxref_array *a, int dev_mask) {
  std::vector<TBlob> v;
  for (int i = 0; i < n; ++i) v.push_back(TBlob(a[i].data, to_tshape(a[i]), dev_mask, a[i].dtype));
  return v;
}

static mxnet::Operator *make_operator(mxref_handle *h, int dev, int nin, const mxref_array *in) {
  std::vector<TShape> is;
  std::vector<int> it;
  for (int i = 0; i < nin; ++i) { is.push_back(to_tshape(in[i])); it.push_back(in[i].dtype); }
  mxnet::Context ctx = (dev == 2) ? mxnet::Context::GPU(0) : mxnet::Context::CPU(0);
  mxnet::Operator *op = h->prop->CreateOperatorEx(ctx, &is, &it);
  if (op == NULL) throw dmlc::Error("CreateOperatorEx returned NULL");
  return op;
}

static OpContext make_ctx(mxref_handle *h, int is_train, int nin, const mxref_array *in, bool fwd) {
  OpContext ctx;
  ctx.is_train = is_train != 0;
  std::vector<mxnet::ResourceRequest> rr;
  if (h->legacy) {
    std::vector<TShape> is;
    for (int i = 0; i < nin; ++i) is.push_back(to_tshape(in[i]));
    rr = fwd ? h->prop->ForwardResource(is) : h->prop->BackwardResource(is);
  }
  for (size_t i = 0; i < rr.size(); ++i) {
    mxnet::Resource r;
    r.req = rr[i];
    ctx.requested.push_back(r);
  }
  return ctx;
}

// dev: 1 = the reference's CPU operator, 2 = its GPU operator (compiled through cuemu.h)
int mxref_forward(void *p, int dev, int is_train, int nin, const mxref_array *in, int nout,
                  const mxref_array *out, const int *req) {
  mxref_handle *h = static_cast<mxref_handle *>(p);
  MXREF_TRY
  const int mask = (dev == 2) ? mxnet::gpu::kDevMask : mxnet::cpu::kDevMask;
  std::vector<TBlob> bi = blobs(nin, in, mask), bo = blobs(nout, out, mask);
  std::vector<OpReqType> rq;
  for (int i = 0; i < nout; ++i) rq.push_back(static_cast<OpReqType>(req ? req[i] : 1));
  OpContext ctx = make_ctx(h, is_train, nin, in, true);
  if (h->legacy) {
    std::unique_ptr<mxnet::Operator> op(make_operator(h, dev, nin, in));
    op->Forward(ctx, bi, rq, bo, std::vector<TBlob>());
  } else {
    const mxnet::FCompute *f = h->op->get_attr<mxnet::FCompute>(dev == 2 ? "FCompute<gpu>" : "FCompute<cpu>");
    if (!f) throw dmlc::Error("operator has no FCompute for this device");
    (*f)(h->attrs, ctx, bi, rq, bo);
  }
  return 0;
  MXREF_CATCH(-1)
}

int mxref_backward(void *p, int dev, int nog, const mxref_array *out_grad, int nin,
                   const mxref_array *in_data, int nout, const mxref_array *out_data, int nig,
                   const mxref_array *in_grad, const int *req) {
  mxref_handle *h = static_cast<mxref_handle *>(p);
  MXREF_TRY
  if (!h->legacy) throw dmlc::Error("backward: create the _backward_* operator and call forward");
  const int mask = (dev == 2) ? mxnet::gpu::kDevMask : mxnet::cpu::kDevMask;
  std::vector<TBlob> og = blobs(nog, out_grad, mask), id = blobs(nin, in_data, mask),
                     od = blobs(nout, out_data, mask), ig = blobs(nig, in_grad, mask);
  std::vector<OpReqType> rq;
  for (int i = 0; i < nig; ++i) rq.push_back(static_cast<OpReqType>(req ? req[i] : 1));
  OpContext ctx = make_ctx(h, 1, nin, in_data, false);
  std::unique_ptr<mxnet::Operator> op(make_operator(h, dev, nin, in_data));
  op->Backward(ctx, og, id, od, rq, ig, std::vector<TBlob>());
  return 0;
  MXREF_CATCH(-1)
}

}  // extern "C"
