// mxshim.h -- a minimal stand-in for the MXNet 1.6 / mshadow / dmlc-core / nnvm headers.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under simpledet_amd/ includes, links or loads this.
//
// Purpose: let the reference's operator sources (/root/reference/operator_cxx/**/*.cc and, with
// cuemu.h, *.cu) compile UNMODIFIED, where they lie, into oracle/_ref/libref_<op>.so, so that the
// CPU oracle (oracle/*.c) and the HIP kernels can be compared with numbers the reference's own
// code produced.  MXNet itself is not installable here (no network, no headers); this file
// re-declares only the API surface those sources touch:
//   dmlc:    CHECK*/LOG, Parameter<> with DMLC_DECLARE_PARAMETER/FIELD, any
//   nnvm:    Tuple<>, NodeAttrs, Op registry (NNVM_REGISTER_OP ... set_attr<>)
//   mshadow: cpu/gpu, index_t, Shape<N>, Tensor<Dev,N,T>, TensorContainer, Copy, scalar / vector
//            assignment expressions, Stream<>
//   mxnet:   TShape, TBlob, OpContext, Resource, Operator, OperatorProperty,
//            MXNET_REGISTER_OP_PROPERTY, DO_BIND_DISPATCH, mxnet_op::Kernel<>::Launch, Fill,
//            mshadow_op::{minimum,maximum,floor,ceil}, MSHADOW_REAL_TYPE_SWITCH
// Everything here is written from the documented behaviour of those APIs; no MXNet source was
// available to copy.  Where a choice could influence numbers it is listed in DESIGN.md section 2
// (index_t = int32, TensorContainer without row padding, half_t absent: the type switch covers
// float and double only).
#ifndef ORACLE_MXSHIM_H_
#define ORACLE_MXSHIM_H_

#include <algorithm>
#include <cctype>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <typeinfo>
#include <unordered_map>
#include <utility>
#include <vector>

#define DMLC_USE_CXX11 1
#define MSHADOW_XINLINE inline
#define MSHADOW_CINLINE inline
#define MSHADOW_FORCE_INLINE inline
#define MXNET_USE_CUDA 1
#define ADD_FILELINE ""

// ------------------------------------------------------------------------------------------ dmlc
namespace dmlc {

struct Error : public std::runtime_error {
  explicit Error(const std::string &s) : std::runtime_error(s) {}
};
struct ParamError : public Error {
  explicit ParamError(const std::string &s) : Error(s) {}
};

class LogMessageFatal {
 public:
  LogMessageFatal(const char *file, int line) { s_ << file << ":" << line << ": "; }
  std::ostringstream &stream() { return s_; }
  ~LogMessageFatal() noexcept(false) { throw Error(s_.str()); }

 private:
  std::ostringstream s_;
};
class LogMessage {
 public:
  std::ostream &stream() { return std::cerr; }
  ~LogMessage() { std::cerr << std::endl; }
};

// a tiny type-erased value (dmlc::any)
class any {
 public:
  any() : t_(NULL) {}
  template <typename T>
  any(const T &v) : p_(std::make_shared<T>(v)), t_(&typeid(T)) {}  // NOLINT
  bool empty() const { return !p_; }
  const std::type_info &type() const { return t_ ? *t_ : typeid(void); }
  template <typename T>
  const T &as() const {
    if (!p_ || *t_ != typeid(T)) throw Error(std::string("dmlc::any: bad cast to ") + typeid(T).name());
    return *static_cast<const T *>(p_.get());
  }

 private:
  std::shared_ptr<void> p_;
  const std::type_info *t_;
};
template <typename T>
inline const T &get(const any &a) { return a.as<T>(); }

struct ParamFieldInfo {
  std::string name, type, type_info_str, description;
};

namespace parameter {

template <typename T>
struct ValueIO {
  static void Parse(const std::string &s, T *out) {
    std::istringstream is(s);
    is >> *out;
    if (is.fail()) throw ParamError("Invalid Parameter format, value='" + s + "'");
    while (true) {  // only trailing white space (or an 'f'/'L' suffix) may follow
      int ch = is.get();
      if (ch == EOF) break;
      if (!isspace(ch) && ch != 'f' && ch != 'L')
        throw ParamError("Some trailing characters could not be parsed: '" + s + "'");
    }
  }
  static std::string Str(const T &v) {
    std::ostringstream os;
    os << v;
    return os.str();
  }
};
template <>
struct ValueIO<bool> {
  static void Parse(const std::string &s, bool *out) {
    std::string l;
    for (size_t i = 0; i < s.size(); ++i)
      if (!isspace(s[i])) l.push_back(static_cast<char>(tolower(s[i])));
    if (l == "true" || l == "1") *out = true;
    else if (l == "false" || l == "0") *out = false;
    else throw ParamError("Invalid bool value '" + s + "'");
  }
  static std::string Str(const bool &v) { return v ? "True" : "False"; }
};

struct FieldAccessEntry {
  std::string key;
  bool has_default;
  FieldAccessEntry() : has_default(false) {}
  virtual ~FieldAccessEntry() {}
  virtual void Set(const std::string &value) = 0;
  virtual std::string Get() const = 0;
};

template <typename T>
struct FieldEntry : public FieldAccessEntry {
  T *ptr;
  std::map<std::string, int> enums;
  bool has_lo, has_hi;
  double lo, hi;
  FieldEntry() : ptr(NULL), has_lo(false), has_hi(false), lo(0), hi(0) {}
  template <typename V>
  FieldEntry &set_default(const V &v) {
    *ptr = static_cast<T>(v);
    has_default = true;
    return *this;
  }
  FieldEntry &describe(const std::string &) { return *this; }
  template <typename A, typename B>
  FieldEntry &set_range(A a, B b) {
    has_lo = has_hi = true;
    lo = static_cast<double>(a);
    hi = static_cast<double>(b);
    return *this;
  }
  template <typename A>
  FieldEntry &set_lower_bound(A a) {
    has_lo = true;
    lo = static_cast<double>(a);
    return *this;
  }
  FieldEntry &set_expect_ndim(int) { return *this; }
  FieldEntry &enforce_nonzero() { return *this; }
  FieldEntry &add_enum(const std::string &name, int v) {
    enums[name] = v;
    return *this;
  }
  template <typename U>
  static void AssignEnum(U *, int) { throw ParamError("enum on a non-int field"); }
  static void AssignEnum(int *p, int v) { *p = v; }
  template <typename U>
  static bool AsDouble(const U &, double *) { return false; }
  static bool AsDouble(const float &v, double *d) { *d = v; return true; }
  static bool AsDouble(const double &v, double *d) { *d = v; return true; }
  static bool AsDouble(const int &v, double *d) { *d = v; return true; }
  void Set(const std::string &value) override {
    if (!enums.empty()) {
      std::string v = value;
      while (!v.empty() && (isspace(v[0]) || v[0] == '\'' || v[0] == '"')) v.erase(0, 1);
      while (!v.empty() && (isspace(v[v.size() - 1]) || v[v.size() - 1] == '\'' ||
                            v[v.size() - 1] == '"'))
        v.erase(v.size() - 1);
      std::map<std::string, int>::const_iterator it = enums.find(v);
      if (it == enums.end()) throw ParamError("Invalid enum value '" + value + "' for " + key);
      AssignEnum(ptr, it->second);
      return;
    }
    ValueIO<T>::Parse(value, ptr);
    double d;
    if ((has_lo || has_hi) && AsDouble(*ptr, &d)) {
      if ((has_lo && d < lo) || (has_hi && d > hi))
        throw ParamError("value " + value + " for Parameter " + key + " exceeds bound");
    }
  }
  std::string Get() const override { return ValueIO<T>::Str(*ptr); }
};

struct ParamManager {
  std::vector<std::unique_ptr<FieldAccessEntry> > entries;
  FieldAccessEntry *Find(const std::string &key) {
    for (size_t i = 0; i < entries.size(); ++i)
      if (entries[i]->key == key) return entries[i].get();
    return NULL;
  }
};

}  // namespace parameter

template <typename PType>
struct Parameter {
  template <typename Container>
  void Init(const Container &kwargs) {
    parameter::ParamManager m;
    static_cast<PType *>(this)->__DECLARE__(&m);
    std::set<std::string> seen;
    for (typename Container::const_iterator it = kwargs.begin(); it != kwargs.end(); ++it) {
      parameter::FieldAccessEntry *e = m.Find(it->first);
      if (e == NULL) {
        // MXNet strips its own bookkeeping attributes (__xxx__) before Init
        if (it->first.size() > 4 && it->first.compare(0, 2, "__") == 0) continue;
        throw ParamError("Cannot find argument '" + it->first + "'");
      }
      e->Set(it->second);
      seen.insert(it->first);
    }
    for (size_t i = 0; i < m.entries.size(); ++i)
      if (!m.entries[i]->has_default && !seen.count(m.entries[i]->key))
        throw ParamError("Required parameter " + m.entries[i]->key + " is not presented");
  }
  std::map<std::string, std::string> __DICT__() const {
    PType tmp(*static_cast<const PType *>(this));
    PType keep(tmp);
    parameter::ParamManager m;
    tmp.__DECLARE__(&m);  // resets defaulted fields of tmp ...
    tmp = keep;           // ... so put the values back (the entries point into tmp)
    std::map<std::string, std::string> d;
    for (size_t i = 0; i < m.entries.size(); ++i) d[m.entries[i]->key] = m.entries[i]->Get();
    return d;
  }
  static std::vector<ParamFieldInfo> __FIELDS__() { return std::vector<ParamFieldInfo>(); }

 protected:
  template <typename T>
  parameter::FieldEntry<T> &DECLARE(parameter::ParamManager *m, const std::string &key, T &ref) {
    parameter::FieldEntry<T> *e = new parameter::FieldEntry<T>();
    e->key = key;
    e->ptr = &ref;
    m->entries.push_back(std::unique_ptr<parameter::FieldAccessEntry>(e));
    return *e;
  }
};

}  // namespace dmlc

#define DMLC_DECLARE_PARAMETER(PType) \
  inline void __DECLARE__(::dmlc::parameter::ParamManager *manager)
#define DMLC_DECLARE_FIELD(FieldName) this->DECLARE(manager, #FieldName, FieldName)
#define MXSHIM_CAT_(a, b) a##b
#define MXSHIM_CAT(a, b) MXSHIM_CAT_(a, b)
#define DMLC_REGISTER_PARAMETER(PType) \
  static int MXSHIM_CAT(__mxshim_param_reg_, __COUNTER__) __attribute__((unused)) = 0
#define DMLC_ATTRIBUTE_UNUSED __attribute__((unused))

#define CHECK(x) \
  if (!(x)) ::dmlc::LogMessageFatal(__FILE__, __LINE__).stream() << "Check failed: " #x << ' '
#define MXSHIM_CHECK_OP(op, x, y)                                     \
  if (!((x)op(y)))                                                    \
  ::dmlc::LogMessageFatal(__FILE__, __LINE__).stream()                \
      << "Check failed: " #x " " #op " " #y << " "
#define CHECK_EQ(x, y) MXSHIM_CHECK_OP(==, x, y)
#define CHECK_NE(x, y) MXSHIM_CHECK_OP(!=, x, y)
#define CHECK_LT(x, y) MXSHIM_CHECK_OP(<, x, y)
#define CHECK_GT(x, y) MXSHIM_CHECK_OP(>, x, y)
#define CHECK_LE(x, y) MXSHIM_CHECK_OP(<=, x, y)
#define CHECK_GE(x, y) MXSHIM_CHECK_OP(>=, x, y)
#define CHECK_NOTNULL(x) (x)
#define DCHECK(x) CHECK(x)
#define DCHECK_EQ(x, y) CHECK_EQ(x, y)
#define LOG_FATAL ::dmlc::LogMessageFatal(__FILE__, __LINE__).stream()
#define LOG_INFO ::dmlc::LogMessage().stream()
#define LOG_WARNING ::dmlc::LogMessage().stream()
#define LOG_ERROR ::dmlc::LogMessage().stream()
#define LOG(severity) LOG_##severity

// --------------------------------------------------------------------------------------- mshadow
typedef void *cudaStream_t;

namespace mshadow {

typedef int32_t index_t;  // MXNet 1.6 default build (MSHADOW_INT64_TENSOR_SIZE off)
typedef float real_t;
typedef real_t default_real_t;

struct cpu {
  static const bool kDevCPU = true;
  static const int kDevMask = 1 << 0;
};
struct gpu {
  static const bool kDevCPU = false;
  static const int kDevMask = 1 << 1;
};

enum TypeFlag { kFloat32 = 0, kFloat64 = 1, kFloat16 = 2, kUint8 = 3, kInt32 = 4, kInt8 = 5, kInt64 = 6 };
template <typename DType> struct DataType;
template <> struct DataType<float> { static const int kFlag = kFloat32; };
template <> struct DataType<double> { static const int kFlag = kFloat64; };
template <> struct DataType<uint8_t> { static const int kFlag = kUint8; };
template <> struct DataType<int32_t> { static const int kFlag = kInt32; };
template <> struct DataType<int8_t> { static const int kFlag = kInt8; };
template <> struct DataType<int64_t> { static const int kFlag = kInt64; };

namespace red {
namespace limits {
template <typename DType> inline DType MinValue();
template <> inline float MinValue<float>() { return -FLT_MAX; }
template <> inline double MinValue<double>() { return -DBL_MAX; }
template <> inline int32_t MinValue<int32_t>() { return INT_MIN; }
template <typename DType> inline DType MaxValue();
template <> inline float MaxValue<float>() { return FLT_MAX; }
template <> inline double MaxValue<double>() { return DBL_MAX; }
}  // namespace limits
}  // namespace red

template <int dimension>
struct Shape {
  static const int kDimension = dimension;
  index_t shape_[dimension];
  Shape() { for (int i = 0; i < dimension; ++i) shape_[i] = 0; }
  index_t &operator[](int i) { return shape_[i]; }
  const index_t &operator[](int i) const { return shape_[i]; }
  bool operator==(const Shape &o) const {
    for (int i = 0; i < dimension; ++i) if (shape_[i] != o.shape_[i]) return false;
    return true;
  }
  bool operator!=(const Shape &o) const { return !(*this == o); }
  size_t Size() const {
    size_t s = 1;
    for (int i = 0; i < dimension; ++i) s *= static_cast<size_t>(shape_[i]);
    return s;
  }
  size_t ProdShape(int b, int e) const {
    size_t s = 1;
    for (int i = b; i < e; ++i) s *= static_cast<size_t>(shape_[i]);
    return s;
  }
  Shape<dimension - 1> SubShape() const {
    Shape<dimension - 1> s;
    for (int i = 1; i < dimension; ++i) s.shape_[i - 1] = shape_[i];
    return s;
  }
};
template <>
struct Shape<0> {
  index_t shape_[1];
};
template <int dim>
inline std::ostream &operator<<(std::ostream &os, const Shape<dim> &s) {
  os << '(';
  for (int i = 0; i < dim; ++i) os << (i ? "," : "") << s[i];
  return os << ')';
}
inline Shape<1> Shape1(index_t a) { Shape<1> s; s[0] = a; return s; }
inline Shape<2> Shape2(index_t a, index_t b) { Shape<2> s; s[0] = a; s[1] = b; return s; }
inline Shape<3> Shape3(index_t a, index_t b, index_t c) {
  Shape<3> s; s[0] = a; s[1] = b; s[2] = c; return s;
}
inline Shape<4> Shape4(index_t a, index_t b, index_t c, index_t d) {
  Shape<4> s; s[0] = a; s[1] = b; s[2] = c; s[3] = d; return s;
}
inline Shape<5> Shape5(index_t a, index_t b, index_t c, index_t d, index_t e) {
  Shape<5> s; s[0] = a; s[1] = b; s[2] = c; s[3] = d; s[4] = e; return s;
}

template <typename Device>
struct Stream {
  cudaStream_t stream_;
  Stream() : stream_(NULL) {}
  void Wait() {}
  static cudaStream_t GetStream(Stream<Device> *s) { return s ? s->stream_ : NULL; }
};

namespace expr {}  // "using namespace mshadow::expr" must parse

template <typename Device, int dimension, typename DType = default_real_t>
struct Tensor {
  static const int kSubdim = dimension - 1;
  DType *dptr_;
  Shape<dimension> shape_;
  index_t stride_;
  Stream<Device> *stream_;
  Tensor() : dptr_(NULL), stride_(0), stream_(NULL) {}
  explicit Tensor(const Shape<dimension> &shape)
      : dptr_(NULL), shape_(shape), stride_(shape[dimension - 1]), stream_(NULL) {}
  Tensor(DType *dptr, const Shape<dimension> &shape)
      : dptr_(dptr), shape_(shape), stride_(shape[dimension - 1]), stream_(NULL) {}
  Tensor(DType *dptr, const Shape<dimension> &shape, Stream<Device> *stream)
      : dptr_(dptr), shape_(shape), stride_(shape[dimension - 1]), stream_(stream) {}
  Tensor(DType *dptr, const Shape<dimension> &shape, index_t stride, Stream<Device> *stream)
      : dptr_(dptr), shape_(shape), stride_(stride), stream_(stream) {}
  void set_stream(Stream<Device> *s) { stream_ = s; }
  index_t size(int idx) const { return shape_[idx]; }
  bool CheckContiguous() const { return shape_[dimension - 1] == stride_; }
  size_t MSize() const {
    size_t s = static_cast<size_t>(stride_);
    for (int i = 0; i < dimension - 1; ++i) s *= static_cast<size_t>(shape_[i]);
    return s;
  }
  size_t RowSpan() const {  // elements between consecutive values of the first index
    size_t s = static_cast<size_t>(stride_);
    for (int i = 1; i < dimension - 1; ++i) s *= static_cast<size_t>(shape_[i]);
    return s;
  }
  Tensor<Device, dimension - 1, DType> operator[](index_t idx) const {
    return Tensor<Device, dimension - 1, DType>(dptr_ + RowSpan() * static_cast<size_t>(idx),
                                                shape_.SubShape(), stride_, stream_);
  }
  Tensor Slice(index_t begin, index_t end) const {
    Tensor s(*this);
    s.dptr_ = dptr_ + RowSpan() * static_cast<size_t>(begin);
    s.shape_[0] = end - begin;
    return s;
  }
  Tensor<Device, 1, DType> FlatTo1D() const {
    return Tensor<Device, 1, DType>(dptr_, Shape1(static_cast<index_t>(shape_.Size())), stream_);
  }
  Tensor<Device, 2, DType> FlatTo2D() const {
    return Tensor<Device, 2, DType>(
        dptr_, Shape2(static_cast<index_t>(shape_.ProdShape(0, dimension - 1)), shape_[dimension - 1]),
        stride_, stream_);
  }
  // scalar fill ("t = 0.f")
  Tensor &operator=(const DType &v) {
    const size_t rows = shape_.ProdShape(0, dimension - 1);
    for (size_t r = 0; r < rows; ++r)
      for (index_t c = 0; c < shape_[dimension - 1]; ++c) dptr_[r * stride_ + c] = v;
    return *this;
  }
  Tensor &operator*=(const DType &v) {
    const size_t rows = shape_.ProdShape(0, dimension - 1);
    for (size_t r = 0; r < rows; ++r)
      for (index_t c = 0; c < shape_[dimension - 1]; ++c) dptr_[r * stride_ + c] *= v;
    return *this;
  }
};

template <typename Device, typename DType>
struct Tensor<Device, 1, DType> {
  DType *dptr_;
  Shape<1> shape_;
  index_t stride_;
  Stream<Device> *stream_;
  Tensor() : dptr_(NULL), stride_(0), stream_(NULL) {}
  explicit Tensor(const Shape<1> &shape) : dptr_(NULL), shape_(shape), stride_(shape[0]), stream_(NULL) {}
  Tensor(DType *dptr, const Shape<1> &shape) : dptr_(dptr), shape_(shape), stride_(shape[0]), stream_(NULL) {}
  Tensor(DType *dptr, const Shape<1> &shape, Stream<Device> *stream)
      : dptr_(dptr), shape_(shape), stride_(shape[0]), stream_(stream) {}
  Tensor(DType *dptr, const Shape<1> &shape, index_t stride, Stream<Device> *stream)
      : dptr_(dptr), shape_(shape), stride_(stride), stream_(stream) {}
  void set_stream(Stream<Device> *s) { stream_ = s; }
  index_t size(int) const { return shape_[0]; }
  bool CheckContiguous() const { return true; }
  size_t MSize() const { return static_cast<size_t>(shape_[0]); }
  DType &operator[](index_t idx) { return dptr_[idx]; }
  const DType &operator[](index_t idx) const { return dptr_[idx]; }
  Tensor Slice(index_t begin, index_t end) const {
    Tensor s(*this);
    s.dptr_ = dptr_ + begin;
    s.shape_[0] = end - begin;
    return s;
  }
  Tensor<Device, 1, DType> FlatTo1D() const { return *this; }
  Tensor &operator=(const DType &v) {
    for (index_t i = 0; i < shape_[0]; ++i) dptr_[i] = v;
    return *this;
  }
#define MXSHIM_VEC_OP(OP)                                                            \
  Tensor &operator OP(const Tensor<Device, 1, DType> &o) {                           \
    if (o.shape_[0] != shape_[0]) throw ::dmlc::Error("mshadow: shape mismatch in " #OP); \
    for (index_t i = 0; i < shape_[0]; ++i) dptr_[i] OP o.dptr_[i];                  \
    return *this;                                                                    \
  }                                                                                  \
  Tensor &operator OP(const DType &v) {                                              \
    for (index_t i = 0; i < shape_[0]; ++i) dptr_[i] OP v;                           \
    return *this;                                                                    \
  }
  MXSHIM_VEC_OP(+=)
  MXSHIM_VEC_OP(-=)
  MXSHIM_VEC_OP(*=)
  MXSHIM_VEC_OP(/=)
#undef MXSHIM_VEC_OP
};

// Copy between any two "devices" (both are host memory here), same shape required
template <typename A, typename B, int dim, typename DType>
inline void Copy(Tensor<A, dim, DType> dst, const Tensor<B, dim, DType> &src) {
  if (!(dst.shape_ == src.shape_)) {
    std::ostringstream os;
    os << "Copy:shape mismatch:" << dst.shape_ << " vs " << src.shape_;
    throw ::dmlc::Error(os.str());
  }
  const size_t rows = dst.shape_.ProdShape(0, dim - 1);
  const index_t cols = dst.shape_[dim - 1];
  for (size_t r = 0; r < rows; ++r)
    std::memcpy(dst.dptr_ + r * dst.stride_, src.dptr_ + r * src.stride_, sizeof(DType) * cols);
}
template <typename A, typename B, int dim, typename DType, typename S>
inline void Copy(Tensor<A, dim, DType> dst, const Tensor<B, dim, DType> &src, S *) {
  Copy(dst, src);
}

template <typename Device, int dimension, typename DType = default_real_t>
class TensorContainer : public Tensor<Device, dimension, DType> {
 public:
  typedef Tensor<Device, dimension, DType> Base;
  explicit TensorContainer(bool pad = false) {}
  explicit TensorContainer(const Shape<dimension> &shape) { Alloc(shape); }
  TensorContainer(const Shape<dimension> &shape, DType initv) { Alloc(shape); Fill(initv); }
  TensorContainer(const TensorContainer &o) : Base() { CopyFrom(o); }
  TensorContainer &operator=(const TensorContainer &o) {
    if (this != &o) CopyFrom(o);
    return *this;
  }
  void Resize(const Shape<dimension> &shape) {
    // mshadow keeps the old contents where they overlap; callers here only Resize empty ones
    std::vector<DType> old;
    old.swap(store_);
    Alloc(shape);
    std::copy(old.begin(), old.begin() + std::min(old.size(), store_.size()), store_.begin());
  }
  void Resize(const Shape<dimension> &shape, DType initv) { Alloc(shape); Fill(initv); }
  TensorContainer &operator=(const DType &v) { Fill(v); return *this; }

 private:
  std::vector<DType> store_;
  void Alloc(const Shape<dimension> &shape) {
    this->shape_ = shape;
    this->stride_ = shape[dimension - 1];
    store_.assign(shape.Size(), DType());
    this->dptr_ = store_.empty() ? NULL : &store_[0];
  }
  void Fill(DType v) { std::fill(store_.begin(), store_.end(), v); }
  void CopyFrom(const TensorContainer &o) {
    store_ = o.store_;
    this->shape_ = o.shape_;
    this->stride_ = o.stride_;
    this->stream_ = o.stream_;
    this->dptr_ = store_.empty() ? NULL : &store_[0];
  }
};

}  // namespace mshadow

#define MSHADOW_REAL_TYPE_SWITCH(type, DType, ...)                                     \
  switch (type) {                                                                      \
    case ::mshadow::kFloat32: { typedef float DType; {__VA_ARGS__} } break;            \
    case ::mshadow::kFloat64: { typedef double DType; {__VA_ARGS__} } break;           \
    case ::mshadow::kFloat16:                                                          \
      LOG(FATAL) << "mxshim: float16 is not provided by the shim (half_t absent)";     \
      break;                                                                           \
    default: LOG(FATAL) << "Unknown type enum " << type;                               \
  }
#define MSHADOW_SGL_DBL_TYPE_SWITCH(type, DType, ...) MSHADOW_REAL_TYPE_SWITCH(type, DType, __VA_ARGS__)
#define MSHADOW_TYPE_SWITCH(type, DType, ...) MSHADOW_REAL_TYPE_SWITCH(type, DType, __VA_ARGS__)

// ------------------------------------------------------------------------------------------ nnvm
namespace nnvm {

typedef int64_t dim_t;

template <typename ValueType>
class Tuple {
 public:
  Tuple() {}
  Tuple(std::initializer_list<ValueType> init) : v_(init) {}  // NOLINT
  template <typename It>
  Tuple(It begin, It end) : v_(begin, end) {}
  explicit Tuple(const std::vector<ValueType> &v) : v_(v) {}
  int ndim() const { return static_cast<int>(v_.size()); }
  const ValueType *begin() const { return v_.data(); }
  const ValueType *end() const { return v_.data() + v_.size(); }
  ValueType *begin() { return v_.data(); }
  ValueType *end() { return v_.data() + v_.size(); }
  ValueType &operator[](size_t i) { return v_[i]; }
  const ValueType &operator[](size_t i) const { return v_[i]; }
  bool operator==(const Tuple &o) const { return v_ == o.v_; }
  bool operator!=(const Tuple &o) const { return !(v_ == o.v_); }
  template <typename It>
  void assign(It b, It e) { v_.assign(b, e); }

  friend std::ostream &operator<<(std::ostream &os, const Tuple<ValueType> &t) {
    os << '[';
    for (size_t i = 0; i < t.v_.size(); ++i) os << (i ? "," : "") << t.v_[i];
    return os << ']';
  }
  // accepts "(a, b)", "[a, b]", "a" (python str(tuple) / str(list) / scalar)
  friend std::istream &operator>>(std::istream &is, Tuple<ValueType> &t) {
    std::vector<ValueType> tmp;
    int ch;
    do { ch = is.get(); } while (ch != EOF && isspace(ch));
    if (ch == EOF) { is.setstate(std::ios::failbit); return is; }
    if (ch != '(' && ch != '[') {
      is.unget();
      ValueType one;
      if (is >> one) { tmp.push_back(one); t.v_ = tmp; }
      return is;
    }
    const int close = (ch == '(') ? ')' : ']';
    while (true) {
      do { ch = is.peek(); if (isspace(ch)) is.get(); } while (isspace(ch));
      if (ch == close) { is.get(); break; }
      ValueType idx;
      if (!(is >> idx)) return is;
      tmp.push_back(idx);
      do { ch = is.get(); } while (isspace(ch));
      if (ch == 'L' || ch == 'f') { do { ch = is.get(); } while (isspace(ch)); }
      if (ch == close) break;
      if (ch != ',') { is.setstate(std::ios::failbit); return is; }
    }
    t.v_ = tmp;
    return is;
  }

 protected:
  std::vector<ValueType> v_;
};

struct NodeAttrs {
  std::string name;
  std::unordered_map<std::string, std::string> dict;
  dmlc::any parsed;
};
using dmlc::get;

struct Node;
typedef std::shared_ptr<Node> NodePtr;
struct NodeEntry {
  NodePtr node;
  uint32_t index;
  uint32_t version;
};
struct Node {
  NodeAttrs attrs;
  std::vector<NodeEntry> inputs;
};

typedef std::function<uint32_t(const NodeAttrs &)> FNumVisibleOutputs;
typedef std::function<std::vector<std::string>(const NodeAttrs &)> FListInputNames;
typedef std::function<std::vector<std::string>(const NodeAttrs &)> FListOutputNames;
typedef std::function<bool(const NodeAttrs &, std::vector<int> *, std::vector<int> *)> FInferType;
typedef std::function<std::vector<NodeEntry>(const NodePtr &, const std::vector<NodeEntry> &)> FGradient;
typedef bool TIsBackward;

class Op {
 public:
  std::string name;
  int num_inputs, num_outputs;
  std::function<void(NodeAttrs *)> attr_parser;
  std::map<std::string, dmlc::any> attrs;
  Op() : num_inputs(1), num_outputs(1) {}
  Op &describe(const std::string &) { return *this; }
  Op &set_num_inputs(int n) { num_inputs = n; return *this; }
  Op &set_num_outputs(int n) { num_outputs = n; return *this; }
  Op &set_attr_parser(std::function<void(NodeAttrs *)> fn) { attr_parser = fn; return *this; }
  template <typename ValueType>
  Op &set_attr(const std::string &attr_name, const ValueType &value, int plevel = 10) {
    attrs[attr_name] = dmlc::any(value);
    return *this;
  }
  Op &add_argument(const std::string &, const std::string &, const std::string &) { return *this; }
  Op &add_arguments(const std::vector<dmlc::ParamFieldInfo> &) { return *this; }
  Op &add_alias(const std::string &) { return *this; }
  template <typename ValueType>
  const ValueType *get_attr(const std::string &attr_name) const {
    std::map<std::string, dmlc::any>::const_iterator it = attrs.find(attr_name);
    return it == attrs.end() ? NULL : &it->second.as<ValueType>();
  }
};

}  // namespace nnvm

// ----------------------------------------------------------------------------------------- mxnet
namespace mxnet {

using mshadow::cpu;
using mshadow::gpu;
using mshadow::index_t;
using mshadow::real_t;

class TShape : public nnvm::Tuple<nnvm::dim_t> {
 public:
  TShape() {}
  TShape(std::initializer_list<nnvm::dim_t> init) : nnvm::Tuple<nnvm::dim_t>(init) {}  // NOLINT
  template <typename It>
  TShape(It b, It e) : nnvm::Tuple<nnvm::dim_t>(b, e) {}
  template <int dim>
  TShape(const mshadow::Shape<dim> &s) {  // NOLINT
    v_.assign(s.shape_, s.shape_ + dim);
  }
  template <int dim>
  TShape &operator=(const mshadow::Shape<dim> &s) {
    v_.assign(s.shape_, s.shape_ + dim);
    return *this;
  }
  size_t Size() const {
    size_t s = 1;
    for (size_t i = 0; i < v_.size(); ++i) s *= static_cast<size_t>(v_[i]);
    return s;
  }
  size_t ProdShape(int b, int e) const {
    size_t s = 1;
    for (int i = b; i < e; ++i) s *= static_cast<size_t>(v_[i]);
    return s;
  }
  template <int dim>
  mshadow::Shape<dim> get() const {
    CHECK_EQ(dim, ndim()) << "dimension do not match target dimension " << dim << " vs " << ndim();
    mshadow::Shape<dim> s;
    for (int i = 0; i < dim; ++i) s[i] = static_cast<mshadow::index_t>(v_[i]);
    return s;
  }
  mshadow::Shape<2> FlatTo2D() const {
    if (ndim() == 0) return mshadow::Shape2(0, 0);
    return mshadow::Shape2(static_cast<index_t>(ProdShape(0, ndim() - 1)),
                           static_cast<index_t>(v_[ndim() - 1]));
  }
};
typedef std::vector<TShape> ShapeVector;

struct Context {
  enum DeviceType { kCPU = cpu::kDevMask, kGPU = gpu::kDevMask, kCPUPinned = 3 };
  DeviceType dev_type;
  int32_t dev_id;
  Context() : dev_type(kCPU), dev_id(0) {}
  int dev_mask() const { return dev_type == kCPUPinned ? static_cast<int>(kCPU) : static_cast<int>(dev_type); }
  static Context CPU(int32_t id = 0) { Context c; c.dev_type = kCPU; c.dev_id = id; return c; }
  static Context GPU(int32_t id = 0) { Context c; c.dev_type = kGPU; c.dev_id = id; return c; }
};
struct RunContext {
  Context ctx;
  void *stream;
  RunContext() : stream(NULL) {}
  template <typename xpu> mshadow::Stream<xpu> *get_stream() const { return NULL; }
};

enum OpReqType { kNullOp, kWriteTo, kWriteInplace, kAddTo };

class TBlob {
 public:
  void *dptr_;
  TShape shape_;
  int type_flag_;
  int dev_mask_;
  TBlob() : dptr_(NULL), type_flag_(mshadow::kFloat32), dev_mask_(cpu::kDevMask) {}
  TBlob(void *dptr, const TShape &shape, int dev_mask, int type_flag = mshadow::kFloat32)
      : dptr_(dptr), shape_(shape), type_flag_(type_flag), dev_mask_(dev_mask) {}
  int ndim() const { return shape_.ndim(); }
  index_t size(index_t idx) const { return static_cast<index_t>(shape_[idx]); }
  size_t Size() const { return shape_.Size(); }
  int dev_mask() const { return dev_mask_; }
  bool CheckContiguous() const { return true; }
  template <typename DType>
  DType *dptr() const {
    CHECK(mshadow::DataType<DType>::kFlag == type_flag_)
        << "TBlob.get_with_shape: data type do not match specified type. Expected: " << type_flag_
        << " v.s. given " << mshadow::DataType<DType>::kFlag;
    return static_cast<DType *>(dptr_);
  }
  template <typename Device, int dim, typename DType>
  mshadow::Tensor<Device, dim, DType> get(mshadow::Stream<Device> *stream = NULL) const {
    return mshadow::Tensor<Device, dim, DType>(dptr<DType>(), shape_.get<dim>(), stream);
  }
  template <typename Device, int dim, typename DType>
  mshadow::Tensor<Device, dim, DType> get_with_shape(const mshadow::Shape<dim> &shape,
                                                     mshadow::Stream<Device> *stream = NULL) const {
    CHECK_EQ(shape.Size(), Size()) << "TBlob.get_with_shape: new and old shape do not match total elements";
    return mshadow::Tensor<Device, dim, DType>(dptr<DType>(), shape, stream);
  }
  template <typename Device, typename DType>
  mshadow::Tensor<Device, 1, DType> FlatTo1D(mshadow::Stream<Device> *stream = NULL) const {
    return mshadow::Tensor<Device, 1, DType>(dptr<DType>(), mshadow::Shape1(static_cast<index_t>(Size())),
                                             stream);
  }
  template <typename Device, typename DType>
  mshadow::Tensor<Device, 2, DType> FlatTo2D(mshadow::Stream<Device> *stream = NULL) const {
    return mshadow::Tensor<Device, 2, DType>(dptr<DType>(), shape_.FlatTo2D(), stream);
  }
};

struct ResourceRequest {
  enum Type { kRandom, kTempSpace, kParallelRandom };
  Type type;
  ResourceRequest() : type(kTempSpace) {}
  ResourceRequest(Type t) : type(t) {}  // NOLINT
};

// temp space: like MXNet's, ONE buffer per resource -- a second request returns the same memory
struct Resource {
  struct Buffers {
    std::vector<char> dev, host;
  };
  ResourceRequest req;
  std::shared_ptr<Buffers> buf;
  Resource() : buf(std::make_shared<Buffers>()) {}
  template <typename xpu, int ndim, typename DType>
  mshadow::Tensor<xpu, ndim, DType> get_space_typed(mshadow::Shape<ndim> shape,
                                                    mshadow::Stream<xpu> *stream) const {
    const size_t bytes = shape.Size() * sizeof(DType) + 64;
    if (buf->dev.size() < bytes) buf->dev.resize(bytes);
    return mshadow::Tensor<xpu, ndim, DType>(reinterpret_cast<DType *>(Align(&buf->dev[0])), shape, stream);
  }
  template <typename xpu, int ndim>
  mshadow::Tensor<xpu, ndim, real_t> get_space(mshadow::Shape<ndim> shape,
                                               mshadow::Stream<xpu> *stream) const {
    return get_space_typed<xpu, ndim, real_t>(shape, stream);
  }
  template <int ndim, typename DType>
  mshadow::Tensor<cpu, ndim, DType> get_host_space_typed(mshadow::Shape<ndim> shape) const {
    const size_t bytes = shape.Size() * sizeof(DType) + 64;
    if (buf->host.size() < bytes) buf->host.resize(bytes);
    return mshadow::Tensor<cpu, ndim, DType>(reinterpret_cast<DType *>(Align(&buf->host[0])), shape, NULL);
  }

 private:
  static char *Align(char *p) {
    uintptr_t u = reinterpret_cast<uintptr_t>(p);
    return reinterpret_cast<char *>((u + 63) & ~static_cast<uintptr_t>(63));
  }
};

struct OpContext {
  bool is_train;
  bool need_grad;
  RunContext run_ctx;
  std::vector<Resource> requested;
  OpContext() : is_train(false), need_grad(false) {}
  template <typename xpu>
  mshadow::Stream<xpu> *get_stream() const { return NULL; }
};

class Operator {
 public:
  virtual ~Operator() {}
  virtual void Forward(const OpContext &ctx, const std::vector<TBlob> &in_data,
                       const std::vector<OpReqType> &req, const std::vector<TBlob> &out_data,
                       const std::vector<TBlob> &aux_states) = 0;
  virtual void Backward(const OpContext &ctx, const std::vector<TBlob> &out_grad,
                        const std::vector<TBlob> &in_data, const std::vector<TBlob> &out_data,
                        const std::vector<OpReqType> &req, const std::vector<TBlob> &in_grad,
                        const std::vector<TBlob> &aux_states) {
    LOG(FATAL) << "Backward is not implemented";
  }
};

class OperatorProperty {
 public:
  virtual ~OperatorProperty() {}
  virtual void Init(const std::vector<std::pair<std::string, std::string> > &kwargs) = 0;
  virtual std::map<std::string, std::string> GetParams() const = 0;
  virtual std::vector<std::string> ListArguments() const { return {"data"}; }
  virtual std::vector<std::string> ListOutputs() const { return {"output"}; }
  virtual std::vector<std::string> ListAuxiliaryStates() const { return {}; }
  virtual int NumOutputs() const { return static_cast<int>(this->ListOutputs().size()); }
  virtual int NumVisibleOutputs() const { return NumOutputs(); }
  virtual bool InferShape(std::vector<TShape> *in_shape, std::vector<TShape> *out_shape,
                          std::vector<TShape> *aux_shape) const = 0;
  virtual bool InferType(std::vector<int> *in_type, std::vector<int> *out_type,
                         std::vector<int> *aux_type) const {
    // the default of the legacy interface: every array has the type of the first known input
    int dtype = -1;
    for (size_t i = 0; i < in_type->size(); ++i)
      if ((*in_type)[i] != -1) { dtype = (*in_type)[i]; break; }
    if (dtype == -1) return false;
    for (size_t i = 0; i < in_type->size(); ++i) (*in_type)[i] = dtype;
    out_type->assign(NumOutputs(), dtype);
    aux_type->assign(ListAuxiliaryStates().size(), dtype);
    return true;
  }
  virtual OperatorProperty *Copy() const = 0;
  virtual Operator *CreateOperator(Context ctx) const = 0;
  virtual Operator *CreateOperatorEx(Context ctx, std::vector<TShape> *in_shape,
                                     std::vector<int> *in_type) const {
    return CreateOperator(ctx);
  }
  virtual std::string TypeString() const = 0;
  virtual std::vector<ResourceRequest> ForwardResource(const std::vector<TShape> &in_shape) const {
    return std::vector<ResourceRequest>();
  }
  virtual std::vector<ResourceRequest> BackwardResource(const std::vector<TShape> &in_shape) const {
    return std::vector<ResourceRequest>();
  }
  virtual std::vector<int> DeclareBackwardDependency(const std::vector<int> &out_grad,
                                                     const std::vector<int> &in_data,
                                                     const std::vector<int> &out_data) const {
    std::vector<int> ret = out_grad;
    ret.insert(ret.end(), in_data.begin(), in_data.end());
    ret.insert(ret.end(), out_data.begin(), out_data.end());
    return ret;
  }
  virtual std::vector<std::pair<int, void *> > ForwardInplaceOption(
      const std::vector<int> &in_data, const std::vector<void *> &out_data) const {
    return std::vector<std::pair<int, void *> >();
  }
  virtual std::vector<std::pair<int, void *> > BackwardInplaceOption(
      const std::vector<int> &out_grad, const std::vector<int> &in_data,
      const std::vector<int> &out_data, const std::vector<void *> &in_grad) const {
    return std::vector<std::pair<int, void *> >();
  }
};

typedef std::function<bool(const nnvm::NodeAttrs &, ShapeVector *, ShapeVector *)> FInferShape;
typedef std::function<void(const nnvm::NodeAttrs &, const OpContext &, const std::vector<TBlob> &,
                           const std::vector<OpReqType> &, const std::vector<TBlob> &)>
    FCompute;

}  // namespace mxnet

// the registry lives in runtime.cc (one copy per libref_<op>.so)
namespace mxshim {
struct PropEntry {
  std::string name;
  std::function<mxnet::OperatorProperty *()> body;
  PropEntry &describe(const std::string &) { return *this; }
  PropEntry &add_argument(const std::string &, const std::string &, const std::string &) { return *this; }
  PropEntry &add_arguments(const std::vector<dmlc::ParamFieldInfo> &) { return *this; }
  PropEntry &set_return_type(const std::string &) { return *this; }
  PropEntry &add_alias(const std::string &) { return *this; }
  PropEntry &set_key_var_num_args(const std::string &) { return *this; }
};
PropEntry &RegisterProp(const char *name, std::function<mxnet::OperatorProperty *()> body);
nnvm::Op &RegisterOp(const char *name);
}  // namespace mxshim

#define MXNET_REGISTER_OP_PROPERTY(name, OperatorPropertyType)                            \
  static ::mxshim::PropEntry &MXSHIM_CAT(__mxshim_prop_, __COUNTER__) __attribute__((unused)) = \
      ::mxshim::RegisterProp(#name, []() -> ::mxnet::OperatorProperty * { return new OperatorPropertyType(); })
#define NNVM_REGISTER_OP(OpName) \
  static ::nnvm::Op &MXSHIM_CAT(__mxshim_op_, __COUNTER__) __attribute__((unused)) = ::mxshim::RegisterOp(#OpName)

#define DO_BIND_DISPATCH(Method, ...)               \
  if (ctx.dev_mask() == ::mxnet::cpu::kDevMask) {   \
    return Method<::mxnet::cpu>(__VA_ARGS__);       \
  } else {                                          \
    return Method<::mxnet::gpu>(__VA_ARGS__);       \
  }

#define Assign(out, req, exp)           \
  {                                     \
    switch (req) {                      \
      case ::mxnet::kNullOp: break;     \
      case ::mxnet::kWriteTo:           \
      case ::mxnet::kWriteInplace: (out) = (exp); break; \
      case ::mxnet::kAddTo: LOG(FATAL) << "mxshim: kAddTo Assign not provided"; break; \
      default: LOG(FATAL) << "not reached"; \
    }                                   \
  }

#define SHAPE_ASSIGN_CHECK(shape_array, index, shape) \
  { (shape_array)[index] = ::mxnet::TShape(shape); }
#define TYPE_ASSIGN_CHECK(type_array, index, type) \
  { (type_array)[index] = (type); }

namespace mxnet {
namespace op {

using mshadow::cpu;
using mshadow::gpu;
using mshadow::index_t;
using mshadow::real_t;
using nnvm::NodeAttrs;
using mxnet::FCompute;

// operator/mshadow_op.h -- the four functors the detection ops use.  MXNet 1.6 semantics:
// minimum/maximum return `a` when it is NaN, otherwise the plain comparison; floor/ceil are the
// libm functions of the operand type.
namespace mshadow_op {
struct minimum {
  template <typename DType>
  MSHADOW_XINLINE static DType Map(DType a, DType b) {
    if (a != a) return a;
    return a < b ? a : b;
  }
};
struct maximum {
  template <typename DType>
  MSHADOW_XINLINE static DType Map(DType a, DType b) {
    if (a != a) return a;
    return a > b ? a : b;
  }
};
struct floor {
  MSHADOW_XINLINE static float Map(float a) { return ::floorf(a); }
  MSHADOW_XINLINE static double Map(double a) { return ::floor(a); }
};
struct ceil {
  MSHADOW_XINLINE static float Map(float a) { return ::ceilf(a); }
  MSHADOW_XINLINE static double Map(double a) { return ::ceil(a); }
};
}  // namespace mshadow_op

namespace mxnet_op {
// Kernel<OP, xpu>::Launch(s, N, args...) == for i in [0, N): OP::Map(i, args...)
// (OpenMP on the CPU, one CUDA thread per i on the GPU; every Map used here is independent per i
//  except the GPU backward's atomicAdd, which cuemu.h turns into a plain serial +=.)
template <typename OP, typename xpu>
struct Kernel {
  template <typename... Args>
  static void Launch(mshadow::Stream<xpu> *, const int N, Args... args) {
    for (int i = 0; i < N; ++i) OP::Map(i, args...);
  }
};
}  // namespace mxnet_op

template <bool is_integer, typename ValueType, typename xpu>
void Fill(mshadow::Stream<xpu> *s, const TBlob &b, const OpReqType req, ValueType val) {
  if (req == kNullOp) return;
  CHECK_NE(req, kAddTo) << "mxshim: Fill with kAddTo";
  MSHADOW_REAL_TYPE_SWITCH(b.type_flag_, DType, {
    DType *p = b.dptr<DType>();
    const size_t n = b.Size();
    for (size_t i = 0; i < n; ++i) p[i] = static_cast<DType>(val);
  });
}

template <typename PType>
inline void ParamParser(nnvm::NodeAttrs *attrs) {
  PType param;
  param.Init(attrs->dict);
  attrs->parsed = dmlc::any(param);
}

// FGradient helper: records which entries the backward node receives
inline std::vector<nnvm::NodeEntry> MakeGradNode(const char *op_name, const nnvm::NodePtr &n,
                                                 const std::vector<nnvm::NodeEntry> &inputs,
                                                 const std::unordered_map<std::string, std::string> &dict) {
  nnvm::NodePtr p = std::make_shared<nnvm::Node>();
  p->attrs.name = op_name;
  p->attrs.dict = dict;
  p->inputs = inputs;
  std::vector<nnvm::NodeEntry> ret;
  for (uint32_t i = 0; i < n->inputs.size(); ++i) {
    nnvm::NodeEntry e;
    e.node = p; e.index = i; e.version = 0;
    ret.push_back(e);
  }
  return ret;
}

}  // namespace op
}  // namespace mxnet

#endif  // ORACLE_MXSHIM_H_
