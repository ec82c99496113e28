// The hi / lo operand split of the fp32-in / fp32-out matrix-core products (GEMM and fused forward).
#pragma once
#include "deform_common.h"

namespace sd {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// ---- split arithmetic ------------------------------------------------------------------------------
// kSplitBF16: hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits kept, any fp32 magnitude.  Error of a
//   K = 2304 product sum ~4.5e-6 x max|C| (nine times the fp32 MFMA path's).
// kSplitF16 (default for the DCN layer): the operand is first scaled by a power of two s so that
//   max|x| * s lies in [2^13, 2^14) (s from a max|x| pre-pass over the operand -- or an upper bound of
//   it), then hi = f16(x * s), lo = f16(x * s - hi): 22 mantissa bits kept wherever |x| >= 2^-17 max|x|,
//   an absolute error below 2^-38 max|x| elsewhere (fp16 subnormals).  hi*hi products are exact in
//   the fp32 accumulator; the dropped lo*lo term is <= 2^-22 of a product.  The accumulator is scaled
//   back by 1/s_a and 1/s_b (exact) on the way out.  Measured: the error of the DCN products against
//   fp64 is that of the fp32 MFMA path (tests/test_deform_conv.py).
constexpr int kSplitBF16 = 1, kSplitF16 = 2;
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

// power-of-two scale of an operand from the bit pattern of (an upper bound of) its max|x|, and its
// inverse; zero, inf and nan maxima scale by 1
__device__ __forceinline__ void f16_split_scale(unsigned max_bits, float& s, float& inv) {
  const int e = (int)((max_bits >> 23) & 255);
  int es = 267 - e;  // biased exponent of s = 2^(13 - (e - 127))
  if ((max_bits & 0x7fffffffu) == 0u || e == 255) es = 127;
  es = es > 254 ? 254 : es;
  s = __uint_as_float((unsigned)es << 23);
  inv = 1.0f / s;  // exact: a power of two within the normal / subnormal range
}

// hi / lo parts of two floats, packed (element 0 in the low half).  PK = false keeps the two
// subtractions scalar: a packed v_pk_add_f32 wants its operands in adjacent registers, and for
// values that come out of two different loads the compiler then shuffles registers right behind
// the loads -- i.e. waits for the prefetch it was supposed to leave in flight.
template <bool PK, int MODE>
__device__ __forceinline__ void split2(float x0, float x1, float scale, unsigned& hi, unsigned& lo) {
  float f0, f1;
  if (MODE == kSplitF16) {
    x0 *= scale;
    x1 *= scale;
    const floatx2 v = {x0, x1};
    const halfx2 h = __builtin_convertvector(v, halfx2);
    hi = __builtin_bit_cast(unsigned, h);
    const floatx2 back = __builtin_convertvector(h, floatx2);
    f0 = back.x;
    f1 = back.y;
  } else {
    const floatx2 v = {x0, x1};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    f0 = __builtin_bit_cast(float, hi << 16);
    f1 = __builtin_bit_cast(float, hi & 0xffff0000u);
  }
  floatx2 r;
  if (PK) {
    r = floatx2{x0 - f0, x1 - f1};
  } else {
    float r0, r1;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(f0));
    asm("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(f1));
    r = floatx2{r0, r1};
  }
  if (MODE == kSplitF16) lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, halfx2));
  else lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

// one 32x32x16 matrix-core step on packed 16-bit operands of either kind
template <int MODE>
__device__ __forceinline__ floatx16 mfma16(uint4 a, uint4 b, floatx16 c) {
  if (MODE == kSplitF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(halfx8, a), __builtin_bit_cast(halfx8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace sd
