// Device helpers of the experimental split-precision path: exact 3-way bf16 split of fp32 values
// (x = hi + mid + lo, three 8-bit significands, round-to-nearest conversions + exact fp32 subtraction).
#pragma once
#include "set_common.h"

namespace set {

typedef float split_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned split_u32x2 __attribute__((ext_vector_type(2)));

typedef float split_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 split_bf16x2 __attribute__((ext_vector_type(2)));

// one pair of values: hi = bf16(x) (round to nearest even, v_cvt_pk_bf16_f32), r1 = x - hi (exact in fp32, packed
// subtract), mid = bf16(r1), r2 = r1 - mid (exact), lo = bf16(r2) (r2 has <= 9 significant bits: exact or off by
// <= 2^-26 |x|).  4.5 VALU operations per value, and the packed results are the LDS plane words directly.
__device__ __forceinline__ void split3_pair(const split_f32x2 x, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, split_bf16x2));
    const split_f32x2 hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
    const split_f32x2 r1 = x - hf;
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, split_bf16x2));
    const split_f32x2 mf = {__uint_as_float(mid << 16), __uint_as_float(mid & 0xffff0000u)};
    const split_f32x2 r2 = r1 - mf;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, split_bf16x2));
}

__device__ __forceinline__ void split3(const split_f32x4 x, split_u32x2& hi, split_u32x2& mid, split_u32x2& lo) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3_pair((split_f32x2){x[0], x[1]}, h0, m0, l0);
    split3_pair((split_f32x2){x[2], x[3]}, h1, m1, l1);
    hi = (split_u32x2){h0, h1};
    mid = (split_u32x2){m0, m1};
    lo = (split_u32x2){l0, l1};
}

}  // namespace set
