// Device helpers of the experimental split-precision path: exact 3-way bf16 split of fp32 values
// (x = hi + mid + lo, 8 + 8 + 8 significand bits, truncation + exact fp32 subtraction).
#pragma once
#include "set_common.h"

namespace set {

typedef float split_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned split_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3(const split_f32x4 x, split_u32x2& hi, split_u32x2& mid, split_u32x2& lo) {
    unsigned u[4], v[4], w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u[e] = __float_as_uint(x[e]);
        const float r1 = x[e] - __uint_as_float(u[e] & 0xffff0000u);
        v[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(v[e] & 0xffff0000u);
        w[e] = __float_as_uint(r2);
    }
    // pack the upper halves: element 0 in the low 16 bits
    hi = (split_u32x2){__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u)};
    mid = (split_u32x2){__builtin_amdgcn_perm(v[1], v[0], 0x07060302u), __builtin_amdgcn_perm(v[3], v[2], 0x07060302u)};
    lo = (split_u32x2){__builtin_amdgcn_perm(w[1], w[0], 0x07060302u), __builtin_amdgcn_perm(w[3], w[2], 0x07060302u)};
}

}  // namespace set
