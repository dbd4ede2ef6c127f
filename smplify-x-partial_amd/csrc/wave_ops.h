// wave_ops.h -- wavefront (64-lane) reductions on DPP instead of ds_bpermute shuffles.
// A __shfl_xor butterfly costs six dependent LDS-crossbar round trips (~600 cycles); the
// DPP scan below is six VALU adds + one v_readlane (~80 cycles) and has a fixed summation
// order, so results are deterministic and independent of batch composition.
#pragma once
#include <hip/hip_runtime.h>

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_shift_or_zero(float x) {
    // lanes with no valid source (or rows masked off) receive +0.0f
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}

// sum over the 64 lanes, returned in every lane (via SGPR broadcast of lane 63)
__device__ __forceinline__ float wave_sum_dpp(float x) {
    x += dpp_shift_or_zero<0x111, 0xf>(x);   // row_shr:1
    x += dpp_shift_or_zero<0x112, 0xf>(x);   // row_shr:2
    x += dpp_shift_or_zero<0x114, 0xf>(x);   // row_shr:4
    x += dpp_shift_or_zero<0x118, 0xf>(x);   // row_shr:8   -> lane 15 of each row = row total
    x += dpp_shift_or_zero<0x142, 0xa>(x);   // row_bcast:15 into rows 1,3
    x += dpp_shift_or_zero<0x143, 0xc>(x);   // row_bcast:31 into rows 2,3 -> lane 63 = total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_shift_or_self(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ float wave_max_dpp(float x) {
    x = fmaxf(x, dpp_shift_or_self<0x111, 0xf>(x));
    x = fmaxf(x, dpp_shift_or_self<0x112, 0xf>(x));
    x = fmaxf(x, dpp_shift_or_self<0x114, 0xf>(x));
    x = fmaxf(x, dpp_shift_or_self<0x118, 0xf>(x));
    x = fmaxf(x, dpp_shift_or_self<0x142, 0xa>(x));
    x = fmaxf(x, dpp_shift_or_self<0x143, 0xc>(x));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// inclusive prefix sum over the 64 lanes (integers): Hillis-Steele inside the rows of 16 on DPP shifts, then the row totals
// travel with row_bcast -- six VALU adds where a __shfl_up ladder costs six LDS-crossbar round trips
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_shift_or_zero_i(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_incl_scan_dpp(int x) {
    x += dpp_shift_or_zero_i<0x111, 0xf>(x);   // row_shr:1
    x += dpp_shift_or_zero_i<0x112, 0xf>(x);   // row_shr:2
    x += dpp_shift_or_zero_i<0x114, 0xf>(x);   // row_shr:4
    x += dpp_shift_or_zero_i<0x118, 0xf>(x);   // row_shr:8   -> inclusive within each row of 16
    x += dpp_shift_or_zero_i<0x142, 0xa>(x);   // row_bcast:15 into rows 1, 3
    x += dpp_shift_or_zero_i<0x143, 0xc>(x);   // row_bcast:31 into rows 2, 3
    return x;
}
