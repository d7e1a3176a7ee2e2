// Shared device helpers for libstainx_hip (gfx950 only: wave64, no portability layer).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/stainx_hip.h"

namespace sx {

constexpr int kWave = 64;
constexpr int kStreamThreads = 256;   // streaming kernels: 4 waves per workgroup

// ---- error reporting (thread-local, never throws) ---------------------------------------------
char* err_buf();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

// ---- element types ----------------------------------------------------------------------------
// Input pixels become float "unit" values exactly as the reference's dtype gate does
// (torch_backend.py:104-113): u8 -> float(u)/255 (correctly rounded division), floats -> float(x).
// x / 255.0f for 0 <= x <= 256, bit for bit the IEEE division (torch's `/ 255` on the float32 value), in three instructions
// instead of the division's ten: q = x * c, the exact residual r = fma(-q, 255, x), q + r * c.  Checked on the device for EVERY
// float32 in [0, 256] (1 132 462 081 values: all grey levels, all bf16 / f16 values of that range included) by
// tools/check_div255.hip; outside that range (infinities) it is not the quotient: callers clamp first.
__device__ __forceinline__ float div255_of_level(float x) {
    const float c = 1.0f / 255.0f;
    const float q = x * c;
    return fmaf(fmaf(-q, 255.0f, x), c, q);
}

template <typename T> struct Elem;
template <> struct Elem<uint8_t> {
    static __device__ __forceinline__ float load(uint8_t v) { return div255_of_level((float)v); }
    static __device__ __forceinline__ uint8_t store(float v) { return (uint8_t)v; }  // trunc, like .to(uint8)
};
template <> struct Elem<__half> {
    static __device__ __forceinline__ float load(__half v) { return __half2float(v); }
    static __device__ __forceinline__ __half store(float v) { return __float2half(v); }  // RNE
};
template <> struct Elem<__hip_bfloat16> {
    static __device__ __forceinline__ float load(__hip_bfloat16 v) { return __bfloat162float(v); }
    static __device__ __forceinline__ __hip_bfloat16 store(float v) { return __float2bfloat16(v); }  // RNE
};
template <> struct Elem<float> {
    static __device__ __forceinline__ float load(float v) { return v; }
    static __device__ __forceinline__ float store(float v) { return v; }
};
template <> struct Elem<double> {
    static __device__ __forceinline__ float load(double v) { return (float)v; }   // .float(): RNE
    static __device__ __forceinline__ double store(float v) { return (double)v; }
};

// A naturally aligned pack of V elements; the streaming kernels use 16-byte packs
// (f32 x4, bf16/f16 x8, u8 x16, f64 x2) or single elements when the tile is not pack-aligned.
template <typename T, int V> struct alignas(sizeof(T) * V) Pack { T v[V]; };

template <typename T, int V>
__device__ __forceinline__ void load_unit(const T* __restrict__ p, float (&out)[V]) {
    if constexpr (V == 1) {
        out[0] = Elem<T>::load(p[0]);
    } else {
        const Pack<T, V> pk = *reinterpret_cast<const Pack<T, V>*>(p);
#pragma unroll
        for (int i = 0; i < V; ++i) out[i] = Elem<T>::load(pk.v[i]);
    }
}

// Same pack, but uint8 pixels stay integer-valued floats 0..255 (callers that fold the /255 into their own
// arithmetic, like the optical density of the Macenko path).
template <typename T> __device__ __forceinline__ float raw_value(T v) {
    if constexpr (sizeof(T) == 1) return (float)v; else return Elem<T>::load(v);
}
template <typename T, int V>
__device__ __forceinline__ void load_raw(const T* __restrict__ p, float (&out)[V]) {
    if constexpr (V == 1) {
        out[0] = raw_value<T>(p[0]);
    } else {
        const Pack<T, V> pk = *reinterpret_cast<const Pack<T, V>*>(p);
#pragma unroll
        for (int i = 0; i < V; ++i) out[i] = raw_value<T>(pk.v[i]);
    }
}

// A pack read for the LAST time in a call (the apply / reconstruct passes): loaded non-temporally where it is 16 bytes, so that it
// leaves no lines in the L2s for the pass's own stores to push out (Macenko reconstruct 65.3 -> 61.5 us on the config-2 batch).
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> load_pack_stream(const T* __restrict__ p) {
    Pack<T, V> pk;
    if constexpr (sizeof(T) * V == 16) {
        typedef uint32_t u4v __attribute__((ext_vector_type(4)));
        const u4v q = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p));
        __builtin_memcpy(&pk, &q, 16);
    } else {
        pk = *reinterpret_cast<const Pack<T, V>*>(p);
    }
    return pk;
}

template <typename T, int V>
__device__ __forceinline__ void store_pack(T* __restrict__ p, const T (&vals)[V]) {
    if constexpr (V == 1) {
        p[0] = vals[0];
    } else {
        Pack<T, V> pk;
#pragma unroll
        for (int i = 0; i < V; ++i) pk.v[i] = vals[i];
        *reinterpret_cast<Pack<T, V>*>(p) = pk;
    }
}

// Streaming (non-temporal) forms for data touched once per launch: the output of a transform is not read again
// by this library, and a non-temporal store does not leave a dirty line behind for the next reader to wait on.
template <typename T, int V>
__device__ __forceinline__ void store_pack_stream(T* __restrict__ p, const T (&vals)[V]) {
    constexpr int kBytes = (int)sizeof(T) * V;
    if constexpr (kBytes % 16 == 0) {
        typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int piece = 0; piece < kBytes / 16; ++piece) {
            f4 v;
            __builtin_memcpy(&v, reinterpret_cast<const char*>(vals) + 16 * piece, 16);
            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(reinterpret_cast<char*>(p) + 16 * piece));
        }
    } else if constexpr (kBytes == 8) {
        unsigned long long v;
        __builtin_memcpy(&v, vals, 8);
        __builtin_nontemporal_store(v, reinterpret_cast<unsigned long long*>(p));
    } else if constexpr (kBytes == 4) {
        unsigned v;
        __builtin_memcpy(&v, vals, 4);
        __builtin_nontemporal_store(v, reinterpret_cast<unsigned*>(p));
    } else {
        store_pack<T, V>(p, vals);
    }
}

// ---- order-preserving uint32 image of a float (radix / bracket selection keys) ------------------
__device__ __forceinline__ uint32_t float_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);      // negative: ~u, positive: u | sign bit -- three ops, no select
}
__device__ __forceinline__ float key_float(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// ---- wave64 / workgroup reductions ---------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;   // lane 0 holds the sum
}
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// ---- wave64 scans / reductions on the DPP path -----------------------------------------------------
// ds_bpermute shuffles cost an LDS round trip each (~100 cycles, six of them in a dependent chain per
// reduction); DPP row shifts + the two gfx9 row broadcasts stay in the VALU.  Pattern: Hillis-Steele inside
// each row of 16 (row_shr 1,2,4,8), row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3; lanes
// without a source keep `identity`.  After it every lane holds the inclusive scan, lane 63 the total.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ uint32_t dpp_move(uint32_t identity, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)src, kCtrl, kRowMask, 0xF, false);
}
#define SX_DPP_SCAN(x, identity, OP)                               \
    x = OP(x, dpp_move<0x111, 0xF>(identity, x)); /* row_shr:1 */  \
    x = OP(x, dpp_move<0x112, 0xF>(identity, x)); /* row_shr:2 */  \
    x = OP(x, dpp_move<0x114, 0xF>(identity, x)); /* row_shr:4 */  \
    x = OP(x, dpp_move<0x118, 0xF>(identity, x)); /* row_shr:8 */  \
    x = OP(x, dpp_move<0x142, 0xA>(identity, x)); /* row_bcast:15 */ \
    x = OP(x, dpp_move<0x143, 0xC>(identity, x)); /* row_bcast:31 */
__device__ __forceinline__ uint32_t op_add_u32(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t op_min_u32(uint32_t a, uint32_t b) { return min(a, b); }
__device__ __forceinline__ uint32_t op_max_u32(uint32_t a, uint32_t b) { return max(a, b); }
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t x) {       // inclusive prefix sum
    SX_DPP_SCAN(x, 0u, op_add_u32)
    return x;
}
__device__ __forceinline__ uint32_t wave_total_u32(uint32_t x) {      // sum, uniform in every lane
    SX_DPP_SCAN(x, 0u, op_add_u32)
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
    SX_DPP_SCAN(x, 0xFFFFFFFFu, op_min_u32)
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
    SX_DPP_SCAN(x, 0u, op_max_u32)
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// doubles go through two 32-bit moves
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_move_f64(double src) {          // identity 0.0
    const unsigned long long u = (unsigned long long)__double_as_longlong(src);
    const uint32_t lo = dpp_move<kCtrl, kRowMask>(0u, (uint32_t)u), hi = dpp_move<kCtrl, kRowMask>(0u, (uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_total_f64(double x) {          // fixed order -> deterministic; total in lane 63 only
    x += dpp_move_f64<0x111, 0xF>(x);
    x += dpp_move_f64<0x112, 0xF>(x);
    x += dpp_move_f64<0x114, 0xF>(x);
    x += dpp_move_f64<0x118, 0xF>(x);
    x += dpp_move_f64<0x142, 0xA>(x);
    x += dpp_move_f64<0x143, 0xC>(x);
    return x;
}

// Position of this lane among the set lanes of `mask` below it.
__device__ __forceinline__ uint32_t rank_in_mask(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

}  // namespace sx
