// Shared host/device helpers for libspi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spi_hip.h"

void spi_set_error(const char* fmt, ...);

#define SPI_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) {                                           \
            spi_set_error(__VA_ARGS__);                          \
            return SPI_ERR_BAD_ARG;                              \
        }                                                        \
    } while (0)

#define SPI_LAUNCH_CHECK(name)                                                          \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            spi_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));       \
            return SPI_ERR_LAUNCH;                                                      \
        }                                                                               \
    } while (0)

static inline hipStream_t as_stream(spi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
constexpr int WAVE = 64;

__device__ __forceinline__ float softplus_f(float x) {          // torch Softplus(beta=1, threshold=20)
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// Hardware-transcendental versions (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each) for the decoder
// MLP, where 96 activations per point would otherwise cost more issue slots than the 4160 FMAs.
// Absolute error <= ~2e-7 on outputs that are O(1); parity tests bound the end-to-end effect.
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
// Raw buffer loads: an out-of-range byte offset returns 0 in hardware, so bounds-checked gathers need no branch.
// (A predicated global load `ok ? p[i] : 0` becomes an exec-masked region that hipcc closes with s_waitcnt vmcnt(0):
// a row of such loads is fully serialised -- the 4x4 FIR ran at 0.8 TB/s because of it.)
constexpr unsigned BUF_OOB = 0x80000000u;        // beyond any num_records we create (< 2 GiB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, 0));
}

__device__ __forceinline__ float softplus_fast(float x) {
    return x > 20.f ? x : __builtin_amdgcn_logf(1.f + exp_fast(x)) * 0.6931471805599453f;
}
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + exp_fast(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}
// inclusive product / sum scans across the 64 lanes of a wave
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        float t = __shfl_up(v, o, WAVE);
        if (lane >= o) v *= t;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        float t = __shfl_up(v, o, WAVE);
        if (lane >= o) v += t;
    }
    return v;
}
#endif
