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

// fused forward epilogue of the convolutions: y = clamp(act(acc + noise * noise_gain + bias) * gain)
struct Epilogue { const float* bias; const float* noise; const float* noise_gain; int act; float alpha, gain, clamp; };
typedef Epilogue WinoEpilogue;

// Winograd F(2x2, 3x3) problem (winograd.hip): stride 1, pad 1, output H x W
struct WinoParams {
    int N, nw;                    // batch; number of weight sets (N with per-sample weights, else 1)
    int Mo, Ci, H, W;
    int bx, by;                   // 16 x 16-pixel blocks per row / column
    int ocp;                      // Mo rounded up to 64 (rows of the transformed weights)
    int64_t in_bs, out_bs, wbs, u_bs;
    int wsm, wsc, widx[9];        // weight addressing: w[n*wbs + m*wsm + c*wsc + widx[(dy+1)*3 + (dx+1)]]
    const int32_t* seg_flags;     // dgrad: zero-segment map of the gradient operand (or NULL)
    const int32_t* out_flags;     // forward: needed-output map (or NULL)
    int nseg;
    int ksplit;                   // > 1: the channel reduction is cut into ksplit ranges of slabs (blockIdx.z = n * ksplit + range); partial outputs are
                                  //      atomically added into a ZEROED out and the epilogue runs afterwards (small layers: too few blocks to fill 256 CUs)
    int f4;                       // 1: F(4x4, 3x3) (wino4_conv_kernel: 36 frequencies, 16 x 32-pixel blocks) instead of F(2x2, 3x3); set by the dispatcher
    int64_t u_bs_of() const { return (int64_t)(f4 ? 36 : 16) * Ci * ocp; }      // floats of one transformed weight set
};
int64_t spi_wino_workspace_bytes(const WinoParams& P) __attribute__((visibility("hidden")));
int spi_wino_launch(WinoParams P, const float* in, const float* w, float* out, const Epilogue& ep, void* workspace, hipStream_t st, bool u_ready = false)
    __attribute__((visibility("hidden")));
// F(3x3, 2x2) weight gradient of the same problem (x [N,Ci,H,W], dy [N,Mo,H,W]) added into a zeroed dw
int spi_wino_wgrad_launch(const WinoParams& P, const float* x, const float* dy, float* dw, hipStream_t st) __attribute__((visibility("hidden")));

// Direct fp16 3x3 convolution of fp16 activation tensors (hconv.hip): forward / dgrad of a stride-1, pad-1 conv whose output-channel count is a multiple
// of 128 and whose reduction channels come in whole 16-channel chunks.  The workspace receives the fp16 LDS image of the weights (`img_ready`: it
// already holds it -- frozen weights).
bool spi_hconv_eligible(const WinoParams& P) __attribute__((visibility("hidden")));
int64_t spi_hconv_workspace_bytes(const WinoParams& P) __attribute__((visibility("hidden")));
int spi_hconv_launch(const WinoParams& P, const void* in, const float* w, void* out, const Epilogue& ep, void* workspace, hipStream_t st, bool img_ready = false)
    __attribute__((visibility("hidden")));

// ... and of a STRIDE-2 3x3 conv (the data gradient of a stride-2 transposed conv; P = that dgrad problem, IH x IW = its gradient operand)
bool spi_hconv_s2_eligible(const WinoParams& P) __attribute__((visibility("hidden")));
int spi_hconv_s2_launch(const WinoParams& P, int IH, int IW, const void* in, const float* w, void* out, void* workspace, hipStream_t st, bool img_ready = false)
    __attribute__((visibility("hidden")));
// ... and the FORWARD of a stride-2 transposed 3x3 conv (P.H x P.W = input, output 2H+1 x 2W+1; both row parities, two launches)
bool spi_hconv_t2_eligible(const WinoParams& P) __attribute__((visibility("hidden")));
int spi_hconv_t2_launch(const WinoParams& P, const void* in, const float* w, void* out, void* workspace, hipStream_t st, bool img_ready = false)
    __attribute__((visibility("hidden")));
// Direct fp16 weight gradient of the same layers (hconv.hip): x [N,Ci,H,W] / dy [N,Mo,H,W] fp16 tensors, dw fp32 and zeroed
bool spi_hwgrad_eligible(const WinoParams& P) __attribute__((visibility("hidden")));
int64_t spi_hwgrad_workspace_bytes(const WinoParams& P) __attribute__((visibility("hidden")));     // partial-sum buffer that replaces the atomics (optional)
int spi_hwgrad_launch(const WinoParams& P, const void* x, const void* dy, float* dw, void* workspace, int64_t workspace_bytes, hipStream_t st)
    __attribute__((visibility("hidden")));

// Zero `n_floats` floats on `st` with a kernel.  Not hipMemsetAsync: as a node of a captured HIP graph a memset whose byte count is not a
// multiple of 16 (the decoder's 33-float bias gradient) leaves garbage behind on replays (ROCm 7.0; tools/ubench/graph_memset.py),
// and the loops replay their steps from graphs.
int spi_zero_async(float* p, int64_t n_floats, hipStream_t st) __attribute__((visibility("hidden")));

static inline hipStream_t as_stream(spi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
constexpr int WAVE = 64;

__device__ __forceinline__ float softplus_f(float x) {          // torch Softplus(beta=1, threshold=20)
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// The conv epilogues' activation (linear / relu / lrelu), gain and clamp -- ONE definition for the implicit GEMM, the Winograd border
// blocks and the Winograd interior fast path.  Written with selects so that NaN and +-inf behave like torch's relu / leaky_relu /
// clamp on the reference's CPU path (NaN propagates; fmaxf / fminf would turn it into 0 or +-clamp and hide a diverging run).
__device__ __forceinline__ float conv_act_gain_clamp(int act, float alpha, float gain, float clamp, float v) {
    if (act == SPI_ACT_RELU) v = v < 0.f ? 0.f : v;
    else if (act == SPI_ACT_LRELU) v = v > 0.f ? v : v * alpha;
    v *= gain;
    if (clamp >= 0.f) v = v > clamp ? clamp : (v < -clamp ? -clamp : v);
    return v;
}

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
// Four 16-byte buffer loads in flight, one wait.  (hipcc 7.2 miscompiles __builtin_amdgcn_raw_buffer_load_b64 / _b128 into a
// single buffer_load_dword -- only the first component arrives -- hence the inline assembly; the compiler does not track
// vmcnt for assembly, so the wait has to sit inside the block.)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// twelve loads (the 3 x 4 bilinear corners of a tri-plane sample), one wait
__device__ __forceinline__ void buf_load12_f32x4(__amdgpu_buffer_rsrc_t rs, const unsigned (&o)[12], f32x4_t (&v)[12]) {
    asm volatile("buffer_load_dwordx4 %0, %12, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %1, %13, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %2, %14, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %3, %15, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %4, %16, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %5, %17, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %6, %18, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %7, %19, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %8, %20, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %9, %21, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %10, %22, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %11, %23, %24, 0 offen\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
                   "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11])
                 : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]), "v"(o[6]), "v"(o[7]), "v"(o[8]), "v"(o[9]), "v"(o[10]),
                   "v"(o[11]), "s"(rs)
                 : "memory");
}
// The same twelve loads WITHOUT the wait, and the wait as its own block: whatever the compiler issues in between (the next pass's coordinate
// loads in gather_tile) is in flight together with the gathers.  The wait is vmcnt(0) -- loads complete in order, so it covers both -- and
// takes the twelve destination registers as read-write operands so that nothing uses them before it.  (The compiler's own waitcnt
// bookkeeping does not see the assembly's loads; its waits for its own, younger loads are therefore conservative, never too short.)
__device__ __forceinline__ void buf_load12_f32x4_nowait(__amdgpu_buffer_rsrc_t rs, const unsigned (&o)[12], f32x4_t (&v)[12]) {
    asm volatile("buffer_load_dwordx4 %0, %12, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %1, %13, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %2, %14, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %3, %15, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %4, %16, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %5, %17, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %6, %18, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %7, %19, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %8, %20, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %9, %21, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %10, %22, %24, 0 offen\n\t"
                 "buffer_load_dwordx4 %11, %23, %24, 0 offen"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
                   "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11])
                 : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]), "v"(o[6]), "v"(o[7]), "v"(o[8]), "v"(o[9]), "v"(o[10]),
                   "v"(o[11]), "s"(rs)
                 : "memory");
}
__device__ __forceinline__ void buf_wait_gathers(f32x4_t (&v)[12]) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                   "+v"(v[10]), "+v"(v[11])
                 :: "memory");
}
__device__ __forceinline__ void buf_load4_f32x4(__amdgpu_buffer_rsrc_t rs, unsigned o0, unsigned o1, unsigned o2, unsigned o3,
                                               f32x4_t& v0, f32x4_t& v1, f32x4_t& v2, f32x4_t& v3) {
    asm volatile("buffer_load_dwordx4 %0, %4, %8, 0 offen\n\t"
                 "buffer_load_dwordx4 %1, %5, %8, 0 offen\n\t"
                 "buffer_load_dwordx4 %2, %6, %8, 0 offen\n\t"
                 "buffer_load_dwordx4 %3, %7, %8, 0 offen\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                 : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(rs)
                 : "memory");
}
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)byte_off, 0, 0));
}

__device__ __forceinline__ float softplus_fast(float x) {
    return x > 20.f ? x : __builtin_amdgcn_logf(1.f + exp_fast(x)) * 0.6931471805599453f;
}
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + exp_fast(-x)); }

// Cross-lane arithmetic on the DPP path (data-parallel primitives: the lane permutation is a modifier of the VALU instruction itself,
// ~1 issue slot), not through `__shfl*` -- hipcc lowers those to `ds_bpermute_b32`, a round trip through the LDS crossbar of >= 64 cycles
// of latency each, and the scans / reductions below are dependent chains of six of them.  CTRL: quad_perm 0x00-0xff, row_shr:n 0x110+n,
// row_ror:n 0x120+n, wave_shr:1 0x138, row_mirror 0x140, row_half_mirror 0x141, row_bcast:15 0x142, row_bcast:31 0x143 (GFX9 encodings).
// Lanes without a source lane, and rows / banks masked out, receive `old`.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float lane_bcast(float v, int src_lane) {          // value of one lane, as a wave-uniform (SGPR) operand
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}
// sum over each aligned group of 8 lanes, in every lane of the group
__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_f32<0xB1>(0.f, v);            // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(0.f, v);            // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(0.f, v);           // row_half_mirror: the other quad of the 8
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = group8_sum(v);
    v += dpp_f32<0x140>(0.f, v);           // row_mirror: row (16-lane) totals in every lane
    v += dpp_f32<0x142, 0xa>(0.f, v);      // rows 1, 3 += lane 15 of rows 0, 2
    v += dpp_f32<0x143, 0xc>(0.f, v);      // rows 2, 3 += lane 31
    return lane_bcast(v, 63);
}
__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, dpp_f32<0xB1>(v, v)); v = fminf(v, dpp_f32<0x4E>(v, v)); v = fminf(v, dpp_f32<0x141>(v, v)); v = fminf(v, dpp_f32<0x140>(v, v));
    v = fminf(v, dpp_f32<0x142, 0xa>(v, v)); v = fminf(v, dpp_f32<0x143, 0xc>(v, v));
    return lane_bcast(v, 63);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v, v)); v = fmaxf(v, dpp_f32<0x4E>(v, v)); v = fmaxf(v, dpp_f32<0x141>(v, v)); v = fmaxf(v, dpp_f32<0x140>(v, v));
    v = fmaxf(v, dpp_f32<0x142, 0xa>(v, v)); v = fmaxf(v, dpp_f32<0x143, 0xc>(v, v));
    return lane_bcast(v, 63);
}
// inclusive product / sum scans across the 64 lanes of a wave: Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8), then the rows chained
__device__ __forceinline__ float wave_scan_mul(float v) {
    v *= dpp_f32<0x111>(1.f, v); v *= dpp_f32<0x112>(1.f, v); v *= dpp_f32<0x114>(1.f, v); v *= dpp_f32<0x118>(1.f, v);
    v *= dpp_f32<0x142, 0xa>(1.f, v); v *= dpp_f32<0x143, 0xc>(1.f, v);
    return v;
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_i32(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false); }
__device__ __forceinline__ int wave_scan_add(int v) {
    v += dpp_i32<0x111>(0, v); v += dpp_i32<0x112>(0, v); v += dpp_i32<0x114>(0, v); v += dpp_i32<0x118>(0, v);
    v += dpp_i32<0x142, 0xa>(0, v); v += dpp_i32<0x143, 0xc>(0, v);
    return v;
}
// sum over each aligned group of 4 lanes (a quad), in every lane of it
__device__ __forceinline__ float quad_sum(float v) { v += dpp_f32<0xB1>(0.f, v); v += dpp_f32<0x4E>(0.f, v); return v; }
__device__ __forceinline__ float wave_scan_add(float v) {
    v += dpp_f32<0x111>(0.f, v); v += dpp_f32<0x112>(0.f, v); v += dpp_f32<0x114>(0.f, v); v += dpp_f32<0x118>(0.f, v);
    v += dpp_f32<0x142, 0xa>(0.f, v); v += dpp_f32<0x143, 0xc>(0.f, v);
    return v;
}
#endif
