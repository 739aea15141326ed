// HBM-bound operator kernels for gfx950: bias_act (all 9 activations, grad orders 0-2), upfirdn2d
// (+ fused noise / bias / activation epilogue), filtered_lrelu (two fused passes), the depth-guided
// warp, the LPIPS normalise-diff-lin tail and multi-tensor Adam.  fp32, 16-byte vector accesses
// wherever the layout allows.
#include "common.hpp"
#include <stdarg.h>
#include <type_traits>
#include <stdio.h>

// ---- error text (thread-local) -------------------------------------------------------------------
static thread_local char g_err[512] = "";
void spi_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* spi_last_error(void) { return g_err; }
extern "C" int spi_abi_version(void) { return SPI_ABI_VERSION; }

__global__ void __launch_bounds__(256) zero_kernel(float* __restrict__ p, int64_t n) {
    const int64_t n4 = n >> 2;
    const bool aligned = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (aligned ? n4 : 0); i += (int64_t)gridDim.x * 256)
        reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (aligned ? n4 * 4 : 0) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0.f;
}

int spi_zero_async(float* p, int64_t n_floats, hipStream_t st) {
    if (n_floats <= 0) return SPI_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(n_floats, 1024), 4096);
    hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, st, p, n_floats);
    return SPI_OK;
}
extern "C" int spi_sizeof_conv_desc(void) { return (int)sizeof(spi_conv_desc); }

// ------------------------------------------------------------------------------------------------
// bias_act: y = clamp(act(x + b) * gain) and its first / second derivatives expressed through the
// saved input (xref) / output (yref), as the reference plugin does (bias_act.cu:27-151).
// ------------------------------------------------------------------------------------------------
struct ActParams { int act; int grad; float alpha, gain, clamp; };

__device__ __forceinline__ float act_apply(const ActParams& p, float x, float xref, float yref, float dy) {
    const float yy = (p.gain != 0.f) ? yref / p.gain : 0.f;      // activation output before the gain
    const int G = p.grad;
    float y = 0.f;
    switch (p.act) {
    case SPI_ACT_LINEAR:
        y = (G < 2) ? x : 0.f; break;
    case SPI_ACT_RELU:
        y = (G == 0) ? fmaxf(x, 0.f) : (G == 1 ? (yy > 0.f ? x : 0.f) : 0.f); break;
    case SPI_ACT_LRELU:
        y = (G == 0) ? (x > 0.f ? x : x * p.alpha) : (G == 1 ? (yy > 0.f ? x : x * p.alpha) : 0.f); break;
    case SPI_ACT_TANH:
        if (G == 0) { const float c = expf(x), d = 1.f / c; y = (x < -80.f) ? -1.f : (x > 80.f ? 1.f : (c - d) / (c + d)); }
        else if (G == 1) y = x * (1.f - yy * yy);
        else y = x * (1.f - yy * yy) * (-2.f * yy);
        break;
    case SPI_ACT_SIGMOID:
        if (G == 0) y = (x < -80.f) ? 0.f : 1.f / (expf(-x) + 1.f);
        else if (G == 1) y = x * yy * (1.f - yy);
        else y = x * yy * (1.f - yy) * (1.f - 2.f * yy);
        break;
    case SPI_ACT_ELU:
        if (G == 0) y = (x >= 0.f) ? x : expf(x) - 1.f;
        else if (G == 1) y = (yy >= 0.f) ? x : x * (yy + 1.f);
        else y = (yy >= 0.f) ? 0.f : x * (yy + 1.f);
        break;
    case SPI_ACT_SELU: {
        const float sc = 1.0507009873554804934193349852946f, sa = 1.6732632423543772848170429916717f;
        if (G == 0) y = (x >= 0.f) ? sc * x : (sc * sa) * (expf(x) - 1.f);
        else if (G == 1) y = (yy >= 0.f) ? x * sc : x * (yy + sc * sa);
        else y = (yy >= 0.f) ? 0.f : x * (yy + sc * sa);
        break; }
    case SPI_ACT_SOFTPLUS:
        if (G == 0) y = (x > 80.f) ? x : logf(expf(x) + 1.f);
        else if (G == 1) y = x * (1.f - expf(-yy));
        else { const float c = expf(-yy); y = x * c * (1.f - c); }
        break;
    case SPI_ACT_SWISH:
        if (G == 0) y = (x < -80.f) ? 0.f : x / (expf(-x) + 1.f);
        else {
            const float c = expf(xref), d = c + 1.f;
            if (G == 1) y = (xref > 40.f) ? x : x * c * (xref + d) / (d * d);
            else y = (xref > 40.f) ? 0.f : x * c * (xref * (2.f - d) + 2.f * d) / (d * d * d);
            yref = (xref < -80.f) ? 0.f : xref / (expf(-xref) + 1.f) * p.gain;
        }
        break;
    default: break;
    }
    y *= p.gain * dy;
    if (p.clamp >= 0.f) {
        if (G == 0) y = fminf(fmaxf(y, -p.clamp), p.clamp);
        else y = (yref > -p.clamp && yref < p.clamp) ? y : 0.f;
    }
    return y;
}

// the same table in double precision (spi_bias_act_t with SPI_DTYPE_F64: bias_act.cpp:81 dispatches the plugin for double, whose internal type
// IS double, bias_act.cu:14-16).  Never on the SPI path; a plain grid-stride kernel.
__device__ __forceinline__ double act_apply_f64(const ActParams& p, double x, double xref, double yref, double dy) {
    const double gain = p.gain, alpha = p.alpha, clampv = p.clamp;
    const double yy = (gain != 0.0) ? yref / gain : 0.0;
    const int G = p.grad;
    double y = 0.0;
    switch (p.act) {
    case SPI_ACT_LINEAR: y = (G < 2) ? x : 0.0; break;
    case SPI_ACT_RELU: y = (G == 0) ? fmax(x, 0.0) : (G == 1 ? (yy > 0.0 ? x : 0.0) : 0.0); break;
    case SPI_ACT_LRELU: y = (G == 0) ? (x > 0.0 ? x : x * alpha) : (G == 1 ? (yy > 0.0 ? x : x * alpha) : 0.0); break;
    case SPI_ACT_TANH:
        if (G == 0) y = tanh(x);
        else if (G == 1) y = x * (1.0 - yy * yy);
        else y = x * (1.0 - yy * yy) * (-2.0 * yy);
        break;
    case SPI_ACT_SIGMOID:
        if (G == 0) y = 1.0 / (exp(-x) + 1.0);
        else if (G == 1) y = x * yy * (1.0 - yy);
        else y = x * yy * (1.0 - yy) * (1.0 - 2.0 * yy);
        break;
    case SPI_ACT_ELU:
        if (G == 0) y = (x >= 0.0) ? x : exp(x) - 1.0;
        else if (G == 1) y = (yy >= 0.0) ? x : x * (yy + 1.0);
        else y = (yy >= 0.0) ? 0.0 : x * (yy + 1.0);
        break;
    case SPI_ACT_SELU: {
        const double sc = 1.0507009873554804934193349852946, sa = 1.6732632423543772848170429916717;
        if (G == 0) y = (x >= 0.0) ? sc * x : (sc * sa) * (exp(x) - 1.0);
        else if (G == 1) y = (yy >= 0.0) ? x * sc : x * (yy + sc * sa);
        else y = (yy >= 0.0) ? 0.0 : x * (yy + sc * sa);
        break; }
    case SPI_ACT_SOFTPLUS:
        if (G == 0) y = (x > 700.0) ? x : log(exp(x) + 1.0);
        else if (G == 1) y = x * (1.0 - exp(-yy));
        else { const double c = exp(-yy); y = x * c * (1.0 - c); }
        break;
    case SPI_ACT_SWISH:
        if (G == 0) y = x / (exp(-x) + 1.0);
        else {
            const double c = exp(xref), d = c + 1.0;
            if (G == 1) y = (xref > 300.0) ? x : x * c * (xref + d) / (d * d);
            else y = (xref > 300.0) ? 0.0 : x * c * (xref * (2.0 - d) + 2.0 * d) / (d * d * d);
            yref = xref / (exp(-xref) + 1.0) * gain;
        }
        break;
    default: break;
    }
    y *= gain * dy;
    if (clampv >= 0.0) {
        if (G == 0) y = fmin(fmax(y, -clampv), clampv);
        else y = (yref > -clampv && yref < clampv) ? y : 0.0;
    }
    return y;
}

__global__ void bias_act_f64_kernel(const double* __restrict__ x, const double* __restrict__ b, const double* __restrict__ xref,
                                    const double* __restrict__ yref, const double* __restrict__ dy, double* __restrict__ y,
                                    int64_t n, int sizeB, int64_t stepB, ActParams p) {
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
        const double bv = b ? b[(g / stepB) % sizeB] : 0.0;
        double xa = x[g], xr = xref ? xref[g] : 0.0;
        if (p.grad == 0) xa += bv; else xr += bv;
        y[g] = act_apply_f64(p, xa, xr, yref ? yref[g] : 0.0, dy ? dy[g] : 1.0);
    }
}

template <bool VEC>
__global__ void bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b, const float* __restrict__ xref,
                                const float* __restrict__ yref, const float* __restrict__ dy, float* __restrict__ y,
                                int64_t n, int sizeB, int64_t stepB, ActParams p) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (VEC) {      // n % 4 == 0, stepB % 4 == 0: a float4 never straddles a bias boundary
        const int64_t n4 = n >> 2;
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride) {
            const float4 xv = reinterpret_cast<const float4*>(x)[g];
            const float bv = b ? b[((g << 2) / stepB) % sizeB] : 0.f;
            float4 xr = xref ? reinterpret_cast<const float4*>(xref)[g] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 yr = yref ? reinterpret_cast<const float4*>(yref)[g] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 dv = dy ? reinterpret_cast<const float4*>(dy)[g] : make_float4(1.f, 1.f, 1.f, 1.f);
            float4 xa = xv;
            if (p.grad == 0) { xa.x += bv; xa.y += bv; xa.z += bv; xa.w += bv; }
            else { xr.x += bv; xr.y += bv; xr.z += bv; xr.w += bv; }
            float4 o;
            o.x = act_apply(p, xa.x, xr.x, yr.x, dv.x); o.y = act_apply(p, xa.y, xr.y, yr.y, dv.y);
            o.z = act_apply(p, xa.z, xr.z, yr.z, dv.z); o.w = act_apply(p, xa.w, xr.w, yr.w, dv.w);
            reinterpret_cast<float4*>(y)[g] = o;
        }
    } else {
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += stride) {
            const float bv = b ? b[(g / stepB) % sizeB] : 0.f;
            float xa = x[g], xr = xref ? xref[g] : 0.f;
            if (p.grad == 0) xa += bv; else xr += bv;
            y[g] = act_apply(p, xa, xr, yref ? yref[g] : 0.f, dy ? dy[g] : 1.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Typed plugin boundary (spi_bias_act_t / spi_upfirdn2d_t): the reference's plugins are instantiated for half / float / double
// (bias_act.cpp:81, upfirdn2d.cpp:67 AT_DISPATCH_FLOATING_TYPES_AND_HALF) and upfirdn2d takes any dense strides (channels_last for the
// fp16 super-resolution blocks, upfirdn2d.cpp:42 suggest_memory_format).  Like there, the arithmetic is fp32 for half tensors
// (InternalType<half> = float, bias_act.cu:14-16): one rounding, at the store; double tensors compute in double (round 5).
// ------------------------------------------------------------------------------------------------
template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
typedef _Float16 spi_half4 __attribute__((ext_vector_type(4)));
template <> struct Vec4<_Float16> { typedef spi_half4 type; };
__device__ __forceinline__ float4 to_f4(const float4& v) { return v; }
__device__ __forceinline__ float4 to_f4(const spi_half4& v) { return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w); }
template <typename T> __device__ __forceinline__ typename Vec4<T>::type from_f4(const float4& v);
template <> __device__ __forceinline__ float4 from_f4<float>(const float4& v) { return v; }
template <> __device__ __forceinline__ spi_half4 from_f4<_Float16>(const float4& v) {
    spi_half4 o; o.x = (_Float16)v.x; o.y = (_Float16)v.y; o.z = (_Float16)v.z; o.w = (_Float16)v.w; return o;      // round-to-nearest-even, once
}

template <typename T, bool VEC>
__global__ void bias_act_t_kernel(const T* __restrict__ x, const T* __restrict__ b, const T* __restrict__ xref, const T* __restrict__ yref,
                                  const T* __restrict__ dy, T* __restrict__ y, int64_t n, int sizeB, int64_t stepB, ActParams p) {
    typedef typename Vec4<T>::type V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (VEC) {      // n % 4 == 0, stepB % 4 == 0, pointers aligned to 4 elements
        const int64_t n4 = n >> 2;
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride) {
            float4 xa = to_f4(reinterpret_cast<const V*>(x)[g]);
            const float bv = b ? (float)b[((g << 2) / stepB) % sizeB] : 0.f;
            float4 xr = xref ? to_f4(reinterpret_cast<const V*>(xref)[g]) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 yr = yref ? to_f4(reinterpret_cast<const V*>(yref)[g]) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 dv = dy ? to_f4(reinterpret_cast<const V*>(dy)[g]) : make_float4(1.f, 1.f, 1.f, 1.f);
            if (p.grad == 0) { xa.x += bv; xa.y += bv; xa.z += bv; xa.w += bv; }
            else { xr.x += bv; xr.y += bv; xr.z += bv; xr.w += bv; }
            float4 o;
            o.x = act_apply(p, xa.x, xr.x, yr.x, dv.x); o.y = act_apply(p, xa.y, xr.y, yr.y, dv.y);
            o.z = act_apply(p, xa.z, xr.z, yr.z, dv.z); o.w = act_apply(p, xa.w, xr.w, yr.w, dv.w);
            reinterpret_cast<V*>(y)[g] = from_f4<T>(o);
        }
    } else {
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += stride) {
            const float bv = b ? (float)b[(g / stepB) % sizeB] : 0.f;
            float xa = (float)x[g], xr = xref ? (float)xref[g] : 0.f;
            if (p.grad == 0) xa += bv; else xr += bv;
            y[g] = (T)act_apply(p, xa, xr, yref ? (float)yref[g] : 0.f, dy ? (float)dy[g] : 1.f);
        }
    }
}

template <typename T>
static int launch_bias_act_t(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y, int64_t n, int sizeB,
                             int64_t stepB, const ActParams& p, spi_stream_t stream) {
    const uintptr_t al = 4 * sizeof(T) - 1;
    const bool aligned = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)xref | (uintptr_t)yref | (uintptr_t)dy) & al) == 0;
    const bool vec = aligned && (n % 4 == 0) && (b == nullptr || stepB % 4 == 0);
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(vec ? n / 4 : n, 256), 256 * 16);
    if (vec) hipLaunchKernelGGL((bias_act_t_kernel<T, true>), dim3(grid), dim3(256), 0, as_stream(stream), (const T*)x, (const T*)b, (const T*)xref,
                                (const T*)yref, (const T*)dy, (T*)y, n, sizeB, stepB, p);
    else hipLaunchKernelGGL((bias_act_t_kernel<T, false>), dim3(grid), dim3(256), 0, as_stream(stream), (const T*)x, (const T*)b, (const T*)xref,
                            (const T*)yref, (const T*)dy, (T*)y, n, sizeB, stepB, p);
    SPI_LAUNCH_CHECK("spi_bias_act_t");
    return SPI_OK;
}

// ------------------------------------------------------------------------------------------------
// fp16 activation tensors (round 5; the reference's use_fp16 blocks, networks_stylegan2.py:421-436): the kernels of the super-resolution
// path that stream activations -- layer-tail backward, channel dot products, gradient-segment map, the tiled 4x4 FIR with its fused tail --
// are instantiated for float and _Float16 TENSORS; arithmetic and every accumulator stay fp32, a value is rounded once when it is stored.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load4(const float* p, float (&v)[4], bool nt) {
    const f32x4_t t = nt ? __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p)) : *reinterpret_cast<const f32x4_t*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load4(const _Float16* p, float (&v)[4], bool nt) {
    const f16x4_t t = nt ? __builtin_nontemporal_load(reinterpret_cast<const f16x4_t*>(p)) : *reinterpret_cast<const f16x4_t*>(p);
    v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w;
}
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }
__device__ __forceinline__ void store4(_Float16* p, float a, float b, float c, float d) {
    const f16x4_t t = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
    *reinterpret_cast<f16x4_t*>(p) = t;
}

// ------------------------------------------------------------------------------------------------
// Layer-tail backward: activation gradient + bias gradient (sum over n, hw) + noise gradient
// (sum over n, c) in one pass.  Block = 256 threads x PX consecutive pixels; blockIdx.y picks a chunk
// of channels.  Per channel the block adds 4 wave-partials to d_bias[c]; the per-pixel sums stay in
// registers over the channel loop and are added to d_pixsum once.
// ------------------------------------------------------------------------------------------------
// optional fused per-(n, c) dot product  zdot[n*C + c] += sum_p dz[n,c,p] * z[n,c,p]  with z = the layer's conv result reconstructed from its output
// (inverse activation, minus bias and noise -- exactly spi_chan_dot's b'): the frozen-weight style gradient needs it (networks_stylegan2._ModConvFrozen),
// and here dz and y are in registers already -- a separate spi_chan_dot pass reads both tensors again (3 % of a stage-1 step).
struct ZDot { float* out; const float* bias; const float* noise; const float* noise_gain; };

template <int PX, typename T = float, bool ZD = false>
__global__ void __launch_bounds__(256) tail_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                       T* __restrict__ dz, float* __restrict__ d_bias,
                                                       float* __restrict__ d_pixsum, const float* __restrict__ noise,
                                                       float* __restrict__ d_strength, int N, int C, int64_t HW, int cchunk,
                                                       ActParams ap, ZDot zd) {
    __shared__ float bpart[512][4];                        // per-channel wave partials of this block (host keeps cchunk <= 512)
    __shared__ float zpart[ZD ? 2048 : 1][4];              // ZDot: per-(n, c) wave partials (host keeps cchunk * N <= 2048)
    const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * PX;
    const int c_beg = blockIdx.y * cchunk, c_end = min(c_beg + cchunk, C);
    const bool ok = p0 < HW;                               // PX == 4 requires HW % 4 == 0, so a thread is all-in or all-out
    float pix[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) pix[j] = 0.f;
    // (n, c) pairs of the chunk are walked UN at a time with all 2*UN loads issued before the first use: with the per-pixel sums
    // wanted the grid is small (every channel split costs one same-address atomic per pixel), so the memory-level parallelism has to
    // come from inside the thread -- one pair at a time ran at 2.9 TB/s on the 512^2 layers.
    constexpr int UN = 4;
    const int npair = (c_end - c_beg) * N;
    float bsum = 0.f;
    int cur_c = c_beg;
    float znz[PX];                                         // ZDot: this thread's noise * strength (pixels are fixed over the channel loop)
    const float z_inv_alpha = (ap.act == SPI_ACT_LRELU) ? 1.f / ap.alpha : 1.f;
#pragma unroll
    for (int j = 0; j < PX; ++j) znz[j] = (ZD && zd.noise && ok) ? zd.noise[p0 + j] * (zd.noise_gain ? zd.noise_gain[0] : 1.f) : 0.f;
    for (int q0 = 0; q0 < npair; q0 += UN) {
        float g[UN][PX], yy[UN][PX];
        int64_t off[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int q = min(q0 + u, npair - 1);
            const int c = c_beg + q / N, n = q - (q / N) * N;
            off[u] = ((int64_t)n * C + c) * HW + p0;
            if (ok) {
                if constexpr (PX == 4) {
                    load4(dy + off[u], g[u], true);                                   // dy and y are read once
                    if (y) load4(y + off[u], yy[u], true);
                } else { g[u][0] = (float)dy[off[u]]; if (y) yy[u][0] = (float)y[off[u]]; }
            } else {
#pragma unroll
                for (int j = 0; j < PX; ++j) g[u][j] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int q = q0 + u;
            if (q >= npair) break;
            const int c = c_beg + q / N;
            if (c != cur_c) {                                  // channel finished: its wave partial goes to LDS
                if (d_bias) {
                    const float bs = wave_sum(bsum);
                    if ((threadIdx.x & 63) == 0) bpart[cur_c - c_beg][threadIdx.x >> 6] = bs;
                }
                bsum = 0.f; cur_c = c;
            }
            float zsum = 0.f;
            if (y && ok) {
                const float zb = (ZD && zd.bias) ? zd.bias[c] : 0.f;
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    const float pre = yy[u][j] / ap.gain;                      // activation output before the gain (as bias_act grad=1)
                    float v = g[u][j];
                    if (ap.act == SPI_ACT_RELU) v = pre > 0.f ? v : 0.f;
                    else if (ap.act == SPI_ACT_LRELU) v = pre > 0.f ? v : v * ap.alpha;
                    v *= ap.gain;
                    if (ap.clamp >= 0.f) v = (yy[u][j] > -ap.clamp && yy[u][j] < ap.clamp) ? v : 0.f;
                    g[u][j] = v;
                    if (ZD) zsum = fmaf(v, (pre > 0.f ? pre : pre * z_inv_alpha) - zb - znz[j], zsum);   // (a clamped / rectified-away element has v == 0)
                }
                if (dz) {
                    if constexpr (PX == 4) store4(dz + off[u], g[u][0], g[u][1], g[u][2], g[u][3]);
                    else dz[off[u]] = (T)g[u][0];
                }
            }
#pragma unroll
            for (int j = 0; j < PX; ++j) { pix[j] += g[u][j]; bsum += g[u][j]; }
            if (ZD) {                                          // wave partial of this (n, c) -> LDS; one atomic per pair and block at the end
                const float zs = wave_sum(zsum);
                if ((threadIdx.x & 63) == 0) zpart[q][threadIdx.x >> 6] = zs;
            }
        }
    }
    if (d_bias && npair > 0) {
        const float bs = wave_sum(bsum);
        if ((threadIdx.x & 63) == 0) bpart[cur_c - c_beg][threadIdx.x >> 6] = bs;
    }
    if (d_bias) {                                          // one global atomic per channel per block, issued by different lanes
        __syncthreads();
        for (int i = threadIdx.x; i < c_end - c_beg; i += 256) {
            const int cl = (i + blockIdx.x) % (c_end - c_beg);                 // staggered so that blocks do not hit the same address together
            atomicAdd(d_bias + c_beg + cl, (bpart[cl][0] + bpart[cl][1]) + (bpart[cl][2] + bpart[cl][3]));
        }
    }
    if (ZD) {                                              // (every block walks the pairs from another start: same-address atomics issued together serialise --
        __syncthreads();                                   //  one atomic per wave straight from the loop ran 3 x slower than a separate spi_chan_dot pass)
        for (int i = threadIdx.x; i < npair; i += 256) {
            const int q = (i + (int)blockIdx.x * 37) % npair;
            atomicAdd(zd.out + (int64_t)(q - (q / N) * N) * C + c_beg + q / N, (zpart[q][0] + zpart[q][1]) + (zpart[q][2] + zpart[q][3]));
        }
    }
    if (d_pixsum && ok) {
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            if (gridDim.y == 1) d_pixsum[p0 + j] = pix[j];
            else atomicAdd(d_pixsum + p0 + j, pix[j]);
        }
    }
    if (d_strength) {                                      // sum_p pixsum[p] * noise[p] is linear in the blocks' partial pixel sums: one atomic per block
        float sn = 0.f;
        if (ok) {
#pragma unroll
            for (int j = 0; j < PX; ++j) sn = fmaf(pix[j], noise[p0 + j], sn);
        }
        sn = wave_sum(sn);
        __shared__ float spart[4];
        if ((threadIdx.x & 63) == 0) spart[threadIdx.x >> 6] = sn;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(d_strength, (spart[0] + spart[1]) + (spart[2] + spart[3]));
    }
}

// ------------------------------------------------------------------------------------------------
// per-row dot products  out[r] += sum_p a[r,p] * b'[r,p]  (b' = b, or the conv result reconstructed from a layer output)
// ------------------------------------------------------------------------------------------------
template <typename T = float>
__global__ void __launch_bounds__(256) chan_dot_kernel(const T* __restrict__ a, const T* __restrict__ b, float* __restrict__ out,
                                                       int C, int64_t HW, int64_t per_block, const float* __restrict__ bias,
                                                       const float* __restrict__ noise, const float* __restrict__ noise_gain,
                                                       int act, float alpha, float gain) {
    const int64_t r = blockIdx.y;
    const int c = (int)(r % C);
    const int64_t p0 = (int64_t)blockIdx.x * per_block, p1 = min(p0 + per_block, HW);
    const T* ar = a + r * HW; const T* br = b + r * HW;
    const float bv = (act && bias) ? bias[c] : 0.f;
    const float ng = (act && noise) ? (noise_gain ? noise_gain[0] : 1.f) : 0.f;
    const float inv_gain = act ? 1.f / gain : 1.f;
    const float inv_alpha = (act == SPI_ACT_LRELU) ? 1.f / alpha : 1.f;
    float s0 = 0.f, s1 = 0.f;
    auto term = [&](float av, float v, int64_t p) {
        if (act) { v *= inv_gain; v = (v > 0.f ? v : v * inv_alpha) - bv - (noise ? noise[p] * ng : 0.f); }
        return av * v;
    };
    // 16-byte loads, four of each operand in flight per thread (scalar loads two at a time ran at 2 TB/s); host guarantees per_block % 4096 == 0
    const bool vec = (HW % 4 == 0) && (((uintptr_t)ar | (uintptr_t)br) % (4 * sizeof(T)) == 0);
    if (vec) {
        for (int64_t p = p0 + threadIdx.x * 4; p < p1; p += 4096) {
            float av[4][4], bvv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t q = min(p + u * 1024, HW - 4);
                load4(ar + q, av[u], false); load4(br + q, bvv[u], false);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t q = p + u * 1024;
                if (q < p1) {
                    s0 += term(av[u][0], bvv[u][0], q) + term(av[u][1], bvv[u][1], q + 1);
                    s1 += term(av[u][2], bvv[u][2], q + 2) + term(av[u][3], bvv[u][3], q + 3);
                }
            }
        }
    } else {
        for (int64_t p = p0 + threadIdx.x; p < p1; p += 512) {
            s0 += term((float)ar[p], (float)br[p], p);
            const int64_t p2 = p + 256;
            if (p2 < p1) s1 += term((float)ar[p2], (float)br[p2], p2);
        }
    }
    __shared__ float red[4];
    const float s = wave_sum(s0 + s1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + r, (red[0] + red[1]) + (red[2] + red[3]));
}

// ------------------------------------------------------------------------------------------------
// upfirdn2d: out[oy,ox] = gain * sum_t k[ty,tx] * U[oy*down + ty, ox*down + tx], U = zero-inserted,
// padded input; k = f flipped unless `flip` (upfirdn2d.py:168-213).  Only taps that land on a real
// sample are visited.  Optional pre-bias (filtered_lrelu step 1) and noise/bias/activation epilogue
// (SynthesisLayer.forward tail, networks_stylegan2.py:320-329) keep the tensor in registers.
// ------------------------------------------------------------------------------------------------
struct UpfirdnParams {
    int N, C, inH, inW, fH, fW, upx, upy, downx, downy, padx0, pady0, flip, outH, outW;
    float gain;
};
constexpr int MAX_TAPS = 256;

template <typename TI = float, typename TO = float>
__global__ void __launch_bounds__(256) upfirdn2d_kernel(const TI* __restrict__ x, const float* __restrict__ f,
                                                        TO* __restrict__ y, UpfirdnParams p,
                                                        const float* __restrict__ pre_bias, const float* __restrict__ noise,
                                                        const float* __restrict__ noise_gain, const float* __restrict__ bias,
                                                        ActParams ap) {
    __shared__ float sf[MAX_TAPS];
    for (int i = threadIdx.x; i < p.fH * p.fW; i += blockDim.x) {
        const int ty = i / p.fW, tx = i % p.fW;
        sf[i] = (p.flip ? f[ty * p.fW + tx] : f[(p.fH - 1 - ty) * p.fW + (p.fW - 1 - tx)]) * p.gain;
    }
    __syncthreads();
    const float ng = (noise && noise_gain) ? noise_gain[0] : (noise ? 1.f : 0.f);
    const int64_t plane = (int64_t)p.outH * p.outW;
    const int64_t total = (int64_t)p.N * p.C * plane;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nc = g / plane;
        const int rem = (int)(g - nc * plane);
        const int oy = rem / p.outW, ox = rem - oy * p.outW;
        const int c = (int)(nc % p.C);
        const TI* xp = x + nc * (int64_t)p.inH * p.inW;
        const float pb = pre_bias ? pre_bias[c] : 0.f;
        const int by = oy * p.downy - p.pady0, bx = ox * p.downx - p.padx0;     // U index of tap 0 relative to sample 0
        int ty0 = (-by) % p.upy; if (ty0 < 0) ty0 += p.upy;
        int tx0 = (-bx) % p.upx; if (tx0 < 0) tx0 += p.upx;
        float acc = 0.f;
        for (int ty = ty0; ty < p.fH; ty += p.upy) {
            const int iy = (by + ty) / p.upy;
            if (by + ty < 0 || iy >= p.inH) continue;
            for (int tx = tx0; tx < p.fW; tx += p.upx) {
                const int ix = (bx + tx) / p.upx;
                if (bx + tx < 0 || ix >= p.inW) continue;
                acc = fmaf(sf[ty * p.fW + tx], (float)xp[(int64_t)iy * p.inW + ix] + pb, acc);
            }
        }
        if (noise) acc += noise[rem] * ng;
        if (ap.act != 0) acc = act_apply(ap, acc + (bias ? bias[c] : 0.f), 0.f, 0.f, 1.f);
        y[g] = (TO)acc;
    }
}

// first pass of filtered_lrelu on half tensors: fp16 input and fp16 bias, fp32 output (bias -> up-FIR -> lrelu * gain, clamp)
__global__ void __launch_bounds__(256) upfirdn2d_hb_kernel(const _Float16* __restrict__ x, const float* __restrict__ f, float* __restrict__ y, UpfirdnParams p,
                                                           const _Float16* __restrict__ pre_bias, ActParams ap) {
    __shared__ float sf[MAX_TAPS];
    for (int i = threadIdx.x; i < p.fH * p.fW; i += blockDim.x) {
        const int ty = i / p.fW, tx = i % p.fW;
        sf[i] = (p.flip ? f[ty * p.fW + tx] : f[(p.fH - 1 - ty) * p.fW + (p.fW - 1 - tx)]) * p.gain;
    }
    __syncthreads();
    const int64_t plane = (int64_t)p.outH * p.outW;
    const int64_t total = (int64_t)p.N * p.C * plane;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nc = g / plane;
        const int rem = (int)(g - nc * plane);
        const int oy = rem / p.outW, ox = rem - oy * p.outW;
        const int c = (int)(nc % p.C);
        const _Float16* xp = x + nc * (int64_t)p.inH * p.inW;
        const float pb = pre_bias ? (float)pre_bias[c] : 0.f;
        const int by = oy * p.downy - p.pady0, bx = ox * p.downx - p.padx0;
        int ty0 = (-by) % p.upy; if (ty0 < 0) ty0 += p.upy;
        int tx0 = (-bx) % p.upx; if (tx0 < 0) tx0 += p.upx;
        float acc = 0.f;
        for (int ty = ty0; ty < p.fH; ty += p.upy) {
            const int iy = (by + ty) / p.upy;
            if (by + ty < 0 || iy >= p.inH) continue;
            for (int tx = tx0; tx < p.fW; tx += p.upx) {
                const int ix = (bx + tx) / p.upx;
                if (bx + tx < 0 || ix >= p.inW) continue;
                acc = fmaf(sf[ty * p.fW + tx], (float)xp[(int64_t)iy * p.inW + ix] + pb, acc);
            }
        }
        y[g] = act_apply(ap, acc, 0.f, 0.f, 1.f);
    }
}

struct Strides4 { int64_t n, c, h, w; };

// generic upfirdn2d over arbitrary dense strides (NCHW or channels_last), typed I/O, fp32 taps and accumulation.  The thread index runs over
// the OUTPUT in its own memory order (the stride with |w| == 1 fastest for NCHW, c fastest for channels_last), so stores stay coalesced.
template <typename T>
__global__ void __launch_bounds__(256) upfirdn2d_t_kernel(const T* __restrict__ x, const float* __restrict__ f, T* __restrict__ y, UpfirdnParams p,
                                                          Strides4 xs, Strides4 ys, int c_fastest) {
    using S = std::conditional_t<std::is_same<T, double>::value, double, float>;      // internal type (upfirdn2d.cu: double for double, else float)
    __shared__ float sf[MAX_TAPS];
    for (int i = threadIdx.x; i < p.fH * p.fW; i += blockDim.x) {
        const int ty = i / p.fW, tx = i % p.fW;
        sf[i] = (p.flip ? f[ty * p.fW + tx] : f[(p.fH - 1 - ty) * p.fW + (p.fW - 1 - tx)]) * p.gain;
    }
    __syncthreads();
    const int64_t total = (int64_t)p.N * p.C * p.outH * p.outW;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        int n, c, oy, ox;
        if (c_fastest) { c = (int)(g % p.C); int64_t r = g / p.C; ox = (int)(r % p.outW); r /= p.outW; oy = (int)(r % p.outH); n = (int)(r / p.outH); }
        else { ox = (int)(g % p.outW); int64_t r = g / p.outW; oy = (int)(r % p.outH); r /= p.outH; c = (int)(r % p.C); n = (int)(r / p.C); }
        const T* xp = x + n * xs.n + c * xs.c;
        const int by = oy * p.downy - p.pady0, bx = ox * p.downx - p.padx0;
        int ty0 = (-by) % p.upy; if (ty0 < 0) ty0 += p.upy;
        int tx0 = (-bx) % p.upx; if (tx0 < 0) tx0 += p.upx;
        S acc = 0;
        for (int ty = ty0; ty < p.fH; ty += p.upy) {
            const int iy = (by + ty) / p.upy;
            if (by + ty < 0 || iy >= p.inH) continue;
            for (int tx = tx0; tx < p.fW; tx += p.upx) {
                const int ix = (bx + tx) / p.upx;
                if (bx + tx < 0 || ix >= p.inW) continue;
                acc += (S)sf[ty * p.fW + tx] * (S)xp[iy * xs.h + ix * xs.w];
            }
        }
        y[n * ys.n + c * ys.c + oy * ys.h + ox * ys.w] = (T)acc;
    }
}

// Specialisations for the generator's 4x4 low-pass (the only filter on the SPI path).
//   UP = DOWN = 1 (FIR after every stride-2 transposed conv and its adjoint): each thread produces four
//   consecutive outputs of one row from a 4 x 7 input patch held in registers (7 loads per output
//   instead of 16, no integer division in the tap loop).
//   Other up/down in {1,2}: one output per thread, tap loops fully unrolled at compile time.
template <int UP, int DOWN>
__global__ void __launch_bounds__(256) upfirdn2d_4x4_kernel(const float* __restrict__ x, const float* __restrict__ f,
                                                            float* __restrict__ y, UpfirdnParams p,
                                                            const float* __restrict__ pre_bias, const float* __restrict__ noise,
                                                            const float* __restrict__ noise_gain, const float* __restrict__ bias,
                                                            ActParams ap) {
    __shared__ float sf[16];
    if (threadIdx.x < 16) {
        const int ty = threadIdx.x >> 2, tx = threadIdx.x & 3;
        sf[threadIdx.x] = (p.flip ? f[ty * 4 + tx] : f[(3 - ty) * 4 + (3 - tx)]) * p.gain;
    }
    __syncthreads();
    float k[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) k[i] = sf[i];
    const float ng = (noise && noise_gain) ? noise_gain[0] : (noise ? 1.f : 0.f);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(x, (int64_t)p.N * p.C * p.inH * p.inW * 4);          // host guarantees < 2 GiB
    const int64_t plane = (int64_t)p.outH * p.outW;
    if (UP == 1 && DOWN == 1) {
        const int gw = (p.outW + 3) >> 2;                        // groups of 4 outputs per row
        const int64_t total = (int64_t)p.N * p.C * p.outH * gw;
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
            const int xg = (int)(g % gw);
            const int64_t rowid = g / gw;
            const int oy = (int)(rowid % p.outH);
            const int64_t nc = rowid / p.outH;
            const int c = (int)(nc % p.C);
            const int ox = xg << 2;
            const unsigned xoff = (unsigned)(nc * p.inH * p.inW);
            const float pb = pre_bias ? pre_bias[c] : 0.f;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const int bx = ox - p.padx0, by = oy - p.pady0;
#pragma unroll
            for (int ty = 0; ty < 4; ++ty) {
                const int iy = by + ty;
                const bool yin = iy >= 0 && iy < p.inH;
                float v[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    const int ix = bx + i;
                    const bool ok = yin && ix >= 0 && ix < p.inW;
                    v[i] = buf_load_f32(rs, ok ? (xoff + (unsigned)(iy * p.inW + ix)) * 4u : BUF_OOB);
                    if (pre_bias) v[i] = ok ? v[i] + pb : 0.f;
                }
#pragma unroll
                for (int o = 0; o < 4; ++o)
#pragma unroll
                    for (int tx = 0; tx < 4; ++tx) acc[o] = fmaf(k[ty * 4 + tx], v[o + tx], acc[o]);
            }
            float* yp = y + nc * plane + (int64_t)oy * p.outW + ox;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (ox + o >= p.outW) break;
                float a = acc[o];
                if (noise) a += noise[(int64_t)oy * p.outW + ox + o] * ng;
                if (ap.act != 0) a = act_apply(ap, a + (bias ? bias[c] : 0.f), 0.f, 0.f, 1.f);
                yp[o] = a;
            }
        }
    } else {
        const int64_t total = (int64_t)p.N * p.C * plane;
        for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
            const int64_t nc = g / plane;
            const int rem = (int)(g - nc * plane);
            const int oy = rem / p.outW, ox = rem - oy * p.outW;
            const int c = (int)(nc % p.C);
            const unsigned xoff = (unsigned)(nc * p.inH * p.inW);
            const float pb = pre_bias ? pre_bias[c] : 0.f;
            const int by = oy * DOWN - p.pady0, bx = ox * DOWN - p.padx0;
            const int ty0 = (UP == 1) ? 0 : (by & 1), tx0 = (UP == 1) ? 0 : (bx & 1);    // first tap that lands on a real sample
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 4 / UP; ++a) {
                const int ty = ty0 + a * UP;
                const int uy = by + ty;
                const int iy = (UP == 1) ? uy : (uy >> 1);
                const bool yin = uy >= 0 && iy < p.inH;
#pragma unroll
                for (int b = 0; b < 4 / UP; ++b) {
                    const int tx = tx0 + b * UP;
                    const int ux = bx + tx;
                    const int ix = (UP == 1) ? ux : (ux >> 1);
                    const bool ok = yin && ux >= 0 && ix < p.inW;
                    const float kv = (UP == 1) ? k[a * 4 + b] : sf[ty * 4 + tx];     // UP = 2: tap index depends on the output parity -> LDS lookup
                    float xv = buf_load_f32(rs, ok ? (xoff + (unsigned)(iy * p.inW + ix)) * 4u : BUF_OOB);
                    if (pre_bias) xv = ok ? xv + pb : 0.f;
                    acc = fmaf(kv, xv, acc);
                }
            }
            if (noise) acc += noise[rem] * ng;
            if (ap.act != 0) acc = act_apply(ap, acc + (bias ? bias[c] : 0.f), 0.f, 0.f, 1.f);
            y[g] = acc;
        }
    }
}

// UP = DOWN = 1 on images >= 128 px: LDS-tiled version.  A 256-thread block produces a 64 x 64 output tile from a
// 67 x 67 input window staged in LDS: 17 row-coalesced global loads per thread, all issued before the first is
// consumed (1.1 loads per output instead of 7); each thread then reads its 7 x 7 patch as fourteen 16-byte LDS
// reads and writes 4 x 4 outputs.  HBM-bound: 8 B/output algorithmic.
constexpr int FT_W = 64, FT_H = 64, FT_LD = 68, FT_ROWS = FT_H + 3, FT_PASS = (FT_ROWS + 3) / 4;
template <typename T = float>
__global__ void __launch_bounds__(256) upfirdn2d_4x4_tiled_kernel(const T* __restrict__ x, const float* __restrict__ f,
                                                                  T* __restrict__ y, UpfirdnParams p,
                                                                  const float* __restrict__ pre_bias, const float* __restrict__ noise,
                                                                  const float* __restrict__ noise_gain, const float* __restrict__ bias,
                                                                  ActParams ap) {
    __shared__ __attribute__((aligned(16))) float tile[FT_PASS * 4 * FT_LD];
    __shared__ float sf[16];
    const int tid = threadIdx.x;
    if (tid < 16) {
        const int ty = tid >> 2, tx = tid & 3;
        sf[tid] = (p.flip ? f[ty * 4 + tx] : f[(3 - ty) * 4 + (3 - tx)]) * p.gain;
    }
    const int nc = blockIdx.z, c = nc % p.C;
    const int ox0 = blockIdx.x * FT_W, oy0 = blockIdx.y * FT_H;
    const T* xp = x + (int64_t)nc * p.inH * p.inW;
    const float pb = pre_bias ? pre_bias[c] : 0.f;
    const int bx = ox0 - p.padx0, by = oy0 - p.pady0;
    constexpr unsigned ES = sizeof(T);
    auto ldx = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) -> float {
        if constexpr (sizeof(T) == 2) return (float)__builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0));
        else return buf_load_f32(rs, off);
    };
    {   // stage the window: lanes along x (64 columns), 4 rows per pass; the 3 halo columns 64..66 by the first threads.
        // Branch-free buffer loads (out-of-image -> hardware zero), so all 18 are in flight together.
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(xp, (int64_t)p.inH * p.inW * ES);
        const int lc = tid & 63, lr = tid >> 6;
        const int ix = bx + lc;
        const bool xin = ix >= 0 && ix < p.inW;
        float v[FT_PASS];
#pragma unroll
        for (int k = 0; k < FT_PASS; ++k) {
            const int iy = by + lr + 4 * k;
            const bool ok = xin && iy >= 0 && iy < p.inH;
            v[k] = ldx(rs, ok ? (unsigned)(iy * p.inW + ix) * ES : BUF_OOB);
            if (pre_bias) v[k] = ok ? v[k] + pb : 0.f;
        }
        const int hr = min(tid / 3, FT_ROWS - 1), hc = 64 + tid % 3;
        const int iyh = by + hr, ixh = bx + hc;
        const bool okh = tid < FT_ROWS * 3 && iyh >= 0 && iyh < p.inH && ixh >= 0 && ixh < p.inW;
        float h = ldx(rs, okh ? (unsigned)(iyh * p.inW + ixh) * ES : BUF_OOB);
        if (pre_bias) h = okh ? h + pb : 0.f;
#pragma unroll
        for (int k = 0; k < FT_PASS; ++k) tile[(lr + 4 * k) * FT_LD + lc] = v[k];
        if (tid < FT_ROWS * 3) tile[hr * FT_LD + hc] = h;
    }
    __syncthreads();
    float k[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) k[i] = sf[i];
    const int tx = tid & 15, ty = tid >> 4;
    float acc[4][4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[o][j] = 0.f;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const float4 a = *reinterpret_cast<const float4*>(&tile[(4 * ty + r) * FT_LD + 4 * tx]);
        const float4 b = *reinterpret_cast<const float4*>(&tile[(4 * ty + r) * FT_LD + 4 * tx + 4]);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int t = r - o;                                   // filter row that input row r feeds for output row o
            if (t < 0 || t > 3) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[o][j] = fmaf(k[t * 4 + q], v[j + q], acc[o][j]);
        }
    }
    const float ng = (noise && noise_gain) ? noise_gain[0] : (noise ? 1.f : 0.f);
    const float bv = bias ? bias[c] : 0.f;
    const float slope = ap.act == SPI_ACT_LRELU ? ap.alpha : 1.f;
    const int ox = ox0 + 4 * tx;
    if (ox >= p.outW) return;
    const bool vec = (p.outW & 3) == 0;                            // rows stay 16-byte aligned
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int oy = oy0 + 4 * ty + o;
        if (oy >= p.outH) break;
        const int64_t pix = (int64_t)oy * p.outW + ox;
        T* yp = y + (int64_t)nc * p.outH * p.outW + pix;
        float r4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = acc[o][j];
            if (noise && ox + j < p.outW) a += noise[pix + j] * ng;
            if (ap.act != 0) {                                     // linear / lrelu only (host dispatch): no per-output switch
                a += bv;
                a = (a > 0.f ? a : a * slope) * ap.gain;
                if (ap.clamp >= 0.f) a = fminf(fmaxf(a, -ap.clamp), ap.clamp);
            }
            r4[j] = a;
        }
        if (vec) store4(yp, r4[0], r4[1], r4[2], r4[3]);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (ox + j < p.outW) yp[j] = (T)r4[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rotate(): unproject target pixels with the target depth, project into the source camera, sample
// source depth / image / mask, keep depth-consistent in-frame pixels (spi/utils/rotate.py:5-116).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float resized_depth(const float* __restrict__ d, int dres, int res, int yy, int xx) {
    // F.interpolate(bilinear, align_corners=False) from dres^2 to res^2, evaluated at integer (yy,xx)
    if (dres == res) return d[yy * dres + xx];
    const float sc = (float)dres / (float)res;
    float sy = fmaxf(((float)yy + 0.5f) * sc - 0.5f, 0.f), sx = fmaxf(((float)xx + 0.5f) * sc - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, dres - 1), x1 = min(x0 + 1, dres - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    return (1.f - ly) * ((1.f - lx) * d[y0 * dres + x0] + lx * d[y0 * dres + x1]) +
           ly * ((1.f - lx) * d[y1 * dres + x0] + lx * d[y1 * dres + x1]);
}

struct Bilin { int x0, y0; float w00, w01, w10, w11; bool v00, v01, v10, v11; };
__device__ __forceinline__ Bilin bilin_setup(float gx, float gy, int res) {
    const float ix = ((gx + 1.f) * (float)res - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)res - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    Bilin b;
    b.x0 = (int)fx; b.y0 = (int)fy;
    const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
    b.w00 = wx0 * wy0; b.w01 = wx1 * wy0; b.w10 = wx0 * wy1; b.w11 = wx1 * wy1;
    const bool x0ok = b.x0 >= 0 && b.x0 < res, x1ok = b.x0 + 1 >= 0 && b.x0 + 1 < res;
    const bool y0ok = b.y0 >= 0 && b.y0 < res, y1ok = b.y0 + 1 >= 0 && b.y0 + 1 < res;
    b.v00 = x0ok && y0ok; b.v01 = x1ok && y0ok; b.v10 = x0ok && y1ok; b.v11 = x1ok && y1ok;
    return b;
}

__global__ void rotate_warp_kernel(const float* __restrict__ tgt_cam, const float* __restrict__ src_inv,
                                   const float* __restrict__ src_cam, const float* __restrict__ tgt_depth,
                                   const float* __restrict__ src_depth, const float* __restrict__ src_image,
                                   const float* __restrict__ src_mask, int N, int res, int dres, float eps,
                                   float* __restrict__ warp_rgb, float* __restrict__ warp_mask) {
    const int64_t plane = (int64_t)res * res;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)N * plane) return;
    const int n = (int)(g / plane);
    const int rem = (int)(g - n * plane);
    const int yy = rem / res, xx = rem - yy * res;
    const float* T = tgt_cam + n * 25; const float* TK = T + 16;
    const float* SK = src_cam + n * 25 + 16; const float* SI = src_inv + n * 16;
    const float* td = tgt_depth + (int64_t)n * dres * dres;
    const float* sd = src_depth + (int64_t)n * dres * dres;
    const float z = resized_depth(td, dres, res, yy, xx);
    const float inv = 1.f / (float)res, half = 0.5f / (float)res;
    const float u = (float)xx * inv + half, v = (float)yy * inv + half;
    float fx = TK[0], sk = TK[1], cx = TK[2], fy = TK[4], cy = TK[5];
    const float xl = (u - cx + cy * sk / fy - sk * v / fy) / fx * z;
    const float yl = (v - cy) / fy * z;
    float w[4], cam[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = T[r * 4 + 0] * xl + T[r * 4 + 1] * yl + T[r * 4 + 2] * z + T[r * 4 + 3];
#pragma unroll
    for (int r = 0; r < 3; ++r) cam[r] = SI[r * 4 + 0] * w[0] + SI[r * 4 + 1] * w[1] + SI[r * 4 + 2] * w[2] + SI[r * 4 + 3] * w[3];
    fx = SK[0]; sk = SK[1]; cx = SK[2]; fy = SK[4]; cy = SK[5];
    const float zc = cam[2];
    const float yc = cam[1] / zc * fy + cy;
    const float xc = cam[0] / zc * fx + sk * yc / fy - cy * sk / fy + cx;
    const float gx = 2.f * xc - 1.f, gy = 2.f * yc - 1.f;
    const float inside = (gx < -1.f || gx > 1.f || gy < -1.f || gy > 1.f) ? 0.f : 1.f;
    const Bilin b = bilin_setup(gx, gy, res);
    float dsrc = 0.f;
    if (b.v00) dsrc += b.w00 * resized_depth(sd, dres, res, b.y0, b.x0);
    if (b.v01) dsrc += b.w01 * resized_depth(sd, dres, res, b.y0, b.x0 + 1);
    if (b.v10) dsrc += b.w10 * resized_depth(sd, dres, res, b.y0 + 1, b.x0);
    if (b.v11) dsrc += b.w11 * resized_depth(sd, dres, res, b.y0 + 1, b.x0 + 1);
    float m = (fabsf(dsrc - zc) < eps) ? inside : 0.f;
    float m2 = 1.f;
    if (src_mask) {
        const float* sm = src_mask + (int64_t)n * plane;
        m2 = 0.f;
        if (b.v00) m2 += b.w00 * sm[(int64_t)b.y0 * res + b.x0];
        if (b.v01) m2 += b.w01 * sm[(int64_t)b.y0 * res + b.x0 + 1];
        if (b.v10) m2 += b.w10 * sm[(int64_t)(b.y0 + 1) * res + b.x0];
        if (b.v11) m2 += b.w11 * sm[(int64_t)(b.y0 + 1) * res + b.x0 + 1];
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float* im = src_image + ((int64_t)n * 3 + ch) * plane;
        float s = 0.f;
        if (b.v00) s += b.w00 * im[(int64_t)b.y0 * res + b.x0];
        if (b.v01) s += b.w01 * im[(int64_t)b.y0 * res + b.x0 + 1];
        if (b.v10) s += b.w10 * im[(int64_t)(b.y0 + 1) * res + b.x0];
        if (b.v11) s += b.w11 * im[(int64_t)(b.y0 + 1) * res + b.x0 + 1];
        s = s * m;
        if (src_mask) s = s * m2;
        warp_rgb[((int64_t)n * 3 + ch) * plane + rem] = s;
    }
    warp_mask[g] = src_mask ? m * m2 : m;
}

// ------------------------------------------------------------------------------------------------
// Affine (style) layers of the generator at inversion batch sizes (networks_stylegan2.py:95-127 FullyConnectedLayer, activation linear):
//     y[n,o] = b[o] + gain * sum_i x[n,i] W[o,i],   N <= 8 rows (1 image, or the 4 pseudo-views).
// A library GEMM runs these [1..4, 512] x [512, O] products on ONE workgroup (18 us forward, 2 x 7.5 us backward, 29 layers per generator
// pass: 2 % of the step's GPU time); they are matrix-VECTOR products bound by reading W once.
//   forward : one wave per output row o, lanes stride over i with float4 loads, DPP wave sums          (O/4 blocks)
//   backward: ONE launch with two kinds of blocks (affine_bwd_kernel): 64-column blocks for dx = gain * g W (W read once, 16 waves reduced
//             through LDS) and elementwise blocks for the outer product dW = gain * g^T x
// ------------------------------------------------------------------------------------------------
constexpr int AFF_NMAX = 8;

__global__ void __launch_bounds__(256) affine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                         float gain, float* __restrict__ y, int N, int I, int O) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= O) return;                                      // wave-uniform
    float acc[AFF_NMAX];
#pragma unroll
    for (int n = 0; n < AFF_NMAX; ++n) acc[n] = 0.f;
    const float4* w4 = reinterpret_cast<const float4*>(w + (int64_t)o * I);
    for (int i4 = lane; i4 < I / 4; i4 += 64) {
        const float4 wv = w4[i4];
#pragma unroll
        for (int n = 0; n < AFF_NMAX; ++n) {
            if (n < N) {
                const float4 xv = reinterpret_cast<const float4*>(x + (int64_t)n * I)[i4];
                acc[n] = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, acc[n]))));
            }
        }
    }
    const float bo = b ? b[o] : 0.f;
#pragma unroll
    for (int n = 0; n < AFF_NMAX; ++n) {
        if (n < N) {                                         // N is a kernel argument: uniform
            const float s = wave_sum(acc[n]);
            if (lane == 0) y[(int64_t)n * O + o] = fmaf(gain, s, bo);
        }
    }
}

__global__ void __launch_bounds__(1024) affine_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ w,
                                                          float gain, float* __restrict__ dx, float* __restrict__ dw, int N, int I, int O, int dx_blocks) {
    // Two kinds of blocks in ONE launch (neither gradient needs the other's operand):
    //   blockIdx.x <  dx_blocks: dx[n, c] = gain * sum_o g[n,o] W[o,c] for 64 columns c; wave w walks rows o = w, w + 16, ... 8 at a time with
    //                            their loads issued together (a 512-row W is two round trips per wave), the 16 waves reduce through LDS
    //   blockIdx.x >= dx_blocks: dW[o, i] = gain * sum_n g[n,o] x[n,i], four elements per thread (a pure outer product: 4 I O bytes written)
    __shared__ float red[16][AFF_NMAX][64];
    if ((int)blockIdx.x >= dx_blocks) {
        const int64_t e = ((int64_t)(blockIdx.x - dx_blocks) * 1024 + threadIdx.x) * 4;
        if (e >= (int64_t)O * I) return;
        const int o = (int)(e / I), i = (int)(e - (int64_t)o * I);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int n = 0; n < AFF_NMAX; ++n) {
            if (n < N) {
                const float gv = g[(int64_t)n * O + o];
                const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)n * I + i);
                acc.x = fmaf(gv, xv.x, acc.x); acc.y = fmaf(gv, xv.y, acc.y); acc.z = fmaf(gv, xv.z, acc.z); acc.w = fmaf(gv, xv.w, acc.w);
            }
        }
        *reinterpret_cast<float4*>(dw + e) = make_float4(gain * acc.x, gain * acc.y, gain * acc.z, gain * acc.w);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int cc = min(c, I - 1);
    float acc[AFF_NMAX];
#pragma unroll
    for (int n = 0; n < AFF_NMAX; ++n) acc[n] = 0.f;
    for (int o0 = wave; o0 < O; o0 += 16 * 8) {
        float wv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wv[k] = w[(int64_t)min(o0 + 16 * k, O - 1) * I + cc];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int o = o0 + 16 * k;
            if (o < O) {                                         // wave-uniform
#pragma unroll
                for (int n = 0; n < AFF_NMAX; ++n)
                    if (n < N) acc[n] = fmaf(g[(int64_t)n * O + o], wv[k], acc[n]);      // g: wave-uniform address, one scalar load
            }
        }
    }
#pragma unroll
    for (int n = 0; n < AFF_NMAX; ++n) red[wave][n][lane] = acc[n];
    __syncthreads();
    for (int t = threadIdx.x; t < N * 64; t += 1024) {
        const int n = t >> 6, l = t & 63;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][n][l];
        const int col = blockIdx.x * 64 + l;
        if (col < I) dx[(int64_t)n * I + col] = gain * s;
    }
}

// ---- all affine layers of a network in one launch each way: the same two kernels over a table of jobs (by value: <= 32 x 64 B of kernel arguments)
struct AffineJobs { int n; int blk0[SPI_AFFINE_MAX_JOBS + 1]; spi_affine_job job[SPI_AFFINE_MAX_JOBS]; };

__device__ __forceinline__ int affine_find_job(const AffineJobs& J, int blk) {      // block-uniform: the job whose block range holds blk
    int j = 0;
    while (j + 1 < J.n && blk >= J.blk0[j + 1]) ++j;
    return j;
}

__global__ void __launch_bounds__(256) affine_multi_fwd_kernel(AffineJobs J, int N, int I, int64_t xs) {
    const int j = affine_find_job(J, blockIdx.x);
    const spi_affine_job& q = J.job[j];
    const int lane = threadIdx.x & 63;
    const int o = ((int)blockIdx.x - J.blk0[j]) * 4 + (threadIdx.x >> 6);
    if (o >= q.O) return;                                    // wave-uniform
    float acc[AFF_NMAX];
#pragma unroll
    for (int n = 0; n < AFF_NMAX; ++n) acc[n] = 0.f;
    const float4* w4 = reinterpret_cast<const float4*>(q.w + (int64_t)o * I);
    for (int i4 = lane; i4 < I / 4; i4 += 64) {
        const float4 wv = w4[i4];
#pragma unroll
        for (int n = 0; n < AFF_NMAX; ++n) {
            if (n < N) {
                const float4 xv = reinterpret_cast<const float4*>(q.x + (int64_t)n * xs)[i4];
                acc[n] = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, acc[n]))));      // same chain as affine_fwd_kernel: bit-equal
            }
        }
    }
    const float bo = q.b ? q.b[o] : 0.f;
#pragma unroll
    for (int n = 0; n < AFF_NMAX; ++n) {
        if (n < N) {
            const float s = wave_sum(acc[n]);
            if (lane == 0) q.y[(int64_t)n * q.O + o] = fmaf(q.gain, s, bo);
        }
    }
}

// per job: ceil(I / 64) blocks for dx (if wanted), then ceil(O * I / 4096) blocks for dw (if wanted); blk0[] holds the running totals
__global__ void __launch_bounds__(1024) affine_multi_bwd_kernel(AffineJobs J, int N, int I, int64_t xs) {
    __shared__ float red[16][AFF_NMAX][64];
    const int j = affine_find_job(J, blockIdx.x);
    const spi_affine_job& q = J.job[j];
    const int O = q.O;
    const float gain = q.gain;
    const float* __restrict__ g = q.g;
    const int local = (int)blockIdx.x - J.blk0[j];
    const int dx_blocks = q.dx_acc ? (I + 63) / 64 : 0;
    if (local >= dx_blocks) {
        const int64_t e = ((int64_t)(local - dx_blocks) * 1024 + threadIdx.x) * 4;
        if (e >= (int64_t)O * I) return;
        const int o = (int)(e / I), i = (int)(e - (int64_t)o * I);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int n = 0; n < AFF_NMAX; ++n) {
            if (n < N) {
                const float gv = g[(int64_t)n * O + o];
                const float4 xv = *reinterpret_cast<const float4*>(q.x + (int64_t)n * xs + i);
                acc.x = fmaf(gv, xv.x, acc.x); acc.y = fmaf(gv, xv.y, acc.y); acc.z = fmaf(gv, xv.z, acc.z); acc.w = fmaf(gv, xv.w, acc.w);
            }
        }
        *reinterpret_cast<float4*>(q.dw + e) = make_float4(gain * acc.x, gain * acc.y, gain * acc.z, gain * acc.w);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = local * 64 + lane;
    const int cc = min(c, I - 1);
    float acc[AFF_NMAX];
#pragma unroll
    for (int n = 0; n < AFF_NMAX; ++n) acc[n] = 0.f;
    for (int o0 = wave; o0 < O; o0 += 16 * 8) {
        float wv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wv[k] = q.w[(int64_t)min(o0 + 16 * k, O - 1) * I + cc];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int o = o0 + 16 * k;
            if (o < O) {
#pragma unroll
                for (int n = 0; n < AFF_NMAX; ++n)
                    if (n < N) acc[n] = fmaf(g[(int64_t)n * O + o], wv[k], acc[n]);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < AFF_NMAX; ++n) red[wave][n][lane] = acc[n];
    __syncthreads();
    for (int t = threadIdx.x; t < N * 64; t += 1024) {
        const int n = t >> 6, l = t & 63;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][n][l];
        const int col = local * 64 + l;
        if (col < I) atomicAdd(q.dx_acc + (int64_t)n * xs + col, gain * s);      // <= 2 jobs share a row of ws: a + b = b + a, the sum does not depend on the order
    }
}

// ------------------------------------------------------------------------------------------------
// Stage-1 noise regulariser: one 1024-thread block per noise buffer walks the whole pooling pyramid
// (<= 6 levels of a 256^2 buffer) with block-level reductions; pooled levels live in an L2-resident
// scratch pyramid so the backward can reuse them.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 block_sum2_1024(float a, float b, float (*red)[2]) {
    a = wave_sum(a); b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { sa += red[i][0]; sb += red[i][1]; }
    return make_float2(sa, sb);
}

__global__ void __launch_bounds__(1024) noise_reg_fwd_kernel(const float* const* __restrict__ bufs, const int32_t* __restrict__ res,
                                                             int64_t region, float* __restrict__ pyramid, float* __restrict__ means,
                                                             float* __restrict__ loss) {
    __shared__ float red[16][2];
    const int t = blockIdx.x, tid = threadIdx.x;
    int R = res[t];
    const float* x = bufs[t];
    float* pyr = pyramid + (int64_t)t * region;
    float total = 0.f;
    for (int lvl = 0;; ++lvl) {
        const int n = R * R;
        float sh = 0.f, sv = 0.f;
        for (int i = tid; i < n; i += 1024) {
            const int y = i / R, xx = i - y * R;
            const float v = x[i];
            sh = fmaf(v, x[y * R + (xx ? xx - 1 : R - 1)], sh);                  // roll(shifts=1, dims=3): out[x] = in[x-1]
            sv = fmaf(v, x[(y ? y - 1 : R - 1) * R + xx], sv);
        }
        const float2 s2 = block_sum2_1024(sh, sv, red);
        const float mh = s2.x / (float)n, mv = s2.y / (float)n;
        if (tid == 0) { means[(t * 8 + lvl) * 2] = mh; means[(t * 8 + lvl) * 2 + 1] = mv; }
        total += mh * mh + mv * mv;
        if (R <= 8) break;
        const int R2 = R >> 1;
        for (int i = tid; i < R2 * R2; i += 1024) {
            const int Y = i / R2, X = i - Y * R2;
            const float* q = x + (2 * Y) * R + 2 * X;
            pyr[i] = ((q[0] + q[1]) + (q[R] + q[R + 1])) * 0.25f;               // avg_pool2d(kernel_size=2)
        }
        __threadfence_block();
        __syncthreads();
        x = pyr; pyr += R2 * R2; R = R2;
    }
    if (tid == 0) atomicAdd(loss, total);
}

__global__ void __launch_bounds__(1024) noise_reg_bwd_kernel(const float* const* __restrict__ bufs, const int32_t* __restrict__ res,
                                                             int64_t region, const float* __restrict__ pyramid,
                                                             const float* __restrict__ means, const float* __restrict__ gout,
                                                             float* __restrict__ grads, const int64_t* __restrict__ goff,
                                                             float* __restrict__ gpyramid) {
    const int t = blockIdx.x, tid = threadIdx.x;
    const int R0 = res[t];
    int L = 1;
    for (int r = R0; r > 8; r >>= 1) ++L;
    const float c = gout[0];
    // level l >= 1 starts at offset sum_{k=1}^{l-1} (R0 >> k)^2 of the buffer's region (same for x and G pyramids)
    for (int l = L - 1; l >= 0; --l) {
        const int R = R0 >> l, n = R * R;
        int64_t off = 0, offp = 0;
        for (int k = 1; k < l; ++k) off += (int64_t)(R0 >> k) * (R0 >> k);
        for (int k = 1; k < l + 1; ++k) offp += (int64_t)(R0 >> k) * (R0 >> k);
        const float* x = l == 0 ? bufs[t] : pyramid + (int64_t)t * region + off;
        float* G = l == 0 ? grads + goff[t] : gpyramid + (int64_t)t * region + off;
        const float* Gp = l == L - 1 ? nullptr : gpyramid + (int64_t)t * region + offp;
        const float a = 2.f * means[(t * 8 + l) * 2] / (float)n * c, b = 2.f * means[(t * 8 + l) * 2 + 1] / (float)n * c;
        for (int i = tid; i < n; i += 1024) {
            const int y = i / R, xx = i - y * R;
            const float hl = x[y * R + (xx ? xx - 1 : R - 1)], hr = x[y * R + (xx + 1 < R ? xx + 1 : 0)];
            const float vu = x[(y ? y - 1 : R - 1) * R + xx], vd = x[(y + 1 < R ? y + 1 : 0) * R + xx];
            float g = a * (hl + hr) + b * (vu + vd);
            if (Gp) g = fmaf(0.25f, Gp[(y >> 1) * (R >> 1) + (xx >> 1)], g);
            G[i] = g;
        }
        __threadfence_block();
        __syncthreads();
    }
}

__global__ void __launch_bounds__(1024) noise_renorm_kernel(float* const* __restrict__ bufs, const int32_t* __restrict__ res) {
    __shared__ float red[16][2];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int n = res[t] * res[t];
    float* x = bufs[t];
    float s = 0.f;
    for (int i = tid; i < n; i += 1024) s += x[i];
    const float mu = block_sum2_1024(s, 0.f, red).x / (float)n;
    float q = 0.f;
    for (int i = tid; i < n; i += 1024) { const float v = x[i] - mu; x[i] = v; q = fmaf(v, v, q); }     // each thread re-reads only its own elements
    const float r = rsqrtf(block_sum2_1024(q, 0.f, red).x / (float)n);
    for (int i = tid; i < n; i += 1024) x[i] *= r;
}

// ------------------------------------------------------------------------------------------------
// LPIPS tail (lpips.py:43-65): unit-normalise both feature stacks over channels, squared
// difference, 1x1 "lin" weights, spatial mean.  A 1024-thread block covers 64 pixels (the lanes:
// channel reads stay coalesced in NCHW) x 16 channel groups (the waves); channel sums are combined
// through LDS.  The deep layers (16^2 x 512 ch) have few pixels, so the parallelism has to come from
// the channel axis (a thread-per-pixel version ran 300 us on one block for 1 MB of input).
// ------------------------------------------------------------------------------------------------
constexpr int LP_CG = 16;
__device__ __forceinline__ float lp_cross(float v, float (*red)[64], int px, int cg) {      // sum over the 16 channel groups
    __syncthreads();
    red[cg][px] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < LP_CG; ++g) s += red[g][px];
    return s;
}

__global__ void __launch_bounds__(1024) lpips_fwd_kernel(const float* __restrict__ fx, const float* __restrict__ fy,
                                                         const float* __restrict__ lin, int C, int64_t HW, float* __restrict__ out) {
    __shared__ float red[LP_CG][64];
    const int n = blockIdx.y, px = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * 64 + px;
    const bool ok = p < HW;
    const float* a = fx + (int64_t)n * C * HW + (ok ? p : 0); const float* b = fy + (int64_t)n * C * HW + (ok ? p : 0);
    float sa = 0.f, sb = 0.f;
#pragma unroll 8
    for (int c = cg; c < C; c += LP_CG) { const float va = a[c * HW], vb = b[c * HW]; sa = fmaf(va, va, sa); sb = fmaf(vb, vb, sb); }
    sa = lp_cross(sa, red, px, cg); sb = lp_cross(sb, red, px, cg);
    const float na = sqrtf(sa) + 1e-10f, nb = sqrtf(sb) + 1e-10f;
    // x / (|x| + eps) as x * (1 / (|x| + eps)): two divisions per PIXEL instead of two per element (an IEEE division is ~10 VALU instructions; the
    // results differ by at most an ulp per operand)
    const float ina = 1.f / na, inb = 1.f / nb;
    float val = 0.f;
#pragma unroll 8
    for (int c = cg; c < C; c += LP_CG) { const float d = a[c * HW] * ina - b[c * HW] * inb; val = fmaf(lin[c], d * d, val); }
    val = wave_sum(ok ? val : 0.f);
    __syncthreads();
    if (px == 0) red[0][cg] = val;                             // same-address global atomics serialise (~10 ns each): one per block
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < LP_CG; ++g) tot += red[0][g];
        atomicAdd(out + n, tot / (float)HW);
    }
}

__global__ void __launch_bounds__(1024) lpips_bwd_kernel(const float* __restrict__ fx, const float* __restrict__ fy,
                                                         const float* __restrict__ lin, const float* __restrict__ d_out, int C,
                                                         int64_t HW, float* __restrict__ d_fx) {
    __shared__ float red[LP_CG][64];
    const int n = blockIdx.y, px = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * 64 + px;
    const bool ok = p < HW;
    const float* a = fx + (int64_t)n * C * HW + (ok ? p : 0); const float* b = fy + (int64_t)n * C * HW + (ok ? p : 0);
    float* o = d_fx + (int64_t)n * C * HW + p;
    float sa = 0.f, sb = 0.f;
#pragma unroll 8
    for (int c = cg; c < C; c += LP_CG) { const float va = a[c * HW], vb = b[c * HW]; sa = fmaf(va, va, sa); sb = fmaf(vb, vb, sb); }
    sa = lp_cross(sa, red, px, cg); sb = lp_cross(sb, red, px, cg);
    const float ra = sqrtf(sa), na = ra + 1e-10f, nb = sqrtf(sb) + 1e-10f;
    const float gsc = d_out[n] / (float)HW;
    const float ina = 1.f / na, inb = 1.f / nb;
    float dot = 0.f;        // sum_c 2 lin_c (a_c - b_c) fx_c
#pragma unroll 8
    for (int c = cg; c < C; c += LP_CG) { const float va = a[c * HW]; const float d = va * ina - b[c * HW] * inb; dot = fmaf(2.f * lin[c] * d, va, dot); }
    dot = lp_cross(dot, red, px, cg);
    const float k2 = (ra > 0.f) ? dot / (ra * na * na) : 0.f;
    if (!ok) return;
#pragma unroll 8
    for (int c = cg; c < C; c += LP_CG) {
        const float va = a[c * HW]; const float d = va * ina - b[c * HW] * inb;
        o[c * HW] = gsc * (2.f * lin[c] * d * ina - va * k2);
    }
}

// ------------------------------------------------------------------------------------------------
// torch.optim.Adam (single-tensor semantics of torch 2.x: lerp first moment, addcmul second,
// denom = sqrt(v)/sqrt(bc2) + eps, p -= lr/bc1 * m/denom) for T tensors in one launch.
// ------------------------------------------------------------------------------------------------
// hyper != NULL: {lr, bc1, bc2_sqrt} are read from device memory (a captured HIP graph replays the launch with new values each step)
// skip != NULL: a device byte; non-zero -> the launch changes nothing (the early-stop decision of a loop that runs ahead of its host, see
// spi_adam_multi_pred)
__global__ void adam_multi_kernel(void* const* __restrict__ ptrs, const int64_t* __restrict__ sizes, float lr, float beta1,
                                  float beta2, float eps, float bc1, float bc2_sqrt, const float* __restrict__ hyper,
                                  const unsigned char* __restrict__ skip) {
    if (skip && *skip) return;
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; }
    const int t = blockIdx.y;
    const int64_t n = sizes[t];
    float* p = static_cast<float*>(ptrs[4 * t + 0]);
    const float* g = static_cast<const float*>(ptrs[4 * t + 1]);
    float* m = static_cast<float*>(ptrs[4 * t + 2]);
    float* v = static_cast<float*>(ptrs[4 * t + 3]);
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * (1.f - beta1);
        vi = vi * beta2 + gi * gi * (1.f - beta2);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
        m[i] = mi; v[i] = vi;
    }
}

// ================================================================================================
// flags[n, s] = any(x[n, :, 16 s .. 16 s + 15] != 0): one block per 1024 pixels of one sample, lanes along pixels (coalesced rows),
// every thread ORs the bit patterns of its 4 pixels over all channels, segments are combined through LDS.
template <typename T = float>
__global__ void __launch_bounds__(256) seg_flags_kernel(const T* __restrict__ x, int32_t* __restrict__ flags, int C, int64_t HW, int nseg) {
    using U = std::conditional_t<sizeof(T) == 2, unsigned short, unsigned>;      // the element's bit pattern
    __shared__ int s_f[64];
    const int tid = threadIdx.x, n = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * 1024;
    if (tid < 64) s_f[tid] = 0;
    __syncthreads();
    const U* xb = reinterpret_cast<const U*>(x) + (int64_t)n * C * HW;
    int64_t pix[4]; bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int64_t p = base + tid + 256 * k; ok[k] = p < HW; pix[k] = ok[k] ? p : HW - 1; }
    unsigned acc[4] = {0u, 0u, 0u, 0u};
    int c = 0;
    for (; c + 4 <= C; c += 4) {
        unsigned v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = xb[(int64_t)(c + u) * HW + pix[k]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] |= v[u][k];
    }
    for (; c < C; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] |= xb[(int64_t)c * HW + pix[k]];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (ok[k] && (acc[k] & (sizeof(T) == 2 ? 0x7fffu : 0x7fffffffu)) != 0u) s_f[(tid + 256 * k) >> 4] = 1;      // sign bit masked: -0.0 counts as zero
    __syncthreads();
    const int seg = blockIdx.x * 64 + tid;
    if (tid < 64 && seg < nseg) flags[(int64_t)n * nseg + seg] = s_f[tid];
}

// ds[s,i] = A_i / st[s,i] - st[s,i] g^2 sum_o dcoef[s,o]^2 C_o ww[o,i]   (frozen-weight style gradient, see spi_style_grad)
__global__ void __launch_bounds__(256) style_grad_kernel(const float* __restrict__ a, const float* __restrict__ cv, const float* __restrict__ st,
                                                         const float* __restrict__ dcoef, const float* __restrict__ ww, float* __restrict__ ds,
                                                         int N, int NS, int I, int O, float g2) {
    // block: 32 channels x 8 slices of the o-sum (a [O, I] GEMV with 2 blocks would crawl: 37 us; this shape: ~6 us)
    extern __shared__ float coef[];                       // [O]: dcoef[s,o]^2 * C_o, then [8][32] partial sums
    float* part = coef + O;
    const int s = blockIdx.y, cx = threadIdx.x & 31, og = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cx;
    const bool shared = (NS == 1 && N > 1);
    if (cv) {
        for (int o = threadIdx.x; o < O; o += 256) {
            float c = 0.f;
            if (shared) for (int n = 0; n < N; ++n) c += cv[(int64_t)n * O + o];
            else c = cv[(int64_t)s * O + o];
            const float d = dcoef[(int64_t)s * O + o];
            coef[o] = d * d * c;
        }
        __syncthreads();
        float acc = 0.f;
        if (i < I) {
#pragma unroll 8
            for (int o = og; o < O; o += 8) acc = fmaf(coef[o], ww[(int64_t)o * I + i], acc);
        }
        part[og * 32 + cx] = acc;
        __syncthreads();
    }
    if (og != 0 || i >= I) return;
    float av = 0.f;
    if (shared) for (int n = 0; n < N; ++n) av += a[(int64_t)n * I + i];
    else av = a[(int64_t)s * I + i];
    const float sv = st[(int64_t)s * I + i];
    float r = fabsf(sv) > 1e-20f ? av / sv : 0.f;
    if (cv) {
        float acc = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) acc += part[g * 32 + cx];
        r -= sv * g2 * acc;
    }
    ds[(int64_t)s * I + i] = r;
}

// ------------------------------------------------------------------------------------------------
// filtered_lrelu_act_ (filtered_lrelu.cpp:217-296, filtered_lrelu.cu:1109-1215): gain -> leaky ReLU -> clamp IN PLACE on the upsampled
// tensor, with the reference's bit-packed sign tensor: 2 bits per element (1 = negative, 2 = clamped), 4 elements per byte,
// uint8 [N*C, sH, sW/4].  mode 0: plain forward; 1: forward + WRITE signs; 2: READ signs at offset (sx, sy) -- the gradient
// pass, v = v * gain * (slope if negative) * (0 if clamped).  One thread per sign byte (4 consecutive elements of a row).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) filtered_lrelu_act_kernel(float* __restrict__ x, uint8_t* __restrict__ s, int64_t NC, int xH, int xW, int sH,
                                                                 int sW, int sx, int sy, float gain, float slope, float clamp, int mode) {
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int ymax = mode == 1 ? max(sH, xH) : xH;
    const int wmax = mode == 1 ? max(sW, xW) : xW;
    if (x4 >= wmax) return;
    for (int64_t q = blockIdx.z; q < NC; q += gridDim.z)
        for (int y = blockIdx.y; y < ymax; y += gridDim.y) {
            uint32_t bits = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = x4 + j;
                if (xx >= xW || y >= xH) continue;
                float* pv = x + (q * xH + y) * (int64_t)xW + xx;
                float v = *pv * gain;
                if (mode == 2) {
                    const uint32_t ux = (uint32_t)(xx + sx), uy = (uint32_t)(y + sy);
                    if (ux < (uint32_t)sW && uy < (uint32_t)sH) {
                        const uint32_t sb = s[(q * sH + uy) * (int64_t)(sW >> 2) + (ux >> 2)] >> ((ux & 3) << 1);
                        if (sb & 1) v *= slope;
                        if (sb & 2) v = 0.f;
                    }
                } else {
                    uint32_t sg = 0;
                    if (v < 0.f) { v *= slope; sg = 1; }
                    if (fabsf(v) > clamp) { v = fminf(fmaxf(v, -clamp), clamp); sg = 2; }
                    bits |= sg << (j << 1);
                }
                *pv = v;
            }
            if (mode == 1 && x4 < sW && y < sH) s[(q * sH + y) * (int64_t)(sW >> 2) + (x4 >> 2)] = (uint8_t)bits;
        }
}

extern "C" {

int spi_bias_act(const float* x, const float* b, const float* xref, const float* yref, const float* dy, float* y, int64_t n,
                 int sizeB, int64_t stepB, int grad, int act, float alpha, float gain, float clamp, spi_stream_t stream) {
    SPI_REQUIRE(x && y && n > 0, "spi_bias_act: null tensor or empty");
    SPI_REQUIRE(act >= SPI_ACT_LINEAR && act <= SPI_ACT_SWISH, "spi_bias_act: unknown activation %d", act);
    SPI_REQUIRE(grad >= 0 && grad <= 2, "spi_bias_act: grad must be 0, 1 or 2");
    SPI_REQUIRE(b == nullptr || (sizeB > 0 && stepB > 0), "spi_bias_act: bias given without sizeB/stepB");
    ActParams p{act, grad, alpha, gain, clamp};
    const bool aligned = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)xref | (uintptr_t)yref | (uintptr_t)dy) & 15) == 0;
    const bool vec = aligned && (n % 4 == 0) && (b == nullptr || stepB % 4 == 0);
    const int64_t work = vec ? n / 4 : n;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(work, 256), 256 * 16);
    if (vec) hipLaunchKernelGGL(bias_act_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), x, b, xref, yref, dy, y, n, sizeB, stepB, p);
    else hipLaunchKernelGGL(bias_act_kernel<false>, dim3(grid), dim3(256), 0, as_stream(stream), x, b, xref, yref, dy, y, n, sizeB, stepB, p);
    SPI_LAUNCH_CHECK("spi_bias_act");
    return SPI_OK;
}

static int launch_upfirdn(const float* x, const float* f, float* y, const UpfirdnParams& p, const float* pre_bias,
                          const float* noise, const float* noise_gain, const float* bias, const ActParams& ap, spi_stream_t stream) {
    const int64_t total = (int64_t)p.N * p.C * p.outH * p.outW;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(total, 256), 256 * 32);
    const bool small = (int64_t)p.N * p.C * p.inH * p.inW * 4 < (int64_t)0x7fffffff;     // one buffer descriptor spans the input
    const bool k44 = p.fH == 4 && p.fW == 4 && p.upx == p.upy && p.downx == p.downy && p.upx <= 2 && p.downx <= 2 && !(p.upx == 2 && p.downx == 2);
    if (k44 && p.upx == 1 && p.downx == 1 && p.outW >= 100 && p.outH >= 100 && (ap.act == 0 || ap.act == SPI_ACT_LINEAR || ap.act == SPI_ACT_LRELU) && (int64_t)p.N * p.C <= 65535) {
        const dim3 g((unsigned)((p.outW + FT_W - 1) / FT_W), (unsigned)((p.outH + FT_H - 1) / FT_H), (unsigned)(p.N * p.C));
        hipLaunchKernelGGL(upfirdn2d_4x4_tiled_kernel<float>, g, dim3(256), 0, as_stream(stream), x, f, y, p, pre_bias, noise, noise_gain, bias, ap);
    } else if (k44 && small && p.upx == 1 && p.downx == 1) {
        const unsigned g4 = (unsigned)std::min<int64_t>(ceil_div64((total + 3) / 4 + (int64_t)p.N * p.C * p.outH, 256), 256 * 32);
        hipLaunchKernelGGL((upfirdn2d_4x4_kernel<1, 1>), dim3(g4), dim3(256), 0, as_stream(stream), x, f, y, p, pre_bias, noise, noise_gain, bias, ap);
    } else if (k44 && small && p.upx == 2) {
        hipLaunchKernelGGL((upfirdn2d_4x4_kernel<2, 1>), dim3(grid), dim3(256), 0, as_stream(stream), x, f, y, p, pre_bias, noise, noise_gain, bias, ap);
    } else if (k44 && small && p.downx == 2) {
        hipLaunchKernelGGL((upfirdn2d_4x4_kernel<1, 2>), dim3(grid), dim3(256), 0, as_stream(stream), x, f, y, p, pre_bias, noise, noise_gain, bias, ap);
    } else
    hipLaunchKernelGGL((upfirdn2d_kernel<float, float>), dim3(grid), dim3(256), 0, as_stream(stream), x, f, y, p, pre_bias, noise, noise_gain, bias, ap);
    SPI_LAUNCH_CHECK("spi_upfirdn2d");
    return SPI_OK;
}

extern "C++" {
template <typename T>
static int tail_bwd_launch(const T* dy, const T* y, T* dz, float* d_bias, float* d_pixsum, const float* noise, float* d_strength, int N, int C,
                           int64_t HW, int act, float alpha, float gain, float clamp, spi_stream_t stream, ZDot zd = ZDot{nullptr, nullptr, nullptr, nullptr}) {
    SPI_REQUIRE(!zd.out || y, "spi_tail_bwd_dot: the dot product with the reconstructed conv result needs the saved output y");
    SPI_REQUIRE(dy && N > 0 && C > 0 && HW > 0, "spi_tail_bwd: bad argument");
    SPI_REQUIRE(!d_strength || (noise && d_pixsum), "spi_tail_bwd: d_strength needs noise and d_pixsum");
    SPI_REQUIRE(!y || (act >= SPI_ACT_LINEAR && act <= SPI_ACT_LRELU && gain != 0.f), "spi_tail_bwd: activation must be linear / relu / lrelu");
    SPI_REQUIRE(y || !dz, "spi_tail_bwd: dz without a saved output (dz == dy)");
    ActParams ap{act, 1, alpha, gain, clamp};
    const bool vec = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dz)) % (4 * sizeof(T)) == 0);
    const int64_t per_block = vec ? 1024 : 256;
    const unsigned gx = (unsigned)ceil_div64(HW, per_block);
    // channel splits: enough blocks (~1024) to fill 256 CUs.  With the per-pixel sums every split adds ONE atomic per pixel address (HW * splits
    // atomics in all, each address touched `splits` times over the whole kernel: no hot spot) -- capped at 16.  (Until round 2 the cap was
    // 128 / gx, i.e. 128-256 blocks on the 256^2 / 512^2 layers: 1.3-2.8 TB/s against 4.1-5.6 without the sums.)
    // Round 3: the cap follows the plane size -- 16 from 128^2 up, up to 64 on the 4^2 ... 64^2 layers, whose single column of 16 blocks
    // took 24 us for 2 MB (224 launches of the profiled run).
    const int64_t pix_cap = std::min<int64_t>(64, std::max<int64_t>(16, 262144 / HW));
    int splits = (int)std::min<int64_t>(C, std::max<int64_t>((C + 511) / 512, std::min<int64_t>(1024 / gx, d_pixsum ? pix_cap : 1024)));
    if (zd.out) splits = std::max(splits, (int)(((int64_t)C * N + 2047) / 2048));   // ZDot: cchunk * N <= 2048 (LDS partials per (n, c))
    SPI_REQUIRE(!zd.out || N <= 2048, "spi_tail_bwd_dot: batch too large");
    // (the dot product reconstructs the conv result from y by dividing negative values by the slope: a leaky ReLU with slope 0 would form 0 * inf.
    //  ReLU is fine -- dz is exactly 0 wherever y cannot be inverted)
    SPI_REQUIRE(!zd.out || act != SPI_ACT_LRELU || alpha != 0.f, "spi_tail_bwd_dot: a leaky ReLU output with slope 0 cannot be inverted");
    int cchunk = (C + splits - 1) / splits;                           // <= 512 (LDS partials)
    if (zd.out) cchunk = std::min(cchunk, std::max(1, 2048 / N));    // ceil(C / ceil(C N / 2048)) can exceed 2048 / N (C = 5, N = 1500): clamp, then recount
    splits = (C + cchunk - 1) / cchunk;
    if (zd.out) {
        if (vec) hipLaunchKernelGGL((tail_bwd_kernel<4, T, true>), dim3(gx, (unsigned)splits), dim3(256), 0, as_stream(stream), dy, y, dz, d_bias, d_pixsum, noise, d_strength, N, C, HW, cchunk, ap, zd);
        else hipLaunchKernelGGL((tail_bwd_kernel<1, T, true>), dim3(gx, (unsigned)splits), dim3(256), 0, as_stream(stream), dy, y, dz, d_bias, d_pixsum, noise, d_strength, N, C, HW, cchunk, ap, zd);
    } else if (vec) hipLaunchKernelGGL((tail_bwd_kernel<4, T>), dim3(gx, (unsigned)splits), dim3(256), 0, as_stream(stream), dy, y, dz, d_bias, d_pixsum, noise, d_strength, N, C, HW, cchunk, ap, zd);
    else hipLaunchKernelGGL((tail_bwd_kernel<1, T>), dim3(gx, (unsigned)splits), dim3(256), 0, as_stream(stream), dy, y, dz, d_bias, d_pixsum, noise, d_strength, N, C, HW, cchunk, ap, zd);
    SPI_LAUNCH_CHECK("spi_tail_bwd");
    return SPI_OK;
}

}  // extern "C++"

int spi_tail_bwd(const float* dy, const float* y, float* dz, float* d_bias, float* d_pixsum, const float* noise, float* d_strength, int N, int C,
                 int64_t HW, int act, float alpha, float gain, float clamp, spi_stream_t stream) {
    return tail_bwd_launch<float>(dy, y, dz, d_bias, d_pixsum, noise, d_strength, N, C, HW, act, alpha, gain, clamp, stream);
}

// the typed variants (dtype: SPI_DTYPE_F32 / SPI_DTYPE_F16 = element type of the ACTIVATION tensors; sums, bias, noise stay fp32)
#define SPI_DTYPE_CHECK(who) do { if (dtype != SPI_DTYPE_F32 && dtype != SPI_DTYPE_F16) { spi_set_error(who ": dtype %d (0 = fp32, 1 = fp16)", dtype); return SPI_ERR_UNSUPPORTED; } } while (0)
int spi_tail_bwd_t(const void* dy, const void* y, void* dz, float* d_bias, float* d_pixsum, const float* noise, float* d_strength, int N, int C,
                   int64_t HW, int act, float alpha, float gain, float clamp, int dtype, spi_stream_t stream) {
    SPI_DTYPE_CHECK("spi_tail_bwd_t");
    if (dtype == SPI_DTYPE_F16)
        return tail_bwd_launch<_Float16>((const _Float16*)dy, (const _Float16*)y, (_Float16*)dz, d_bias, d_pixsum, noise, d_strength, N, C, HW, act, alpha, gain, clamp, stream);
    return tail_bwd_launch<float>((const float*)dy, (const float*)y, (float*)dz, d_bias, d_pixsum, noise, d_strength, N, C, HW, act, alpha, gain, clamp, stream);
}

int spi_tail_bwd_dot_t(const void* dy, const void* y, void* dz, float* d_bias, float* d_pixsum, const float* noise, float* d_strength, int N, int C,
                       int64_t HW, int act, float alpha, float gain, float clamp, const float* z_bias, const float* z_noise, const float* z_noise_gain,
                       float* zdot, int dtype, spi_stream_t stream) {
    SPI_DTYPE_CHECK("spi_tail_bwd_dot_t");
    SPI_REQUIRE(zdot != nullptr, "spi_tail_bwd_dot_t: null zdot (spi_tail_bwd_t is the call without the dot product)");
    const ZDot zd{zdot, z_bias, z_noise, z_noise_gain};
    if (dtype == SPI_DTYPE_F16)
        return tail_bwd_launch<_Float16>((const _Float16*)dy, (const _Float16*)y, (_Float16*)dz, d_bias, d_pixsum, noise, d_strength, N, C, HW, act, alpha, gain, clamp, stream, zd);
    return tail_bwd_launch<float>((const float*)dy, (const float*)y, (float*)dz, d_bias, d_pixsum, noise, d_strength, N, C, HW, act, alpha, gain, clamp, stream, zd);
}

extern "C++" {
template <typename T>
static int chan_dot_launch(const T* a, const T* b, float* out, int64_t rows, int C, int64_t HW, const float* bias, const float* noise,
                           const float* noise_gain, int act, float alpha, float gain, spi_stream_t stream) {
    SPI_REQUIRE(a && b && out && rows > 0 && rows < 65536 && C > 0 && HW > 0, "spi_chan_dot: bad argument");
    SPI_REQUIRE(act == 0 || ((act == SPI_ACT_LINEAR || act == SPI_ACT_LRELU) && gain != 0.f && (act != SPI_ACT_LRELU || alpha != 0.f)),
                "spi_chan_dot: only linear / lrelu outputs can be inverted");
    // ~1024 blocks in total, at least 2048 pixels per block
    const int64_t want = std::max<int64_t>(1, 1024 / rows);
    const int64_t per_block = std::max<int64_t>(4096, ((HW + want - 1) / want + 4095) / 4096 * 4096);
    dim3 grid((unsigned)ceil_div64(HW, per_block), (unsigned)rows);
    hipLaunchKernelGGL(chan_dot_kernel<T>, grid, dim3(256), 0, as_stream(stream), a, b, out, C, HW, per_block, bias, noise, noise_gain, act, alpha, gain);
    SPI_LAUNCH_CHECK("spi_chan_dot");
    return SPI_OK;
}

}  // extern "C++"

int spi_chan_dot(const float* a, const float* b, float* out, int64_t rows, int C, int64_t HW, const float* bias, const float* noise,
                 const float* noise_gain, int act, float alpha, float gain, spi_stream_t stream) {
    return chan_dot_launch<float>(a, b, out, rows, C, HW, bias, noise, noise_gain, act, alpha, gain, stream);
}

int spi_chan_dot_t(const void* a, const void* b, float* out, int64_t rows, int C, int64_t HW, const float* bias, const float* noise,
                   const float* noise_gain, int act, float alpha, float gain, int dtype, spi_stream_t stream) {
    SPI_DTYPE_CHECK("spi_chan_dot_t");
    if (dtype == SPI_DTYPE_F16) return chan_dot_launch<_Float16>((const _Float16*)a, (const _Float16*)b, out, rows, C, HW, bias, noise, noise_gain, act, alpha, gain, stream);
    return chan_dot_launch<float>((const float*)a, (const float*)b, out, rows, C, HW, bias, noise, noise_gain, act, alpha, gain, stream);
}

int spi_style_grad(const float* a, const float* cv, const float* st, const float* dcoef, const float* ww, float* ds, int N, int NS,
                   int I, int O, float style_gain, spi_stream_t stream) {
    SPI_REQUIRE(a && st && ds && N > 0 && (NS == N || NS == 1) && I > 0 && O > 0 && O <= 8192, "spi_style_grad: bad argument");
    SPI_REQUIRE(cv == nullptr || (dcoef && ww), "spi_style_grad: the demodulation term needs dcoef and ww");
    hipLaunchKernelGGL(style_grad_kernel, dim3((unsigned)((I + 31) / 32), (unsigned)NS), dim3(256), (size_t)(O + 256) * sizeof(float), as_stream(stream),
                       a, cv, st, dcoef, ww, ds, N, NS, I, O, style_gain * style_gain);
    SPI_LAUNCH_CHECK("spi_style_grad");
    return SPI_OK;
}

int spi_seg_flags(const float* x, int32_t* flags, int N, int C, int64_t HW, spi_stream_t stream) {
    SPI_REQUIRE(x && flags && N > 0 && N < 65536 && C > 0 && HW > 0 && HW < (1ll << 31), "spi_seg_flags: bad argument");
    const int nseg = (int)ceil_div64(HW, SPI_SEG_PIXELS);
    hipLaunchKernelGGL(seg_flags_kernel<float>, dim3((unsigned)ceil_div64(HW, 1024), (unsigned)N), dim3(256), 0, as_stream(stream), x, flags, C, HW, nseg);
    SPI_LAUNCH_CHECK("spi_seg_flags");
    return SPI_OK;
}

int spi_seg_flags_t(const void* x, int32_t* flags, int N, int C, int64_t HW, int dtype, spi_stream_t stream) {
    SPI_DTYPE_CHECK("spi_seg_flags_t");
    if (dtype == SPI_DTYPE_F32) return spi_seg_flags((const float*)x, flags, N, C, HW, stream);
    SPI_REQUIRE(x && flags && N > 0 && N < 65536 && C > 0 && HW > 0 && HW < (1ll << 31), "spi_seg_flags_t: bad argument");
    const int nseg = (int)ceil_div64(HW, SPI_SEG_PIXELS);
    hipLaunchKernelGGL(seg_flags_kernel<_Float16>, dim3((unsigned)ceil_div64(HW, 1024), (unsigned)N), dim3(256), 0, as_stream(stream), (const _Float16*)x, flags, C, HW, nseg);
    SPI_LAUNCH_CHECK("spi_seg_flags_t");
    return SPI_OK;
}

int spi_upfirdn2d(const float* x, const float* f, float* y, int N, int C, int inH, int inW, int fH, int fW, int upx, int upy,
                  int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain, int outH, int outW,
                  const float* noise, const float* noise_gain, const float* bias, int act, float alpha, float act_gain,
                  float clamp, spi_stream_t stream) {
    SPI_REQUIRE(x && f && y, "spi_upfirdn2d: null tensor");
    SPI_REQUIRE(N > 0 && C > 0 && inH > 0 && inW > 0 && fH > 0 && fW > 0 && fH * fW <= MAX_TAPS, "spi_upfirdn2d: bad sizes");
    SPI_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, "spi_upfirdn2d: bad up/down factors");
    const int eh = (inH * upy + pady0 + pady1 - fH + downy) / downy, ew = (inW * upx + padx0 + padx1 - fW + downx) / downx;
    SPI_REQUIRE(outH == eh && outW == ew && outH > 0 && outW > 0, "spi_upfirdn2d: output size must be %dx%d, got %dx%d", eh, ew, outH, outW);
    SPI_REQUIRE(act >= 0 && act <= SPI_ACT_SWISH, "spi_upfirdn2d: unknown activation");
    UpfirdnParams p{N, C, inH, inW, fH, fW, upx, upy, downx, downy, padx0, pady0, flip, outH, outW, gain};
    ActParams ap{act, 0, alpha, act_gain, clamp};
    return launch_upfirdn(x, f, y, p, nullptr, noise, noise_gain, bias, ap, stream);
}

// spi_upfirdn2d with its fused layer tail on fp16 (or fp32) NCHW tensors: the 4x4 / up = down = 1 FIR of the up-sampling layers and its adjoint on
// images of >= 100 pixels (the LDS-tiled kernel); every other shape in fp16 goes through spi_upfirdn2d_t (+ spi_bias_act_t), as in the reference.
int spi_upfirdn2d_fused_t(const void* x, const float* f, void* y, int N, int C, int inH, int inW, int fH, int fW, int upx, int upy,
                          int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain, int outH, int outW,
                          const float* noise, const float* noise_gain, const float* bias, int act, float alpha, float act_gain,
                          float clamp, int dtype, spi_stream_t stream) {
    SPI_DTYPE_CHECK("spi_upfirdn2d_fused_t");
    if (dtype == SPI_DTYPE_F32)
        return spi_upfirdn2d((const float*)x, f, (float*)y, N, C, inH, inW, fH, fW, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain, outH, outW,
                             noise, noise_gain, bias, act, alpha, act_gain, clamp, stream);
    SPI_REQUIRE(x && f && y && N > 0 && C > 0 && inH > 0 && inW > 0, "spi_upfirdn2d_fused_t: bad argument");
    const int eh = (inH * upy + pady0 + pady1 - fH + downy) / downy, ew = (inW * upx + padx0 + padx1 - fW + downx) / downx;
    SPI_REQUIRE(outH == eh && outW == ew && outH > 0 && outW > 0, "spi_upfirdn2d_fused_t: output size must be %dx%d, got %dx%d", eh, ew, outH, outW);
    const bool tiled = fH == 4 && fW == 4 && upx == 1 && upy == 1 && downx == 1 && downy == 1 && outW >= 100 && outH >= 100 &&
                       (act == 0 || act == SPI_ACT_LINEAR || act == SPI_ACT_LRELU) && (int64_t)N * C <= 65535;
    if (!tiled) { spi_set_error("spi_upfirdn2d_fused_t: fp16 tensors are served for the 4x4 filter with up = down = 1 on >= 100-pixel images (use spi_upfirdn2d_t)"); return SPI_ERR_UNSUPPORTED; }
    UpfirdnParams p{N, C, inH, inW, fH, fW, upx, upy, downx, downy, padx0, pady0, flip, outH, outW, gain};
    ActParams ap{act, 0, alpha, act_gain, clamp};
    const dim3 g((unsigned)((outW + FT_W - 1) / FT_W), (unsigned)((outH + FT_H - 1) / FT_H), (unsigned)(N * C));
    hipLaunchKernelGGL(upfirdn2d_4x4_tiled_kernel<_Float16>, g, dim3(256), 0, as_stream(stream), (const _Float16*)x, f, (_Float16*)y, p, (const float*)nullptr, noise, noise_gain, bias, ap);
    SPI_LAUNCH_CHECK("spi_upfirdn2d_fused_t");
    return SPI_OK;
}

int spi_bias_act_t(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y, int64_t n, int sizeB,
                   int64_t stepB, int grad, int act, float alpha, float gain, float clamp, int dtype, spi_stream_t stream) {
    if (dtype == SPI_DTYPE_F32)
        return spi_bias_act((const float*)x, (const float*)b, (const float*)xref, (const float*)yref, (const float*)dy, (float*)y, n, sizeB, stepB,
                            grad, act, alpha, gain, clamp, stream);
    if (dtype != SPI_DTYPE_F16 && dtype != SPI_DTYPE_F64) { spi_set_error("spi_bias_act_t: dtype %d (0 = fp32, 1 = fp16, 2 = fp64)", dtype); return SPI_ERR_UNSUPPORTED; }
    SPI_REQUIRE(x && y && n > 0, "spi_bias_act_t: null tensor or empty");
    SPI_REQUIRE(act >= SPI_ACT_LINEAR && act <= SPI_ACT_SWISH, "spi_bias_act_t: unknown activation %d", act);
    SPI_REQUIRE(grad >= 0 && grad <= 2, "spi_bias_act_t: grad must be 0, 1 or 2");
    SPI_REQUIRE(b == nullptr || (sizeB > 0 && stepB > 0), "spi_bias_act_t: bias given without sizeB/stepB");
    ActParams p{act, grad, alpha, gain, clamp};
    if (dtype == SPI_DTYPE_F64) {
        const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(n, 256), 256 * 16);
        hipLaunchKernelGGL(bias_act_f64_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (const double*)x, (const double*)b, (const double*)xref,
                           (const double*)yref, (const double*)dy, (double*)y, n, sizeB, stepB, p);
        SPI_LAUNCH_CHECK("spi_bias_act_t");
        return SPI_OK;
    }
    return launch_bias_act_t<_Float16>(x, b, xref, yref, dy, y, n, sizeB, stepB, p, stream);
}

int spi_upfirdn2d_t(const void* x, const float* f, void* y, int N, int C, int inH, int inW, const int64_t* x_strides, const int64_t* y_strides,
                    int fH, int fW, int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                    int outH, int outW, int dtype, spi_stream_t stream) {
    SPI_REQUIRE(x && f && y, "spi_upfirdn2d_t: null tensor");
    SPI_REQUIRE(N > 0 && C > 0 && inH > 0 && inW > 0 && fH > 0 && fW > 0 && fH * fW <= MAX_TAPS, "spi_upfirdn2d_t: bad sizes");
    SPI_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, "spi_upfirdn2d_t: bad up/down factors");
    const int eh = (inH * upy + pady0 + pady1 - fH + downy) / downy, ew = (inW * upx + padx0 + padx1 - fW + downx) / downx;
    SPI_REQUIRE(outH == eh && outW == ew && outH > 0 && outW > 0, "spi_upfirdn2d_t: output size must be %dx%d, got %dx%d", eh, ew, outH, outW);
    if (dtype != SPI_DTYPE_F32 && dtype != SPI_DTYPE_F16 && dtype != SPI_DTYPE_F64) { spi_set_error("spi_upfirdn2d_t: dtype %d (0 = fp32, 1 = fp16, 2 = fp64)", dtype); return SPI_ERR_UNSUPPORTED; }
    const Strides4 xs = x_strides ? Strides4{x_strides[0], x_strides[1], x_strides[2], x_strides[3]} : Strides4{(int64_t)C * inH * inW, (int64_t)inH * inW, inW, 1};
    const Strides4 ys = y_strides ? Strides4{y_strides[0], y_strides[1], y_strides[2], y_strides[3]} : Strides4{(int64_t)C * outH * outW, (int64_t)outH * outW, outW, 1};
    const bool dense_nchw = xs.w == 1 && xs.h == inW && xs.c == (int64_t)inH * inW && xs.n == xs.c * C && ys.w == 1 && ys.h == outW && ys.c == (int64_t)outH * outW && ys.n == ys.c * C;
    if (dtype == SPI_DTYPE_F32 && dense_nchw)            // the tuned fp32 kernels of the loop
        return spi_upfirdn2d((const float*)x, f, (float*)y, N, C, inH, inW, fH, fW, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain, outH,
                             outW, nullptr, nullptr, nullptr, 0, 0.f, 1.f, -1.f, stream);
    // both tensors must be dense in SOME permutation with non-overlapping elements: the two layouts the reference produces
    const bool cl_x = xs.c == 1 && xs.w == C && xs.h == (int64_t)inW * C && xs.n == (int64_t)inH * inW * C;
    const bool cl_y = ys.c == 1 && ys.w == C && ys.h == (int64_t)outW * C && ys.n == (int64_t)outH * outW * C;
    const bool nchw_x = xs.w == 1 && xs.h == inW && xs.c == (int64_t)inH * inW && xs.n == xs.c * C;
    const bool nchw_y = ys.w == 1 && ys.h == outW && ys.c == (int64_t)outH * outW && ys.n == ys.c * C;
    SPI_REQUIRE((cl_x || nchw_x) && (cl_y || nchw_y), "spi_upfirdn2d_t: tensors must be dense NCHW or channels_last (upfirdn2d.cpp:23 'non-overlapping and dense')");
    UpfirdnParams p{N, C, inH, inW, fH, fW, upx, upy, downx, downy, padx0, pady0, flip, outH, outW, gain};
    const int64_t total = (int64_t)N * C * outH * outW;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(total, 256), 256 * 32);
    if (dtype == SPI_DTYPE_F16) hipLaunchKernelGGL(upfirdn2d_t_kernel<_Float16>, dim3(grid), dim3(256), 0, as_stream(stream), (const _Float16*)x, f, (_Float16*)y, p, xs, ys, cl_y ? 1 : 0);
    else if (dtype == SPI_DTYPE_F64) hipLaunchKernelGGL(upfirdn2d_t_kernel<double>, dim3(grid), dim3(256), 0, as_stream(stream), (const double*)x, f, (double*)y, p, xs, ys, cl_y ? 1 : 0);
    else hipLaunchKernelGGL(upfirdn2d_t_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)x, f, (float*)y, p, xs, ys, cl_y ? 1 : 0);
    SPI_LAUNCH_CHECK("spi_upfirdn2d_t");
    return SPI_OK;
}

int spi_filtered_lrelu(const float* x, const float* fu, const float* fd, const float* b, float* tmp, float* y, int N, int C,
                       int inH, int inW, int fuH, int fuW, int fdH, int fdW, int up, int down, int px0, int px1, int py0, int py1,
                       float gain, float slope, float clamp, int flip, int outH, int outW, spi_stream_t stream) {
    SPI_REQUIRE(x && fu && fd && tmp && y, "spi_filtered_lrelu: null tensor");
    SPI_REQUIRE(up >= 1 && down >= 1 && fuH * fuW <= MAX_TAPS && fdH * fdW <= MAX_TAPS, "spi_filtered_lrelu: bad factors / filter too large");
    const int midH = inH * up + py0 + py1 - fuH + 1, midW = inW * up + px0 + px1 - fuW + 1;
    const int eh = (midH - fdH + down) / down, ew = (midW - fdW + down) / down;
    SPI_REQUIRE(midH > 0 && midW > 0 && outH == eh && outW == ew, "spi_filtered_lrelu: output size must be %dx%d", eh, ew);
    UpfirdnParams p1{N, C, inH, inW, fuH, fuW, up, up, 1, 1, px0, py0, flip, midH, midW, (float)(up * up)};
    ActParams a1{SPI_ACT_LRELU, 0, slope, gain, clamp};
    int rc = launch_upfirdn(x, fu, tmp, p1, b, nullptr, nullptr, nullptr, a1, stream);
    if (rc) return rc;
    UpfirdnParams p2{N, C, midH, midW, fdH, fdW, 1, 1, down, down, 0, 0, flip, outH, outW, 1.f};
    ActParams a2{0, 0, 0.f, 1.f, -1.f};
    return launch_upfirdn(tmp, fd, y, p2, nullptr, nullptr, nullptr, nullptr, a2, stream);
}

// filtered_lrelu on half tensors (filtered_lrelu.cpp:151,265 dispatch the plugin for half as well): x, b and y are fp16, both filters and
// the intermediate (up-sampled, activated) tensor `tmp` are fp32 -- the plugin's internal type for half -- so the result is rounded once.
int spi_filtered_lrelu_t(const void* x, const float* fu, const float* fd, const void* b, float* tmp, void* y, int N, int C,
                         int inH, int inW, int fuH, int fuW, int fdH, int fdW, int up, int down, int px0, int px1, int py0, int py1,
                         float gain, float slope, float clamp, int flip, int outH, int outW, int dtype, spi_stream_t stream) {
    SPI_DTYPE_CHECK("spi_filtered_lrelu_t");
    if (dtype == SPI_DTYPE_F32)
        return spi_filtered_lrelu((const float*)x, fu, fd, (const float*)b, tmp, (float*)y, N, C, inH, inW, fuH, fuW, fdH, fdW, up, down, px0, px1, py0, py1,
                                  gain, slope, clamp, flip, outH, outW, stream);
    SPI_REQUIRE(x && fu && fd && tmp && y, "spi_filtered_lrelu_t: null tensor");
    SPI_REQUIRE(up >= 1 && down >= 1 && fuH * fuW <= MAX_TAPS && fdH * fdW <= MAX_TAPS, "spi_filtered_lrelu_t: bad factors / filter too large");
    const int midH = inH * up + py0 + py1 - fuH + 1, midW = inW * up + px0 + px1 - fuW + 1;
    const int eh = (midH - fdH + down) / down, ew = (midW - fdW + down) / down;
    SPI_REQUIRE(midH > 0 && midW > 0 && outH == eh && outW == ew, "spi_filtered_lrelu_t: output size must be %dx%d", eh, ew);
    // the bias arrives in the tensors' dtype (filtered_lrelu.py:117 b.to(x.dtype)): widen its C entries into the head of a small fp32 scratch
    // that the first pass reads as pre-bias -- kept in `tmp`'s tail is not possible (tmp is exactly mid-sized), so the caller passes b as fp16
    // and the kernel converts on load through a typed pre-bias pointer: done with a tiny conversion launch into a static-size stack of C floats
    // is not available on the device either -> the first pass takes the bias as fp16 directly.
    UpfirdnParams p1{N, C, inH, inW, fuH, fuW, up, up, 1, 1, px0, py0, flip, midH, midW, (float)(up * up)};
    ActParams a1{SPI_ACT_LRELU, 0, slope, gain, clamp};
    const int64_t tot1 = (int64_t)N * C * midH * midW, tot2 = (int64_t)N * C * outH * outW;
    hipLaunchKernelGGL((upfirdn2d_hb_kernel), dim3((unsigned)std::min<int64_t>(ceil_div64(tot1, 256), 256 * 32)), dim3(256), 0, as_stream(stream),
                       (const _Float16*)x, fu, tmp, p1, (const _Float16*)b, a1);
    UpfirdnParams p2{N, C, midH, midW, fdH, fdW, 1, 1, down, down, 0, 0, flip, outH, outW, 1.f};
    ActParams a2{0, 0, 0.f, 1.f, -1.f};
    hipLaunchKernelGGL((upfirdn2d_kernel<float, _Float16>), dim3((unsigned)std::min<int64_t>(ceil_div64(tot2, 256), 256 * 32)), dim3(256), 0, as_stream(stream),
                       (const float*)tmp, fd, (_Float16*)y, p2, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, a2);
    SPI_LAUNCH_CHECK("spi_filtered_lrelu_t");
    return SPI_OK;
}

int spi_filtered_lrelu_act(float* x, uint8_t* signs, int64_t NC, int xH, int xW, int sH, int sW, int sx, int sy, float gain, float slope,
                           float clamp, int mode, spi_stream_t stream) {
    SPI_REQUIRE(x && NC > 0 && xH > 0 && xW > 0, "spi_filtered_lrelu_act: bad argument");
    SPI_REQUIRE(mode >= 0 && mode <= 2, "spi_filtered_lrelu_act: mode must be 0 (plain), 1 (write signs) or 2 (read signs)");
    SPI_REQUIRE(mode == 0 || (signs && sH > 0 && sW > 0 && (sW & 3) == 0), "spi_filtered_lrelu_act: sign tensor [NC, sH, sW/4] needs sW %% 4 == 0");
    const int w = mode == 1 ? std::max(sW, xW) : xW, h = mode == 1 ? std::max(sH, xH) : xH;
    dim3 grid((unsigned)((w + 1023) / 1024), (unsigned)std::min(h, 65535), (unsigned)std::min<int64_t>(NC, 65535));
    hipLaunchKernelGGL(filtered_lrelu_act_kernel, grid, dim3(256), 0, as_stream(stream), x, signs, NC, xH, xW, sH, sW, sx, sy, gain, slope,
                       clamp < 0.f ? INFINITY : clamp, mode);
    SPI_LAUNCH_CHECK("spi_filtered_lrelu_act");
    return SPI_OK;
}

int spi_rotate_warp(const float* tgt_cam, const float* src_cam_inv, const float* src_cam, const float* tgt_depth,
                    const float* src_depth, const float* src_image, const float* src_mask, int N, int res, int dres, float eps,
                    float* warp_rgb, float* warp_mask, spi_stream_t stream) {
    SPI_REQUIRE(tgt_cam && src_cam_inv && src_cam && tgt_depth && src_depth && src_image && warp_rgb && warp_mask, "spi_rotate_warp: null tensor");
    SPI_REQUIRE(N > 0 && res > 0 && dres > 0, "spi_rotate_warp: bad sizes");
    const int64_t total = (int64_t)N * res * res;
    hipLaunchKernelGGL(rotate_warp_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, as_stream(stream), tgt_cam,
                       src_cam_inv, src_cam, tgt_depth, src_depth, src_image, src_mask, N, res, dres, eps, warp_rgb, warp_mask);
    SPI_LAUNCH_CHECK("spi_rotate_warp");
    return SPI_OK;
}

// ------------------------------------------------------------------------------------------------
// modulate / demodulate (networks_stylegan2.py:62-69).  One block per output channel o; the weight row
// W[o] (I*T floats) is staged tap-major in LDS once and reused for every sample n.
//   forward : v = W s,  d = rsqrt(|v|^2 + eps),  w'' = v d        (written [N,O,T,I])
//   backward: dv = d g - d^3 v (g.v);  dW = sum_n dv s;  ds[n,i] += sum_{o,t} dv W
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// one block = one output channel `o` of one layer; `smem`: I * T floats of dynamic LDS
__device__ __forceinline__ void modulate_fwd_body(const float* __restrict__ weight, const float* __restrict__ styles,
                                                  float* __restrict__ w_out, float* __restrict__ dcoef, int N, int O,
                                                  int I, int T, int demod, float sgain, int o, float* smem, float* red) {
    float* Ws = smem;                  // [I][T] copy of W[o] in storage order (one coalesced sweep; read back with lane stride T: odd, conflict-free)
    const int tid = threadIdx.x, IT = I * T;
    const float* wr = weight + (int64_t)o * IT;
    // loops are (tap, channel) nests: no per-element division by T or modulo I (each costs ~30 VALU instructions on gfx950)
    for (int e = tid; e < IT; e += 256) Ws[e] = wr[e];
    __syncthreads();
    for (int n = 0; n < N; ++n) {
        const float* sn = styles + (int64_t)n * I;
        float ss = 0.f;
        if (demod) {
            for (int i = tid; i < I; i += 256) {
                const float sv = sn[i] * sgain;
                for (int t = 0; t < T; ++t) { const float v = Ws[i * T + t] * sv; ss = fmaf(v, v, ss); }
            }
            ss = block_sum_256(ss, red);
        }
        const float d = demod ? rsqrtf(ss + 1e-8f) : 1.f;
        if (tid == 0 && dcoef) dcoef[(int64_t)n * O + o] = d;
        float* dst = w_out + ((int64_t)n * O + o) * IT;
        for (int i = tid; i < I; i += 256) {
            const float sv = sn[i] * sgain * d;
            for (int t = 0; t < T; ++t) dst[t * I + i] = Ws[i * T + t] * sv;
        }
    }
}

__global__ void __launch_bounds__(256) modulate_fwd_kernel(const float* __restrict__ weight, const float* __restrict__ styles,
                                                           float* __restrict__ w_out, float* __restrict__ dcoef, int N, int O,
                                                           int I, int T, int demod, float sgain) {
    extern __shared__ float smem[];
    __shared__ float red[4];
    modulate_fwd_body(weight, styles, w_out, dcoef, N, O, I, T, demod, sgain, (int)blockIdx.x, smem, red);
}

// one block = one output channel `o` of one layer; `smem`: 3 * I * T floats of dynamic LDS
__device__ __forceinline__ void modulate_bwd_body(const float* __restrict__ weight, const float* __restrict__ styles,
                                                  const float* __restrict__ dcoef, const float* __restrict__ g,
                                                  float* __restrict__ d_weight, float* __restrict__ d_styles, int N,
                                                  int O, int I, int T, int demod, float sgain, int o, float* smem, float* red) {
    const int IT = I * T;
    float* Ws = smem;                  // [I][T] the weight row, in storage order
    float* Acc = smem + IT;            // [I][T] running dW, in storage order
    float* Gs = smem + 2 * IT;         // [T][I] this sample's gradient of the modulated row (tap-major, as the conv kernels read it)
    const int tid = threadIdx.x;
    const float* wr = weight + (int64_t)o * IT;
    // Every global access is a linear, coalesced sweep over the row (round 3; the [I][T] <-> [T][I] transposition happens in the LDS
    // indices: lanes walk channels, stride T = 9 or 1 floats -- odd, so conflict-free).  Until then thread i read and wrote its 9 taps
    // with stride-T lane addresses (18 cache lines per wave-instruction) and fetched g twice: 15.6 us per 512 x 512 x 9 layer.
    // Thread `tid` owns channels tid, tid + 256, ... in both passes: no race on Acc, and the d_styles sum of a channel is finished by
    // the thread that produced its dv.
    for (int e = tid; e < IT; e += 256) { Ws[e] = wr[e]; Acc[e] = 0.f; }
    for (int n = 0; n < N; ++n) {
        const float* sn = styles + (int64_t)n * I;
        const float* gn = g + ((int64_t)n * O + o) * IT;
        if (n) __syncthreads();                                   // the previous sample's Gs is still being read
        for (int e = tid; e < IT; e += 256) Gs[e] = gn[e];
        __syncthreads();
        float gv = 0.f;
        if (demod) {
            for (int i = tid; i < I; i += 256) {
                const float sv = sn[i] * sgain;
                for (int t = 0; t < T; ++t) gv = fmaf(Gs[t * I + i], Ws[i * T + t] * sv, gv);
            }
            gv = block_sum_256(gv, red);
        }
        const float d = demod ? dcoef[(int64_t)n * O + o] : 1.f;
        const float k3 = demod ? d * d * d * gv : 0.f;
        for (int i = tid; i < I; i += 256) {
            const float sv = sn[i] * sgain;
            float a = 0.f;
            for (int t = 0; t < T; ++t) {
                const float wv = Ws[i * T + t];
                const float dv = d * Gs[t * I + i] - k3 * (wv * sv);
                Acc[i * T + t] = fmaf(dv, sv, Acc[i * T + t]);
                a = fmaf(dv, wv, a);
            }
            atomicAdd(d_styles + (int64_t)n * I + i, a * sgain);
        }
    }
    if (d_weight) {
        __syncthreads();
        float* dst = d_weight + (int64_t)o * IT;
        for (int e = tid; e < IT; e += 256) dst[e] = Acc[e];
    }
}

__global__ void __launch_bounds__(256) modulate_bwd_kernel(const float* __restrict__ weight, const float* __restrict__ styles,
                                                           const float* __restrict__ dcoef, const float* __restrict__ g,
                                                           float* __restrict__ d_weight, float* __restrict__ d_styles, int N,
                                                           int O, int I, int T, int demod, float sgain) {
    extern __shared__ float smem[];
    __shared__ float red[4];
    modulate_bwd_body(weight, styles, dcoef, g, d_weight, d_styles, N, O, I, T, demod, sgain, (int)blockIdx.x, smem, red);
}

// ---- the modulation of ALL layers of a network in one launch each way: the same bodies over a table of jobs (by value)
struct ModulateJobs { int n; int blk0[SPI_MODULATE_MAX_JOBS + 1]; spi_modulate_job job[SPI_MODULATE_MAX_JOBS]; };

__device__ __forceinline__ int modulate_find_job(const ModulateJobs& J, int blk) {
    int j = 0;
    while (j + 1 < J.n && blk >= J.blk0[j + 1]) ++j;
    return j;
}

__global__ void __launch_bounds__(256) modulate_multi_fwd_kernel(ModulateJobs J, int N) {
    extern __shared__ float smem[];
    __shared__ float red[4];
    const int j = modulate_find_job(J, blockIdx.x);
    const spi_modulate_job& q = J.job[j];
    modulate_fwd_body(q.weight, q.styles, q.w_out, q.dcoef, N, q.O, q.I, q.T, q.demodulate, q.style_gain, (int)blockIdx.x - J.blk0[j], smem, red);
}

__global__ void __launch_bounds__(256) modulate_multi_bwd_kernel(ModulateJobs J, int N) {
    extern __shared__ float smem[];
    __shared__ float red[4];
    const int j = modulate_find_job(J, blockIdx.x);
    const spi_modulate_job& q = J.job[j];
    modulate_bwd_body(q.weight, q.styles, q.dcoef, q.g, q.d_weight, q.d_styles, N, q.O, q.I, q.T, q.demodulate, q.style_gain, (int)blockIdx.x - J.blk0[j], smem, red);
}

int spi_modulate_fwd(const float* weight, const float* styles, float* w_out, float* dcoef, int N, int O, int I, int T,
                     int demodulate, float style_gain, spi_stream_t stream) {
    SPI_REQUIRE(weight && styles && w_out, "spi_modulate_fwd: null tensor");
    SPI_REQUIRE(N > 0 && O > 0 && I > 0 && T > 0 && (int64_t)I * T * 12 <= 64 * 1024, "spi_modulate_fwd: bad sizes (I*T must be <= 5461)");
    SPI_REQUIRE(!demodulate || dcoef, "spi_modulate_fwd: demodulation needs the dcoef output");
    hipLaunchKernelGGL(modulate_fwd_kernel, dim3((unsigned)O), dim3(256), (size_t)I * T * 4, as_stream(stream), weight, styles, w_out,
                       dcoef, N, O, I, T, demodulate, style_gain);
    SPI_LAUNCH_CHECK("spi_modulate_fwd");
    return SPI_OK;
}

int spi_modulate_bwd(const float* weight, const float* styles, const float* dcoef, const float* g, float* d_weight,
                     float* d_styles, int N, int O, int I, int T, int demodulate, float style_gain, spi_stream_t stream) {
    SPI_REQUIRE(weight && styles && g && d_styles, "spi_modulate_bwd: null tensor");
    SPI_REQUIRE(N > 0 && O > 0 && I > 0 && T > 0 && (int64_t)I * T * 12 <= 64 * 1024, "spi_modulate_bwd: bad sizes (I*T must be <= 5461)");
    SPI_REQUIRE(!demodulate || dcoef, "spi_modulate_bwd: demodulation needs dcoef from the forward pass");
    hipLaunchKernelGGL(modulate_bwd_kernel, dim3((unsigned)O), dim3(256), (size_t)I * T * 12, as_stream(stream), weight, styles, dcoef,
                       g, d_weight, d_styles, N, O, I, T, demodulate, style_gain);
    SPI_LAUNCH_CHECK("spi_modulate_bwd");
    return SPI_OK;
}

static int modulate_multi(const spi_modulate_job* jobs, int n_jobs, int N, bool bwd, spi_stream_t stream) {
    const char* who = bwd ? "spi_modulate_multi_bwd" : "spi_modulate_multi_fwd";
    SPI_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= SPI_MODULATE_MAX_JOBS && N > 0, "%s: need 1 <= n_jobs <= %d and N > 0", who, SPI_MODULATE_MAX_JOBS);
    ModulateJobs J;
    J.n = n_jobs;
    int blocks = 0;
    int64_t max_it = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const spi_modulate_job& q = jobs[j];
        SPI_REQUIRE(q.weight && q.styles && q.O > 0 && q.I > 0 && q.T > 0 && (int64_t)q.I * q.T * 12 <= 64 * 1024, "%s: job %d: null tensor or bad sizes (I*T must be <= 5461)", who, j);
        SPI_REQUIRE(!q.demodulate || q.dcoef, "%s: job %d: demodulation needs dcoef", who, j);
        SPI_REQUIRE(bwd ? (q.g && q.d_styles) : (q.w_out != nullptr), "%s: job %d: missing %s", who, j, bwd ? "g / d_styles" : "w_out");
        J.job[j] = q; J.blk0[j] = blocks; blocks += q.O;
        max_it = std::max<int64_t>(max_it, (int64_t)q.I * q.T);
    }
    J.blk0[n_jobs] = blocks;
    if (bwd) hipLaunchKernelGGL(modulate_multi_bwd_kernel, dim3((unsigned)blocks), dim3(256), (size_t)max_it * 12, as_stream(stream), J, N);
    else hipLaunchKernelGGL(modulate_multi_fwd_kernel, dim3((unsigned)blocks), dim3(256), (size_t)max_it * 4, as_stream(stream), J, N);
    SPI_LAUNCH_CHECK(who);
    return SPI_OK;
}

int spi_modulate_multi_fwd(const spi_modulate_job* jobs, int n_jobs, int N, spi_stream_t stream) { return modulate_multi(jobs, n_jobs, N, false, stream); }
int spi_modulate_multi_bwd(const spi_modulate_job* jobs, int n_jobs, int N, spi_stream_t stream) { return modulate_multi(jobs, n_jobs, N, true, stream); }

int spi_noise_reg_fwd(const float* const* bufs, const int32_t* res, int T, int max_res, float* pyramid, float* means,
                      float* loss, spi_stream_t stream) {
    // `means` holds 8 levels x 2 sums per buffer (res, res/2, ... down to 8): 8 * 2^7 = 1024 is the largest resolution that fits
    SPI_REQUIRE(bufs && res && pyramid && means && loss && T > 0 && max_res >= 1 && max_res <= 1024, "spi_noise_reg_fwd: bad argument (max_res <= 1024)");
    hipLaunchKernelGGL(noise_reg_fwd_kernel, dim3((unsigned)T), dim3(1024), 0, as_stream(stream), bufs, res, (int64_t)max_res * max_res / 2,
                       pyramid, means, loss);
    SPI_LAUNCH_CHECK("spi_noise_reg_fwd");
    return SPI_OK;
}

int spi_noise_reg_bwd(const float* const* bufs, const int32_t* res, int T, int max_res, const float* pyramid, const float* means,
                      const float* gout, float* grads, const int64_t* goff, float* gpyramid, spi_stream_t stream) {
    SPI_REQUIRE(bufs && res && pyramid && means && gout && grads && goff && gpyramid && T > 0 && max_res >= 1 && max_res <= 1024,
                "spi_noise_reg_bwd: bad argument");
    hipLaunchKernelGGL(noise_reg_bwd_kernel, dim3((unsigned)T), dim3(1024), 0, as_stream(stream), bufs, res, (int64_t)max_res * max_res / 2,
                       pyramid, means, gout, grads, goff, gpyramid);
    SPI_LAUNCH_CHECK("spi_noise_reg_bwd");
    return SPI_OK;
}

int spi_noise_renorm(float* const* bufs, const int32_t* res, int T, spi_stream_t stream) {
    SPI_REQUIRE(bufs && res && T > 0, "spi_noise_renorm: bad argument");
    hipLaunchKernelGGL(noise_renorm_kernel, dim3((unsigned)T), dim3(1024), 0, as_stream(stream), bufs, res);
    SPI_LAUNCH_CHECK("spi_noise_renorm");
    return SPI_OK;
}

int spi_lpips_layer_fwd(const float* fx, const float* fy, const float* lin, int N, int C, int64_t HW, float* out, spi_stream_t stream) {
    SPI_REQUIRE(fx && fy && lin && out && N > 0 && C > 0 && HW > 0, "spi_lpips_layer_fwd: bad argument");
    hipLaunchKernelGGL(lpips_fwd_kernel, dim3((unsigned)ceil_div64(HW, 64), (unsigned)N), dim3(1024), 0, as_stream(stream), fx, fy, lin, C, HW, out);
    SPI_LAUNCH_CHECK("spi_lpips_layer_fwd");
    return SPI_OK;
}

int spi_lpips_layer_bwd(const float* fx, const float* fy, const float* lin, const float* d_out, int N, int C, int64_t HW,
                        float* d_fx, spi_stream_t stream) {
    SPI_REQUIRE(fx && fy && lin && d_out && d_fx && N > 0 && C > 0 && HW > 0, "spi_lpips_layer_bwd: bad argument");
    hipLaunchKernelGGL(lpips_bwd_kernel, dim3((unsigned)ceil_div64(HW, 64), (unsigned)N), dim3(1024), 0, as_stream(stream), fx, fy, lin, d_out, C, HW, d_fx);
    SPI_LAUNCH_CHECK("spi_lpips_layer_bwd");
    return SPI_OK;
}

int spi_adam_multi(void* const* ptrs, const int64_t* sizes, int T, int64_t max_size, float lr, float beta1, float beta2,
                   float eps, int step, spi_stream_t stream) {
    SPI_REQUIRE(ptrs && sizes && T > 0 && max_size > 0 && step >= 1, "spi_adam_multi: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    const unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(max_size, 256 * 4), 2048);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(gx, (unsigned)T), dim3(256), 0, as_stream(stream), ptrs, sizes, lr, beta1, beta2, eps, bc1, bc2_sqrt,
                       (const float*)nullptr, (const unsigned char*)nullptr);
    SPI_LAUNCH_CHECK("spi_adam_multi");
    return SPI_OK;
}

int spi_adam_multi_pred(void* const* ptrs, const int64_t* sizes, int T, int64_t max_size, float lr, float beta1, float beta2,
                        float eps, int step, const unsigned char* skip, spi_stream_t stream) {
    SPI_REQUIRE(ptrs && sizes && skip && T > 0 && max_size > 0 && step >= 1, "spi_adam_multi_pred: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    const unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(max_size, 256 * 4), 2048);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(gx, (unsigned)T), dim3(256), 0, as_stream(stream), ptrs, sizes, lr, beta1, beta2, eps, bc1, bc2_sqrt,
                       (const float*)nullptr, skip);
    SPI_LAUNCH_CHECK("spi_adam_multi_pred");
    return SPI_OK;
}

int spi_adam_multi_dev(void* const* ptrs, const int64_t* sizes, int T, int64_t max_size, const float* hyper, float beta1, float beta2,
                       float eps, spi_stream_t stream) {
    SPI_REQUIRE(ptrs && sizes && hyper && T > 0 && max_size > 0, "spi_adam_multi_dev: bad argument");
    const unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(max_size, 256 * 4), 2048);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(gx, (unsigned)T), dim3(256), 0, as_stream(stream), ptrs, sizes, 0.f, beta1, beta2, eps, 1.f, 1.f, hyper, (const unsigned char*)nullptr);
    SPI_LAUNCH_CHECK("spi_adam_multi_dev");
    return SPI_OK;
}

int spi_affine_fwd(const float* x, const float* w, const float* b, float gain, float* y, int N, int I, int O, spi_stream_t stream) {
    SPI_REQUIRE(x && w && y && N >= 1 && N <= AFF_NMAX && I >= 4 && I % 4 == 0 && O >= 1, "spi_affine_fwd: need 1 <= N <= %d rows, in_features a multiple of 4 (got N = %d, I = %d, O = %d)", AFF_NMAX, N, I, O);
    hipLaunchKernelGGL(affine_fwd_kernel, dim3((unsigned)((O + 3) / 4)), dim3(256), 0, as_stream(stream), x, w, b, gain, y, N, I, O);
    SPI_LAUNCH_CHECK("spi_affine_fwd");
    return SPI_OK;
}

int spi_affine_bwd(const float* g, const float* x, const float* w, float gain, float* dx, float* dw, int N, int I, int O, spi_stream_t stream) {
    SPI_REQUIRE(g && (dx || dw) && (!dx || w) && (!dw || x) && N >= 1 && N <= AFF_NMAX && I >= 4 && I % 4 == 0 && O >= 1, "spi_affine_bwd: need 1 <= N <= %d rows, in_features a multiple of 4 (got N = %d, I = %d, O = %d)", AFF_NMAX, N, I, O);
    const int dx_blocks = dx ? (I + 63) / 64 : 0;
    const int dw_blocks = dw ? (int)ceil_div64((int64_t)O * I, 4096) : 0;
    hipLaunchKernelGGL(affine_bwd_kernel, dim3((unsigned)(dx_blocks + dw_blocks)), dim3(1024), 0, as_stream(stream), g, x, w, gain, dx, dw, N, I, O, dx_blocks);
    SPI_LAUNCH_CHECK("spi_affine_bwd");
    return SPI_OK;
}

static int affine_jobs_check(const spi_affine_job* jobs, int n_jobs, int N, int I, int64_t xs, const char* who) {
    SPI_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= SPI_AFFINE_MAX_JOBS && N >= 1 && N <= AFF_NMAX && I >= 4 && I % 4 == 0 && xs >= I && xs % 4 == 0,
                "%s: need 1 <= n_jobs <= %d, 1 <= N <= %d rows, in_features a multiple of 4, x_row_stride >= I and a multiple of 4 (got n_jobs = %d, N = %d, I = %d, stride = %lld)",
                who, SPI_AFFINE_MAX_JOBS, AFF_NMAX, n_jobs, N, I, (long long)xs);
    for (int j = 0; j < n_jobs; ++j)
        SPI_REQUIRE(jobs[j].O >= 1 && jobs[j].w && jobs[j].x, "%s: job %d: O >= 1, x and w non-NULL", who, j);      // (parameters are views of one flat buffer: 4-byte aligned, like spi_affine_fwd takes them)
    return SPI_OK;
}

int spi_affine_multi_fwd(const spi_affine_job* jobs, int n_jobs, int N, int I, int64_t x_row_stride, spi_stream_t stream) {
    if (int rc = affine_jobs_check(jobs, n_jobs, N, I, x_row_stride, "spi_affine_multi_fwd")) return rc;
    AffineJobs J;
    J.n = n_jobs;
    int blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        SPI_REQUIRE(jobs[j].y, "spi_affine_multi_fwd: job %d has no output", j);
        J.job[j] = jobs[j]; J.blk0[j] = blocks; blocks += (jobs[j].O + 3) / 4;
    }
    J.blk0[n_jobs] = blocks;
    hipLaunchKernelGGL(affine_multi_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), J, N, I, x_row_stride);
    SPI_LAUNCH_CHECK("spi_affine_multi_fwd");
    return SPI_OK;
}

int spi_affine_multi_bwd(const spi_affine_job* jobs, int n_jobs, int N, int I, int64_t x_row_stride, spi_stream_t stream) {
    if (int rc = affine_jobs_check(jobs, n_jobs, N, I, x_row_stride, "spi_affine_multi_bwd")) return rc;
    AffineJobs J;
    J.n = 0;
    int blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        if (!jobs[j].g || (!jobs[j].dx_acc && !jobs[j].dw)) continue;              // no incoming gradient / nothing wanted: skipped
        J.job[J.n] = jobs[j]; J.blk0[J.n] = blocks;
        blocks += (jobs[j].dx_acc ? (I + 63) / 64 : 0) + (jobs[j].dw ? (int)ceil_div64((int64_t)jobs[j].O * I, 4096) : 0);
        ++J.n;
    }
    if (J.n == 0) return SPI_OK;
    J.blk0[J.n] = blocks;
    hipLaunchKernelGGL(affine_multi_bwd_kernel, dim3((unsigned)blocks), dim3(1024), 0, as_stream(stream), J, N, I, x_row_stride);
    SPI_LAUNCH_CHECK("spi_affine_multi_bwd");
    return SPI_OK;
}

}  // extern "C"
