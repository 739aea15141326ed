// Contextual-loss chain of BoxCXLoss (spi/criteria/bbox_cx_loss.py:93-129) as two row-wise kernels per direction.
//
// The reference runs, on a [B, P1, P2] cosine matrix (P1 = P2 = 1600 for the 40x40 VGG19 maps):
//     dist = 1 - sim                                          (compute_cosine_distance, :93-108)
//     rel  = clamp(dist / (min_j dist + 1e-5), -10, 10)       (compute_relative_distance, :111-115)
//     w    = exp((1 - rel) / band_width);  cx = w / sum_j w   (compute_cx, :118-121)
//     out[b] = mean_j max_i cx[b, i, j]                       (compute_cx_loss, :124-129, before the -log)
// i.e. ~10 elementwise / reduction launches forward and ~15 backward over 10 MB per matrix.  Here one wave owns one row i: the
// row's minimum, its normaliser and its cx values are three passes over 6.4 KB that stay in L1 / L2; the column maximum is kept
// in LDS per block of 32 rows (ds_max_u64 on {cx bits, ~i}, so the row index of the maximum comes with it) and a second small kernel
// finishes the maximum over the blocks and the mean.  The backward recomputes cx from the saved row statistics and writes d sim
// once.  HBM-bound: algorithmic bytes = 4 B * P1 * P2 per matrix forward (read sim), 8 B backward (read sim, write d sim).
//
// Ties: torch.amax / amin split the gradient evenly over exact ties, these kernels give it to the first index -- measure zero on
// float features (the same remark as criteria/bbox_cx_loss.py makes for amin vs min(dim)[0]).
#include "common.hpp"

namespace {

constexpr int CX_ROWS = 32;       // rows per block (4 waves x 8 rows)
constexpr int CX_THREADS = 256;
constexpr int CX_CACHE_COLS = 4096;  // widest row whose weights are cached in LDS between the passes (4 strips + column maxima = 96 KB)

// rel and w of one element, written once so that forward and backward round identically
__device__ __forceinline__ float cx_weight(float sim, float m, float inv_bw, float& d, bool& clamped) {
    d = 1.f - sim;
    float t = d / m;
    clamped = t > 10.f || t < -10.f;
    t = t > 10.f ? 10.f : (t < -10.f ? -10.f : t);
    return expf((1.f - t) * inv_bw);
}

// CACHE: the row's weights w_ij of pass 2 are kept in LDS (one P2-float strip per wave) for pass 3; without it (P2 > CX_CACHE_COLS) they are recomputed.
template <bool CACHE>
__global__ void __launch_bounds__(CX_THREADS) cx_rows_kernel(const float* __restrict__ sim, int P1, int P2, float inv_bw, float* __restrict__ row_min,
                                                            float* __restrict__ row_sum, int32_t* __restrict__ row_argmin,
                                                            unsigned long long* __restrict__ part) {
    extern __shared__ unsigned long long colmax[];           // [P2]  {float bits of cx, ~row}   (+ CACHE: 4 x [P2] floats)
    const int b = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* wrow = reinterpret_cast<float*>(colmax + P2) + (int64_t)wv * P2;
    for (int j = threadIdx.x; j < P2; j += CX_THREADS) colmax[j] = 0ull;
    __syncthreads();
    const float* sb = sim + (int64_t)b * P1 * P2;
    for (int r = wv; r < CX_ROWS; r += CX_THREADS / 64) {
        const int i = blk * CX_ROWS + r;
        if (i >= P1) break;
        const float* row = sb + (int64_t)i * P2;
        float best = INFINITY;
        int bj = 0x7fffffff;
        for (int j = lane; j < P2; j += 64) {
            float d = 1.f - row[j];
            if (d < best) { best = d; bj = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(best, o);
            int oj = __shfl_xor(bj, o);
            if (ov < best || (ov == best && oj < bj)) { best = ov; bj = oj; }
        }
        const float m = best + 1e-5f;
        float s = 0.f;
        for (int j = lane; j < P2; j += 64) {
            float d; bool c;
            float w = cx_weight(row[j], m, inv_bw, d, c);
            if (CACHE) wrow[j] = w;                           // read back by the same lane: no barrier
            s += w;
        }
        s = wave_sum(s);
        const unsigned inv_i = 0xffffffffu - (unsigned)i;
        for (int j = lane; j < P2; j += 64) {
            float d; bool c;
            float cx = (CACHE ? wrow[j] : cx_weight(row[j], m, inv_bw, d, c)) / s;
            atomicMax(&colmax[j], ((unsigned long long)__float_as_uint(cx) << 32) | inv_i);     // cx >= 0: its bits order like the value
        }
        if (lane == 0) {
            row_min[(int64_t)b * P1 + i] = best;
            row_sum[(int64_t)b * P1 + i] = s;
            row_argmin[(int64_t)b * P1 + i] = bj;
        }
    }
    __syncthreads();
    unsigned long long* pb = part + ((int64_t)b * gridDim.x + blk) * P2;
    for (int j = threadIdx.x; j < P2; j += CX_THREADS) pb[j] = colmax[j];
}

// maximum over the row blocks, and the mean over the columns.  ONE block per batch element walks all P2 columns and reduces in a fixed
// order, so out[b] has the same bits on every run (round 3 split the columns over ceil(P2 / 256) blocks that finished with a float
// atomicAdd each: the loss's last bits depended on which block arrived first, which made graph-vs-eager comparisons of the stage-2 loss
// flaky -- ADVICE r03).  The work is P2 * nblk 8-byte reads per element: microseconds.
__global__ void __launch_bounds__(CX_THREADS) cx_cols_kernel(const unsigned long long* __restrict__ part, int nblk, int P2, int32_t* __restrict__ col_argmax,
                                                            float* __restrict__ out) {
    __shared__ float red[CX_THREADS / 64];
    const int b = blockIdx.y;
    const unsigned long long* pb = part + (int64_t)b * nblk * P2;
    float s = 0.f;
    for (int j = threadIdx.x; j < P2; j += CX_THREADS) {
        unsigned long long best = 0ull;
        for (int k = 0; k < nblk; ++k) {
            unsigned long long v = pb[(int64_t)k * P2 + j];
            best = v > best ? v : best;
        }
        col_argmax[(int64_t)b * P2 + j] = (int32_t)(0xffffffffu - (unsigned)(best & 0xffffffffull));
        s += __uint_as_float((unsigned)(best >> 32));
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < CX_THREADS / 64; ++k) t += red[k];
        out[b] += t / (float)P2;                             // (out is zeroed by the caller: the contract of round 3 is kept)
    }
}

__global__ void __launch_bounds__(CX_THREADS) cx_bwd_kernel(const float* __restrict__ sim, const float* __restrict__ d_out, int P1, int P2, float inv_bw,
                                                           const float* __restrict__ row_min, const float* __restrict__ row_sum,
                                                           const int32_t* __restrict__ row_argmin, const int32_t* __restrict__ col_argmax,
                                                           float* __restrict__ d_sim) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int i = blockIdx.x * (CX_THREADS / 64) + (threadIdx.x >> 6);
    if (i >= P1) return;
    const float* row = sim + ((int64_t)b * P1 + i) * P2;
    float* drow = d_sim + ((int64_t)b * P1 + i) * P2;
    const int32_t* arg = col_argmax + (int64_t)b * P2;
    const float g = d_out[b] / (float)P2;                    // d out / d (column maximum)
    const float m = row_min[(int64_t)b * P1 + i] + 1e-5f, R = row_sum[(int64_t)b * P1 + i];
    // A = sum_j G_ij cx_ij over the columns whose maximum sits in this row; most rows own no column: their gradient is zero
    float a = 0.f;
    bool any = false;
    for (int j = lane; j < P2; j += 64)
        if (arg[j] == i) {
            float d; bool c;
            a += g * (cx_weight(row[j], m, inv_bw, d, c) / R);
            any = true;
        }
    if (!__any(any)) {
        for (int j = lane; j < P2; j += 64) drow[j] = 0.f;
        return;
    }
    a = wave_sum(a);
    float dm = 0.f;
    for (int j = lane; j < P2; j += 64) {
        float d; bool c;
        float w = cx_weight(row[j], m, inv_bw, d, c);
        float dw = ((arg[j] == i ? g : 0.f) - a) / R;
        float dt = c ? 0.f : dw * w * -inv_bw;
        dm -= dt * d / (m * m);
        drow[j] = -(dt / m);
    }
    dm = wave_sum(dm);
    const int js = row_argmin[(int64_t)b * P1 + i];
    if (lane == (js & 63)) drow[js] -= dm;                   // the lane that wrote column js: d min_j dist -> d sim at the arg-minimum
}

}  // namespace

extern "C" int64_t spi_contextual_workspace_bytes(int B, int P1, int P2) {
    if (B <= 0 || P1 <= 0 || P2 <= 0) return 0;
    return (int64_t)B * ceil_div64(P1, CX_ROWS) * P2 * 8;
}

extern "C" int spi_contextual_fwd(const float* sim, int B, int P1, int P2, float band_width, float* out, float* row_min, float* row_sum,
                                  int32_t* row_argmin, int32_t* col_argmax, void* workspace, spi_stream_t stream) {
    SPI_REQUIRE(sim && out && row_min && row_sum && row_argmin && col_argmax && workspace, "spi_contextual_fwd: null pointer");
    SPI_REQUIRE(B > 0 && P1 > 0 && P2 > 0 && band_width > 0.f, "spi_contextual_fwd: bad shape or band width");
    SPI_REQUIRE(P2 <= 16384 && B <= 65535, "spi_contextual_fwd: P2 %d > 16384 columns (the column maxima of a block live in LDS) or B %d > 65535", P2, B);
    const int nblk = (int)ceil_div64(P1, CX_ROWS);
    const bool cache = P2 <= CX_CACHE_COLS;
    const size_t lds = (size_t)P2 * 8 + (cache ? (size_t)P2 * 4 * (CX_THREADS / 64) : 0);
    const void* fn = cache ? reinterpret_cast<const void*>(cx_rows_kernel<true>) : reinterpret_cast<const void*>(cx_rows_kernel<false>);
    if (lds > 48 * 1024) {                                   // per call, not once per process: the attribute is per device
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { spi_set_error("spi_contextual_fwd: %s", hipGetErrorString(e)); return SPI_ERR_LAUNCH; }
    }
    if (int rc = spi_zero_async(out, B, as_stream(stream))) return rc;
    if (cache)
        hipLaunchKernelGGL(cx_rows_kernel<true>, dim3(nblk, B), dim3(CX_THREADS), lds, as_stream(stream), sim, P1, P2, 1.f / band_width, row_min, row_sum,
                           row_argmin, (unsigned long long*)workspace);
    else
        hipLaunchKernelGGL(cx_rows_kernel<false>, dim3(nblk, B), dim3(CX_THREADS), lds, as_stream(stream), sim, P1, P2, 1.f / band_width, row_min, row_sum,
                           row_argmin, (unsigned long long*)workspace);
    SPI_LAUNCH_CHECK("spi_contextual_fwd");
    hipLaunchKernelGGL(cx_cols_kernel, dim3(1, B), dim3(CX_THREADS), 0, as_stream(stream), (const unsigned long long*)workspace, nblk, P2, col_argmax, out);
    SPI_LAUNCH_CHECK("spi_contextual_fwd");
    return SPI_OK;
}

extern "C" int spi_contextual_bwd(const float* sim, const float* d_out, int B, int P1, int P2, float band_width, const float* row_min, const float* row_sum,
                                  const int32_t* row_argmin, const int32_t* col_argmax, float* d_sim, spi_stream_t stream) {
    SPI_REQUIRE(sim && d_out && row_min && row_sum && row_argmin && col_argmax && d_sim, "spi_contextual_bwd: null pointer");
    SPI_REQUIRE(B > 0 && B <= 65535 && P1 > 0 && P2 > 0 && band_width > 0.f, "spi_contextual_bwd: bad shape or band width");
    hipLaunchKernelGGL(cx_bwd_kernel, dim3((unsigned)ceil_div64(P1, CX_THREADS / 64), B), dim3(CX_THREADS), 0, as_stream(stream), sim, d_out, P1, P2,
                       1.f / band_width, row_min, row_sum, row_argmin, col_argmax, d_sim);
    SPI_LAUNCH_CHECK("spi_contextual_bwd");
    return SPI_OK;
}

// =================================================================================================
// roi_align (torchvision.ops.roi_align with spatial_scale = 1, sampling_ratio = -1, aligned = False -- what BoxCXLoss calls,
// spi/criteria/bbox_cx_loss.py:64-76), one box per image: out[n, c, oy, ox] = mean over the bin's gh x gw bilinear samples of x[n, c].
//   bin size = max(box extent, 1) / out, grid = ceil(bin size) samples per axis, sample s of bin p at lo + (p + (s + 0.5) / grid) * bin;
//   a sample outside [-1, size] contributes 0, one in [-1, 0) is clamped to 0, the last row / column repeats (no interpolation past it).
// One thread per output element, boxes read from device memory (no host-side plan, nothing to synchronise); the backward scatters the same
// weights with atomics into a zeroed gradient.  Replaces ~25 gather / multiply / mean launches per box and direction.
// =================================================================================================
namespace {

struct RoiAxis { int lo, hi; float f; float valid; };
__device__ __forceinline__ RoiAxis roi_axis(float t, int size) {
    RoiAxis a;
    a.valid = (t >= -1.f && t <= (float)size) ? 1.f : 0.f;
    t = fmaxf(t, 0.f);
    int lo = (int)floorf(t);
    if (lo >= size - 1) { lo = size - 1; a.hi = lo; t = (float)lo; } else a.hi = lo + 1;
    a.lo = lo; a.f = t - (float)lo;
    return a;
}

template <bool BWD>
__global__ void __launch_bounds__(256) roi_align_kernel(const float* __restrict__ x, const float* __restrict__ boxes, const float* __restrict__ dy,
                                                       float* __restrict__ out, int N, int C, int H, int W, int O) {
    const int64_t total = (int64_t)N * C * O * O;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        const int ox = (int)(g % O), oy = (int)((g / O) % O);
        const int c = (int)((g / ((int64_t)O * O)) % C), n = (int)(g / ((int64_t)O * O * C));
        const float x1 = boxes[n * 4 + 0], y1 = boxes[n * 4 + 1], x2 = boxes[n * 4 + 2], y2 = boxes[n * 4 + 3];
        const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
        const float bw = rw / (float)O, bh = rh / (float)O;
        const int gw = (int)ceilf(rw / (float)O), gh = (int)ceilf(rh / (float)O);
        const int64_t plane = ((int64_t)n * C + c) * H * W;
        const float scale = 1.f / (float)(gh * gw);
        const float gy = BWD ? dy[g] * scale : 0.f;
        float acc = 0.f;
        for (int sy = 0; sy < gh; ++sy) {
            // (explicitly rounded products / sums: a contracted fma would move the sample by an ulp of its ~100-pixel coordinate against the CPU oracle)
            const RoiAxis ay = roi_axis(__fadd_rn(__fadd_rn(__fmul_rn((float)oy, bh), __fdiv_rn(__fmul_rn((float)sy + 0.5f, bh), (float)gh)), y1), H);
            for (int sx = 0; sx < gw; ++sx) {
                const RoiAxis ax = roi_axis(__fadd_rn(__fadd_rn(__fmul_rn((float)ox, bw), __fdiv_rn(__fmul_rn((float)sx + 0.5f, bw), (float)gw)), x1), W);
                const float v = ay.valid * ax.valid;
                const float w00 = (1.f - ay.f) * (1.f - ax.f), w01 = (1.f - ay.f) * ax.f, w10 = ay.f * (1.f - ax.f), w11 = ay.f * ax.f;
                const int64_t r0 = plane + (int64_t)ay.lo * W, r1 = plane + (int64_t)ay.hi * W;
                if (BWD) {
                    const float gv = gy * v;
                    if (gv != 0.f) {
                        atomicAdd(out + r0 + ax.lo, gv * w00); atomicAdd(out + r0 + ax.hi, gv * w01);
                        atomicAdd(out + r1 + ax.lo, gv * w10); atomicAdd(out + r1 + ax.hi, gv * w11);
                    }
                } else {
                    // (the same association as the gather formulation: rows interpolated in x first, then blended in y)
                    const float top = x[r0 + ax.lo] * (1.f - ax.f) + x[r0 + ax.hi] * ax.f;
                    const float bot = x[r1 + ax.lo] * (1.f - ax.f) + x[r1 + ax.hi] * ax.f;
                    acc += (top * (1.f - ay.f) + bot * ay.f) * v;
                }
            }
        }
        if (!BWD) out[g] = acc * scale;
    }
}

}  // namespace

extern "C" int spi_roi_align_fwd(const float* x, const float* boxes, float* out, int N, int C, int H, int W, int O, spi_stream_t stream) {
    SPI_REQUIRE(x && boxes && out && N > 0 && C > 0 && H > 0 && W > 0 && O > 0, "spi_roi_align_fwd: bad argument");
    const int64_t total = (int64_t)N * C * O * O;
    hipLaunchKernelGGL(roi_align_kernel<false>, dim3((unsigned)std::min<int64_t>(ceil_div64(total, 256), 8192)), dim3(256), 0, as_stream(stream), x, boxes,
                       (const float*)nullptr, out, N, C, H, W, O);
    SPI_LAUNCH_CHECK("spi_roi_align_fwd");
    return SPI_OK;
}

extern "C" int spi_roi_align_bwd(const float* boxes, const float* dy, float* dx, int N, int C, int H, int W, int O, spi_stream_t stream) {
    SPI_REQUIRE(boxes && dy && dx && N > 0 && C > 0 && H > 0 && W > 0 && O > 0, "spi_roi_align_bwd: bad argument");
    if (int rc = spi_zero_async(dx, (int64_t)N * C * H * W, as_stream(stream))) return rc;
    const int64_t total = (int64_t)N * C * O * O;
    hipLaunchKernelGGL(roi_align_kernel<true>, dim3((unsigned)std::min<int64_t>(ceil_div64(total, 256), 8192)), dim3(256), 0, as_stream(stream),
                       (const float*)nullptr, boxes, dy, dx, N, C, H, W, O);
    SPI_LAUNCH_CHECK("spi_roi_align_bwd");
    return SPI_OK;
}
