// Direct 3x3 (stride 1, pad 1) convolution of fp16 activation tensors on v_mfma_f32_32x32x16_f16 -- the forward and data-gradient passes of the
// use_fp16 super-resolution blocks' conv1 layers (eg3d/training/networks_stylegan2.py:421-436, superresolution.py:271-277; BASELINE configs[4]).
//
// Why not the implicit GEMM of conv.hip: at 16x the fp32 matrix rate a K = 16 slab is 8 MFMAs of 32 cycles per wave, and igemm_kernel stages
// every input element NINE times (once per tap) through registers into LDS -- ~175 instructions of staging, a barrier and an LDS round trip per
// 8 MFMAs: 16 % matrix-pipe busy (profiles/r05h_pmc_conv_fp16_tensors_summary.txt).  Here
//   * a block owns a 16 x 32-pixel output tile x 128 output channels and walks the input channels in chunks of 16; the chunk's 18 x 34 input
//     patch (halo included) is staged ONCE, as 16-byte cells [channel/8][row][x][8 halves] -- the NCHW -> "8 channels of one pixel" transpose
//     happens on the way in (eight 2-byte buffer loads per cell, coalesced along x, hardware zero fill outside the image) -- and all nine taps read
//     their B fragments from it at shifted cell addresses (one ds_read_b128 each);
//   * the weights are converted to fp16 once per launch by hconv_weight_kernel into the exact LDS image ([chunk][tap][channel/8][128 rows][8])
//     in the caller's workspace, so staging them is a straight 16-byte copy (36 KB per chunk);
//   * a wave owns 4 output rows x 32 pixels x 64 output channels: per (kx, chunk) it reads 6 input-row fragments and 6 weight fragments for
//     24 MFMAs (an input-row fragment serves three output rows through ky, a weight fragment four rows): 0.5 LDS reads per MFMA, 72 MFMAs
//     (2 304 matrix cycles) per barrier.
// GEMM orientation as in conv.hip: A = weights (m = output channel), B = pixels; D layout col = lane & 31 (pixel), row = (r & 3) + 8 (r >> 2) +
// 4 (lane >> 5).  fp32 accumulation, one rounding at the store; epilogue (noise, bias, activation, gain, clamp) on the accumulators.
#include "common.hpp"
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));      // (native vector: arrays of HIP's uint4 struct stay allocas and end up in LDS / scratch)

constexpr int HC_TY = 16, HC_TX = 32, HC_BM = 128, HC_KC = 16, HC_NT = 512;
constexpr int HC_IY = HC_TY + 2, HC_IX = HC_TX + 2;
constexpr int HC_IN_CELLS = 2 * HC_IY * HC_IX;                        // 1224 cells of 16 bytes
constexpr int HC_WT_CELLS = 9 * 2 * HC_BM;                            // 2304
constexpr int HC_IN_PASSES = (HC_IN_CELLS + HC_NT - 1) / HC_NT;       // 3
constexpr int HC_WT_PASSES = (HC_WT_CELLS + HC_NT - 1) / HC_NT;       // 5

struct HConvParams {
    int N, nw, Mo, Ci, H, W;
    int tx, ty;                     // tiles per row / column
    int64_t in_bs, out_bs;          // elements per sample
    const int32_t* seg_flags;       // dgrad: zero-segment map of the gradient operand (or NULL)
    const int32_t* out_flags;       // forward: needed-output map (or NULL)
    int nseg;
};

// fp32 weights (any of conv.hip's layouts: w[n*wbs + m*wsm + c*wsc + widx[tap]]) -> the fp16 LDS image, one 16-byte cell per thread
__global__ void __launch_bounds__(256) hconv_weight_kernel(WinoParams P, const float* __restrict__ w, u32x4_t* __restrict__ img, int64_t cells) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= cells) return;
    const int nchunk = P.Ci / HC_KC;
    const int co = (int)(id % HC_BM);
    const int g = (int)((id / HC_BM) % 2);
    const int tap = (int)((id / (2 * HC_BM)) % 9);
    const int c = (int)((id / HC_WT_CELLS) % nchunk);
    const int mb = (int)((id / ((int64_t)HC_WT_CELLS * nchunk)) % (P.Mo / HC_BM));
    const int nwi = (int)(id / ((int64_t)HC_WT_CELLS * nchunk * (P.Mo / HC_BM)));
    const float* src = w + (int64_t)nwi * P.wbs + (int64_t)(mb * HC_BM + co) * P.wsm + (int64_t)(c * HC_KC + g * 8) * P.wsc + P.widx[tap];
    half8_t h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)src[(int64_t)j * P.wsc];
    img[id] = __builtin_bit_cast(u32x4_t, h);
}

__global__ void __launch_bounds__(HC_NT, 2) hconv_kernel(HConvParams P, const _Float16* __restrict__ in, const u32x4_t* __restrict__ wimg,
                                                         _Float16* __restrict__ out, Epilogue ep) {
    __shared__ __attribute__((aligned(16))) u32x4_t In_s[2][HC_IN_CELLS];
    __shared__ __attribute__((aligned(16))) u32x4_t Wt_s[2][HC_WT_CELLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    const int rg = wave & 3, ch = wave >> 2;                  // row group (4 rows), output-channel half (64)
    const int tyi = blockIdx.x / P.tx, txi = blockIdx.x - tyi * P.tx;
    const int y0 = tyi * HC_TY, x0 = txi * HC_TX;
    const int mb = blockIdx.y, n = blockIdx.z;
    const int HW = P.H * P.W;
    _Float16* ob = out + (int64_t)n * P.out_bs + (int64_t)mb * HC_BM * HW;

    // ---- data-driven skipping (same contract as igemm_kernel): forward tiles nobody needs / dgrad tiles whose receptive field holds no
    //      flagged gradient segment are written as zeros
    if (P.out_flags || P.seg_flags) {
        const int halo = P.seg_flags ? 1 : 0;
        const int32_t* fl = (P.seg_flags ? P.seg_flags : P.out_flags) + (int64_t)n * P.nseg;
        const int ya = max(y0 - halo, 0), yb = min(y0 + HC_TY - 1 + halo, P.H - 1);
        const int xa = max(x0 - halo, 0), xb = min(x0 + HC_TX - 1 + halo, P.W - 1);
        int any = 0;
        for (int idx = tid; idx < (HC_TY + 2) * 4; idx += HC_NT) {
            const int y = ya + (idx >> 2);
            if (y <= yb) {
                const int sg = ((y * P.W + xa) >> 4) + (idx & 3);
                if (sg <= ((y * P.W + xb) >> 4)) any |= fl[sg];
            }
        }
        if (!__syncthreads_or(any)) {
            const int xe = min(HC_TX, P.W - x0), ye = min(HC_TY, P.H - y0);
            if ((P.W & 7) == 0 && (HW & 7) == 0) {              // whole 16-byte runs (x0 is a multiple of 32)
                for (int e = tid; e < HC_BM * HC_TY * (HC_TX / 8); e += HC_NT) {
                    const int xq = e % (HC_TX / 8), r = (e / (HC_TX / 8)) % HC_TY, m = e / (HC_TY * (HC_TX / 8));
                    if (r < ye && xq * 8 < xe)
                        *reinterpret_cast<uint4*>(ob + (int64_t)m * HW + (int64_t)(y0 + r) * P.W + x0 + xq * 8) = make_uint4(0u, 0u, 0u, 0u);
                }
            } else {
                for (int e = tid; e < HC_BM * HC_TY * HC_TX; e += HC_NT) {
                    const int x = e % HC_TX, r = (e / HC_TX) % HC_TY, m = e / (HC_TY * HC_TX);
                    if (r < ye && x < xe) ob[(int64_t)m * HW + (int64_t)(y0 + r) * P.W + x0 + x] = (_Float16)0.f;
                }
            }
            return;
        }
    }

    // ---- staging coordinates (constant over the chunks)
    const _Float16* inb = in + (int64_t)n * P.in_bs;
    const __amdgpu_buffer_rsrc_t rsI = make_rsrc(inb, P.in_bs * 2);
    unsigned ivoff[HC_IN_PASSES];
#pragma unroll
    for (int p = 0; p < HC_IN_PASSES; ++p) {
        const int cell = tid + p * HC_NT;
        const int g = cell / (HC_IY * HC_IX), rem = cell - g * (HC_IY * HC_IX);
        const int py = rem / HC_IX, px = rem - py * HC_IX;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = cell < HC_IN_CELLS && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
        ivoff[p] = ok ? (unsigned)((g * 8 * HW + iy * P.W + ix) * 2) : BUF_OOB;
    }
    const int nchunk = P.Ci / HC_KC;
    const int nwi = P.nw > 1 ? n : 0;
    const u32x4_t* wbase = wimg + ((int64_t)nwi * (P.Mo / HC_BM) + mb) * nchunk * HC_WT_CELLS;
    const int chs2 = __builtin_amdgcn_readfirstlane(HW * 2);

    unsigned xr[HC_IN_PASSES][8];
    u32x4_t wr[HC_WT_PASSES];
    auto issue = [&](int c) __attribute__((always_inline)) {
        c = min(c, nchunk - 1);
        const int soff0 = __builtin_amdgcn_readfirstlane(c * HC_KC * chs2);
#pragma unroll
        for (int p = 0; p < HC_IN_PASSES; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                xr[p][j] = __builtin_bit_cast(unsigned short, __builtin_amdgcn_raw_buffer_load_b16(rsI, (int)ivoff[p], soff0 + j * chs2, 0));      // (zero-extended by the load)
        const u32x4_t* wp = wbase + (int64_t)c * HC_WT_CELLS;
#pragma unroll
        for (int p = 0; p < HC_WT_PASSES; ++p) wr[p] = wp[min(tid + p * HC_NT, HC_WT_CELLS - 1)];
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < HC_IN_PASSES; ++p) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(xr[p][j]));      // pins the packing BEHIND the MFMAs (it is pure arithmetic: the compiler
                                                                                //  otherwise packs right after the loads -- and waits for them there)
            const u32x4_t v = {xr[p][0] | (xr[p][1] << 16), xr[p][2] | (xr[p][3] << 16), xr[p][4] | (xr[p][5] << 16), xr[p][6] | (xr[p][7] << 16)};
            if ((p + 1) * HC_NT <= HC_IN_CELLS || tid + p * HC_NT < HC_IN_CELLS) In_s[buf][tid + p * HC_NT] = v;
        }
#pragma unroll
        for (int p = 0; p < HC_WT_PASSES; ++p)
            if ((p + 1) * HC_NT <= HC_WT_CELLS || tid + p * HC_NT < HC_WT_CELLS) Wt_s[buf][tid + p * HC_NT] = wr[p];
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[r][i][q] = 0.f;

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const half8_t* I = reinterpret_cast<const half8_t*>(In_s[buf]) + (fk * HC_IY + rg * 4) * HC_IX + fr;
        const half8_t* Wt = reinterpret_cast<const half8_t*>(Wt_s[buf]) + fk * HC_BM + ch * 64 + fr;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            half8_t xf[6], wf[3][2];
#pragma unroll
            for (int q = 0; q < 6; ++q) xf[q] = I[q * HC_IX + kx];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[ky][i] = Wt[(ky * 3 + kx) * 2 * HC_BM + i * 32];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[r][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ky][i], xf[r + ky], acc[r][i], 0, 0, 0);
        }
    };

    // ---- pipeline: chunk c+1 is in flight (registers) while chunk c is multiplied; one barrier per chunk
    issue(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) issue(c + 1);
        asm volatile("" ::: "memory");                     // (the compiler otherwise sinks the LDS writes of `commit` in front of the MFMAs --
        __builtin_amdgcn_sched_barrier(0);                 //  and with them the wait for the loads just issued)
        compute(buf);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nchunk) commit(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue
    const float ng = ep.noise ? (ep.noise_gain ? ep.noise_gain[0] : 1.f) : 0.f;
    const __amdgpu_buffer_rsrc_t rsO = make_rsrc(ob, (int64_t)HC_BM * HW * 2);
    const int x = x0 + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + rg * 4 + r;
        const bool ok = y < P.H && x < P.W;
        const int pix = y * P.W + x;
        const float nz = (ep.noise && ok) ? ep.noise[pix] * ng : 0.f;
        const unsigned voff = ok ? (unsigned)((4 * fk * HW + pix) * 2) : BUF_OOB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int mrow = ch * 64 + i * 32 + (q & 3) + 8 * (q >> 2);        // + 4 fk (in voff)
                float v = acc[r][i][q] + nz;
                if (ep.bias) v += ep.bias[mb * HC_BM + mrow + 4 * fk];
                if (ep.act) v = conv_act_gain_clamp(ep.act, ep.alpha, ep.gain, ep.clamp, v);
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), rsO, (int)voff, mrow * chs2, 0);
            }
        }
    }
}

// ---- host side (called from conv.hip; WinoParams carries the problem: 3x3, stride 1, pad 1, weights addressed through wsm / wsc / widx)
int64_t spi_hconv_workspace_bytes(const WinoParams& P) { return (int64_t)P.nw * P.Mo * P.Ci * 9 * 2; }

// (at least half the CUs get a block: smaller problems keep the implicit GEMM, whose 128 x 256 / 64 x 64 tiles spread them over more CUs)
bool spi_hconv_eligible(const WinoParams& P) {
    const int64_t blocks = (int64_t)((P.W + HC_TX - 1) / HC_TX) * ((P.H + HC_TY - 1) / HC_TY) * (P.Mo / HC_BM) * P.N;
    return P.Mo % HC_BM == 0 && P.Ci % HC_KC == 0 && P.in_bs * 2 < (1ll << 31) && P.out_bs * 2 < (1ll << 31) && blocks >= 128;
}

int spi_hconv_launch(const WinoParams& Wp, const void* in, const float* w, void* out, const Epilogue& ep, void* workspace, hipStream_t st, bool img_ready) {
    HConvParams P;
    P.N = Wp.N; P.nw = Wp.nw; P.Mo = Wp.Mo; P.Ci = Wp.Ci; P.H = Wp.H; P.W = Wp.W;
    P.tx = (Wp.W + HC_TX - 1) / HC_TX; P.ty = (Wp.H + HC_TY - 1) / HC_TY;
    P.in_bs = Wp.in_bs; P.out_bs = Wp.out_bs;
    P.seg_flags = Wp.seg_flags; P.out_flags = Wp.out_flags; P.nseg = Wp.nseg;
    u32x4_t* img = static_cast<u32x4_t*>(workspace);
    if (!img_ready) {
        const int64_t cells = (int64_t)Wp.nw * Wp.Mo * Wp.Ci * 9 / 8;
        hipLaunchKernelGGL(hconv_weight_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, Wp, w, img, cells);
    }
    dim3 grid((unsigned)(P.tx * P.ty), (unsigned)(Wp.Mo / HC_BM), (unsigned)Wp.N);
    hipLaunchKernelGGL(hconv_kernel, grid, dim3(HC_NT), 0, st, P, static_cast<const _Float16*>(in), img, static_cast<_Float16*>(out), ep);
    return SPI_OK;
}
