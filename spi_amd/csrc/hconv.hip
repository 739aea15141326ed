// (this file: hconv_kernel -- stride-1 forward / dgrad; hconv_s2_kernel -- stride-2 conv = dgrad of the transposed conv0; hconv_t2_kernel -- forward of the
//  transposed conv0; hwgrad_kernel -- stride-1 weight gradient; all for fp16 activation tensors, the use_fp16 super-resolution blocks of configs[4])
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
// GEMM orientation: A = pixels (m), B = weights (n = output channel): a lane ends up with runs of 4 consecutive pixels of ONE output channel,
// which the epilogue (noise, bias, activation, gain, clamp on the fp32 accumulators, one rounding) packs into 8-byte LDS writes; the tile
// leaves through LDS in 16-byte pieces.
#include "common.hpp"
#include <algorithm>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));      // (native vector: arrays of HIP's uint4 struct stay allocas and end up in LDS / scratch)

constexpr int HC_TY = 16, HC_TX = 32, HC_BM = 128, HC_KC = 16, HC_NT = 512;
constexpr int HC_IY = HC_TY + 2, HC_IX = HC_TX + 2;
constexpr int HC_IN_CELLS = 2 * HC_IY * HC_IX;                        // 1224 cells of 16 bytes
constexpr int HC_WT_CELLS = 9 * 2 * HC_BM;                            // 2304
constexpr int HC_IN_PASSES = (HC_IN_CELLS + HC_NT - 1) / HC_NT;       // 3
constexpr int HC_WT_PASSES = (HC_WT_CELLS + HC_NT - 1) / HC_NT;       // 5
constexpr int HC_OST = HC_TY * HC_TX + 4;                             // halves per output channel of the epilogue's LDS tile (+4: conflict-free 8-byte writes)
constexpr int HC_OT_BYTES = HC_BM * HC_OST * 2;                       // the epilogue's output tile; the tile's noise values (fp32) sit behind it
constexpr int HC_LDS_BYTES = (2 * (HC_IN_CELLS + HC_WT_CELLS) * 16 > HC_OT_BYTES + HC_TY * HC_TX * 4) ? 2 * (HC_IN_CELLS + HC_WT_CELLS) * 16 : HC_OT_BYTES + HC_TY * HC_TX * 4;

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
    // LDS: two (input patch, weight chunk) buffers during the channel loop; afterwards the same bytes hold the output tile for the transposing epilogue
    __shared__ __attribute__((aligned(16))) u32x4_t smem[HC_LDS_BYTES / 16];
    u32x4_t (*In_s)[HC_IN_CELLS] = reinterpret_cast<u32x4_t (*)[HC_IN_CELLS]>(smem);
    u32x4_t (*Wt_s)[HC_WT_CELLS] = reinterpret_cast<u32x4_t (*)[HC_WT_CELLS]>(smem + 2 * HC_IN_CELLS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    const int rg = wave & 3, ch = wave >> 2;                  // row group (4 rows), output-channel half (64)
    // workgroups go to the 8 XCDs round-robin: give every XCD a contiguous band of tiles, so that x-neighbours (which share the 128-byte lines of a
    // row: a 34-pixel patch row straddles two of them) and the halo rows meet in ONE L2
    int bx = blockIdx.x;
    const int ntile = P.tx * P.ty;
    if ((ntile & 7) == 0) bx = (bx & 7) * (ntile >> 3) + (bx >> 3);
    const int tyi = bx / P.tx, txi = bx - tyi * P.tx;
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

    // one kx column of a chunk: 6 input-row fragments (A operand: m = pixel), 6 weight fragments (B operand: n = output channel), 24 MFMAs in three
    // ky groups of 8.  The operands of a column's FIRST group (4 input rows, 2 weight fragments: `xq`, `wq`) are read one group ahead -- behind the
    // first group of the column before -- so that only the first column of a chunk (its buffer becomes readable at the barrier) waits for LDS.
    // D layout: lane & 31 = output channel, register q = pixel (q & 3) + 8 (q >> 2) + 4 (lane >> 5) of the 32-pixel row.
    half8_t xq[4], wq[2];
    auto preload = [&](int buf) __attribute__((always_inline)) {
        const half8_t* I = reinterpret_cast<const half8_t*>(In_s[buf]) + (fk * HC_IY + rg * 4) * HC_IX + fr;
        const half8_t* Wt = reinterpret_cast<const half8_t*>(Wt_s[buf]) + fk * HC_BM + ch * 64 + fr;
#pragma unroll
        for (int q = 0; q < 4; ++q) xq[q] = I[q * HC_IX];
#pragma unroll
        for (int i = 0; i < 2; ++i) wq[i] = Wt[i * 32];
    };
    auto step = [&](int buf, auto KX_) __attribute__((always_inline)) {
        constexpr int kx = decltype(KX_)::value;
        const half8_t* I = reinterpret_cast<const half8_t*>(In_s[buf]) + (fk * HC_IY + rg * 4) * HC_IX + fr;
        const half8_t* Wt = reinterpret_cast<const half8_t*>(Wt_s[buf]) + fk * HC_BM + ch * 64 + fr;
        half8_t xr4 = I[4 * HC_IX + kx], xr5 = I[5 * HC_IX + kx], w1[2], w2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { w1[i] = Wt[(3 + kx) * 2 * HC_BM + i * 32]; w2[i] = Wt[(6 + kx) * 2 * HC_BM + i * 32]; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[r][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xq[r], wq[i], acc[r][i], 0, 0, 0);
        half8_t nx[4], nw[2];
        if (kx < 2) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) nx[q] = I[q * HC_IX + kx + 1];
#pragma unroll
            for (int i = 0; i < 2; ++i) nw[i] = Wt[(kx + 1) * 2 * HC_BM + i * 32];
            __builtin_amdgcn_sched_barrier(0);
        }
        const half8_t xrow[6] = {xq[0], xq[1], xq[2], xq[3], xr4, xr5};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[r][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xrow[r + 1], w1[i], acc[r][i], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[r][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xrow[r + 2], w2[i], acc[r][i], 0, 0, 0);
        if (kx < 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) xq[q] = nx[q];
#pragma unroll
            for (int i = 0; i < 2; ++i) wq[i] = nw[i];
        }
    };
#define HC_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

    // ---- pipeline (one register stage, two LDS buffers, one barrier per chunk).  Iteration c multiplies chunk c from LDS[c & 1]; behind its first
    //      kx column the registers (chunk c+1, requested an iteration ago) are written to the other buffer -- free since the barrier that ended
    //      iteration c-1 --, behind the second column the loads of chunk c+2 are issued into the same registers.  The fences keep the compiler
    //      from gathering the stores / the packing arithmetic in front of the MFMAs (it waits for the loads wherever it puts them).
    issue(0);
    commit(0);
    if (nchunk > 1) issue(1);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        // (the two waves of a SIMD -- wave w and w + 4, i.e. the two channel halves -- place their commit / issue phases behind DIFFERENT columns:
        //  while one packs and writes, the other one's MFMAs keep the matrix pipe busy)
        preload(buf);
        if (ch == 0) {
            step(buf, std::integral_constant<int, 0>{});
            HC_FENCE();
            if (c + 1 < nchunk) commit(buf ^ 1);
            HC_FENCE();
            step(buf, std::integral_constant<int, 1>{});
            HC_FENCE();
            if (c + 2 < nchunk) issue(c + 2);
            HC_FENCE();
            step(buf, std::integral_constant<int, 2>{});
        } else {
            step(buf, std::integral_constant<int, 0>{});
            HC_FENCE();
            step(buf, std::integral_constant<int, 1>{});
            HC_FENCE();
            if (c + 1 < nchunk) commit(buf ^ 1);
            HC_FENCE();
            if (c + 2 < nchunk) issue(c + 2);
            HC_FENCE();
            step(buf, std::integral_constant<int, 2>{});
        }
        __syncthreads();
    }

    // ---- epilogue: noise, bias, activation, gain, clamp on the accumulators; ONE rounding; the tile goes through LDS as [channel][row][x] halves so
    //      that it leaves in 16-byte pieces (8 pixels of a row) -- straight from the MFMA layout it would be 128 two-byte stores per lane, which
    //      cost a third of the kernel (tools/ubench/hconv_probe.py: 28 of 92 k cycles per block)
    _Float16* Ot = reinterpret_cast<_Float16*>(smem);
    {
        typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
        // The element-wise work runs 128 times per lane at 4 cycles per vector instruction: the uniform decisions are taken ONCE (three code paths),
        // not per element.  mode 0: nothing to apply (data gradient, plain conv).  mode 1: linear / lrelu with 0 <= slope <= 1 and gain > 0 --
        // lrelu(v) * g == max(v * g, v * slope * g), NaN in -> NaN out like the select form; clamp with selects (NaN stays NaN, as torch.clamp).
        // mode 2: everything else through conv_act_gain_clamp.
        const float slope = ep.act == SPI_ACT_LRELU ? ep.alpha : 1.f;
        const bool any_epi = ep.noise || ep.bias || ep.act;
        const int mode = !any_epi ? 0 : ((ep.act == 0 || ((ep.act == SPI_ACT_LINEAR || ep.act == SPI_ACT_LRELU) && slope >= 0.f && slope <= 1.f && ep.gain > 0.f)) ? 1 : 2);
        const float g1 = ep.act ? ep.gain : 1.f, g2 = g1 * slope;
        const float cpos = (ep.act && ep.clamp >= 0.f) ? ep.clamp : __builtin_inff();
        // the tile's noise * strength goes through LDS once (one load per thread) instead of 64 broadcast loads per lane
        float* Nt = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(smem) + HC_OT_BYTES);
        if (ep.noise) {
            const float ng = ep.noise_gain ? ep.noise_gain[0] : 1.f;
            const __amdgpu_buffer_rsrc_t rsN = make_rsrc(ep.noise, (int64_t)HW * 4);
            const int nrow = tid >> 5, nx = tid & 31;          // (pixels beyond the row's end read the next row's noise: they are never stored)
            Nt[tid] = buf_load_f32(rsN, (unsigned)((y0 + nrow) * P.W + x0 + nx) * 4u) * ng;
            __syncthreads();
        }
        auto tile_out = [&](auto MODE) __attribute__((always_inline)) {
            constexpr int M = decltype(MODE)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rg * 4 + r;
                float nz[16];
                if (M != 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4_t t = ep.noise ? *reinterpret_cast<const f32x4_t*>(Nt + row * HC_TX + 8 * j + 4 * fk) : f32x4_t{0.f, 0.f, 0.f, 0.f};
                        nz[4 * j] = t[0]; nz[4 * j + 1] = t[1]; nz[4 * j + 2] = t[2]; nz[4 * j + 3] = t[3];
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int co = ch * 64 + i * 32 + fr;
                    const float b = (M != 0 && ep.bias) ? ep.bias[mb * HC_BM + co] : 0.f;
                    _Float16* dst = Ot + co * HC_OST + row * HC_TX + 4 * fk;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        half4_t h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[r][i][qq * 4 + e];
                            if (M == 1) {
                                v += nz[qq * 4 + e] + b;
                                const float a = v * g1, c = v * g2;
                                const float w = fmaxf(a, c);               // (a NaN v makes both operands NaN: the maximum is NaN, as the select form gives)
                                v = w > cpos ? cpos : (w < -cpos ? -cpos : w);
                            } else if (M == 2) {
                                v += nz[qq * 4 + e] + b;
                                if (ep.act) v = conv_act_gain_clamp(ep.act, ep.alpha, ep.gain, ep.clamp, v);
                            }
                            h[e] = (_Float16)v;
                        }
                        *reinterpret_cast<half4_t*>(dst + 8 * qq) = h;
                    }
                }
            }
        };
        if (mode == 0) tile_out(std::integral_constant<int, 0>{});
        else if (mode == 1) tile_out(std::integral_constant<int, 1>{});
        else tile_out(std::integral_constant<int, 2>{});
    }
    __syncthreads();
    {
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const bool wide = (P.W & 7) == 0 && (HW & 7) == 0 && (reinterpret_cast<uintptr_t>(ob) & 15) == 0;
#pragma unroll 4
        for (int k = 0; k < HC_BM * HC_TY * (HC_TX / 8) / HC_NT; ++k) {
            const int id = tid + k * HC_NT;
            const int xq = id & 3, row = (id >> 2) & (HC_TY - 1), co = id >> 6;
            const int y = y0 + row, x = x0 + xq * 8;
            const u32x2_t* src = reinterpret_cast<const u32x2_t*>(Ot + co * HC_OST + row * HC_TX + xq * 8);
            const u32x2_t lo = src[0], hi = src[1];
            if (y < P.H && x < P.W) {
                _Float16* dst = ob + (int64_t)co * HW + (int64_t)y * P.W + x;
                if (wide) *reinterpret_cast<u32x4_t*>(dst) = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
                else {
                    const unsigned w4[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (x + e < P.W) dst[e] = __builtin_bit_cast(_Float16, (unsigned short)(w4[e >> 1] >> ((e & 1) * 16)));
                }
            }
        }
    }
}
#undef HC_FENCE

// =====================================================================================================================================
// The same direct scheme for a STRIDE-2 3x3 convolution of fp16 tensors -- the data gradient of the use_fp16 blocks' stride-2 transposed conv0
// (dx[i,y,x] = sum_{o,ky,kx} W[o,i,ky,kx] dz[o, 2y+ky, 2x+kx]; conv2d_resample.py:114-131's conv_transpose2d run backwards).  A block owns 8 x 32
// output pixels x 128 channels; the chunk's 17 x 65 patch of dz is staged once as cells, a fragment reads every second cell of a row (pixel 2x + kx;
// 2-way LDS bank conflicts: the matrix pipe is 16 x faster than these reads need).  A wave: 4 rows x 32 pixels x 32 channels, 9 patch-row fragments and
// 3 weight fragments for 12 MFMAs per kx.  Output leaves through LDS in 16-byte pieces (whole rows of one block: no line is shared between blocks).
constexpr int H2_TY = 8, H2_TX = 32;
constexpr int H2_IY = 2 * H2_TY + 1, H2_IX = 2 * H2_TX + 1;
constexpr int H2_IN_CELLS = 2 * H2_IY * H2_IX;                        // 2210
constexpr int H2_IN_PASSES = (H2_IN_CELLS + HC_NT - 1) / HC_NT;       // 5
constexpr int H2_OST = H2_TY * H2_TX + 4;
constexpr int H2_LDS_BYTES = 2 * (H2_IN_CELLS + HC_WT_CELLS) * 16;    // 144 448 (> the 66 560 bytes of the epilogue tile)

struct HConvS2Params {
    int N, nw, Mo, Ci, H, W, IH, IW;    // output H x W, input (gradient operand) IH x IW
    int tx, ty;
    int64_t in_bs, out_bs;
    const int32_t* seg_flags; int nseg;
};

__global__ void __launch_bounds__(HC_NT, 2) hconv_s2_kernel(HConvS2Params P, const _Float16* __restrict__ in, const u32x4_t* __restrict__ wimg,
                                                            _Float16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) u32x4_t smem[H2_LDS_BYTES / 16];
    u32x4_t (*In_s)[H2_IN_CELLS] = reinterpret_cast<u32x4_t (*)[H2_IN_CELLS]>(smem);
    u32x4_t (*Wt_s)[HC_WT_CELLS] = reinterpret_cast<u32x4_t (*)[HC_WT_CELLS]>(smem + 2 * H2_IN_CELLS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    const int rg = wave & 1, cg = wave >> 1;                  // row group (4 rows), output-channel quarter (32)
    int bx = blockIdx.x;
    const int ntile = P.tx * P.ty;
    if ((ntile & 7) == 0) bx = (bx & 7) * (ntile >> 3) + (bx >> 3);          // an XCD owns a band of tiles (see hconv_kernel)
    const int tyi = bx / P.tx, txi = bx - tyi * P.tx;
    const int y0 = tyi * H2_TY, x0 = txi * H2_TX;
    const int mb = blockIdx.y, n = blockIdx.z;
    const int HW = P.H * P.W, IHW = P.IH * P.IW;
    _Float16* ob = out + (int64_t)n * P.out_bs + (int64_t)mb * HC_BM * HW;

    if (P.seg_flags) {                                                        // no flagged gradient segment in the receptive field: zeros
        const int32_t* fl = P.seg_flags + (int64_t)n * P.nseg;
        const int ya = 2 * y0, yb = min(2 * (y0 + H2_TY - 1) + 2, P.IH - 1);
        const int xa = 2 * x0, xb = min(2 * (x0 + H2_TX - 1) + 2, P.IW - 1);
        int any = 0;
        for (int idx = tid; idx < H2_IY * 8; idx += HC_NT) {
            const int y = ya + (idx >> 3);
            if (y <= yb) {
                const int sg = ((y * P.IW + xa) >> 4) + (idx & 7);
                if (sg <= ((y * P.IW + xb) >> 4)) any |= fl[sg];
            }
        }
        if (!__syncthreads_or(any)) {
            const int xe = min(H2_TX, P.W - x0), ye = min(H2_TY, P.H - y0);
            for (int e = tid; e < HC_BM * H2_TY * H2_TX; e += HC_NT) {
                const int x = e % H2_TX, r = (e / H2_TX) % H2_TY, m = e / (H2_TY * H2_TX);
                if (r < ye && x < xe) ob[(int64_t)m * HW + (int64_t)(y0 + r) * P.W + x0 + x] = (_Float16)0.f;
            }
            return;
        }
    }

    const _Float16* inb = in + (int64_t)n * P.in_bs;
    const __amdgpu_buffer_rsrc_t rsI = make_rsrc(inb, P.in_bs * 2);
    unsigned ivoff[H2_IN_PASSES];
#pragma unroll
    for (int p = 0; p < H2_IN_PASSES; ++p) {
        const int cell = tid + p * HC_NT;
        const int g = cell / (H2_IY * H2_IX), rem = cell - g * (H2_IY * H2_IX);
        const int py = rem / H2_IX, px = rem - py * H2_IX;
        const int iy = 2 * y0 + py, ix = 2 * x0 + px;
        const bool ok = cell < H2_IN_CELLS && iy < P.IH && ix < P.IW;
        ivoff[p] = ok ? (unsigned)((g * 8 * IHW + iy * P.IW + ix) * 2) : BUF_OOB;
    }
    const int nchunk = P.Ci / HC_KC;
    const int nwi = P.nw > 1 ? n : 0;
    const u32x4_t* wbase = wimg + ((int64_t)nwi * (P.Mo / HC_BM) + mb) * nchunk * HC_WT_CELLS;
    const int chs2 = __builtin_amdgcn_readfirstlane(IHW * 2);

    unsigned xr[H2_IN_PASSES][8];
    u32x4_t wr[HC_WT_PASSES];
    auto issue = [&](int c) __attribute__((always_inline)) {
        c = min(c, nchunk - 1);
        const int soff0 = __builtin_amdgcn_readfirstlane(c * HC_KC * chs2);
#pragma unroll
        for (int p = 0; p < H2_IN_PASSES; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                xr[p][j] = __builtin_bit_cast(unsigned short, __builtin_amdgcn_raw_buffer_load_b16(rsI, (int)ivoff[p], soff0 + j * chs2, 0));
        const u32x4_t* wp = wbase + (int64_t)c * HC_WT_CELLS;
#pragma unroll
        for (int p = 0; p < HC_WT_PASSES; ++p) wr[p] = wp[min(tid + p * HC_NT, HC_WT_CELLS - 1)];
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < H2_IN_PASSES; ++p) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(xr[p][j]));
            const u32x4_t v = {xr[p][0] | (xr[p][1] << 16), xr[p][2] | (xr[p][3] << 16), xr[p][4] | (xr[p][5] << 16), xr[p][6] | (xr[p][7] << 16)};
            if ((p + 1) * HC_NT <= H2_IN_CELLS || tid + p * HC_NT < H2_IN_CELLS) In_s[buf][tid + p * HC_NT] = v;
        }
#pragma unroll
        for (int p = 0; p < HC_WT_PASSES; ++p)
            if ((p + 1) * HC_NT <= HC_WT_CELLS || tid + p * HC_NT < HC_WT_CELLS) Wt_s[buf][tid + p * HC_NT] = wr[p];
    };

    f32x16 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[r][q] = 0.f;

    auto step = [&](int buf, int kx) __attribute__((always_inline)) {
        const half8_t* I = reinterpret_cast<const half8_t*>(In_s[buf]) + (fk * H2_IY + 8 * rg) * H2_IX + 2 * fr + kx;
        const half8_t* Wt = reinterpret_cast<const half8_t*>(Wt_s[buf]) + fk * HC_BM + cg * 32 + fr;
        half8_t xf[9], wf[3];
#pragma unroll
        for (int q = 0; q < 9; ++q) xf[q] = I[q * H2_IX];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) wf[ky] = Wt[(ky * 3 + kx) * 2 * HC_BM];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[2 * r + ky], wf[ky], acc[r], 0, 0, 0);
    };
#define HC_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    issue(0);
    commit(0);
    if (nchunk > 1) issue(1);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        step(buf, 0);
        HC_FENCE();
        if (c + 1 < nchunk) commit(buf ^ 1);
        HC_FENCE();
        step(buf, 1);
        HC_FENCE();
        if (c + 2 < nchunk) issue(c + 2);
        HC_FENCE();
        step(buf, 2);
        __syncthreads();
    }
#undef HC_FENCE

    // ---- epilogue (a data gradient: nothing to apply): round once, tile through LDS as [channel][row][x], out in 16-byte pieces
    _Float16* Ot = reinterpret_cast<_Float16*>(smem);
    {
        typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
        const int co = cg * 32 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            _Float16* dst = Ot + co * H2_OST + (rg * 4 + r) * H2_TX + 4 * fk;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const half4_t h = {(_Float16)acc[r][qq * 4], (_Float16)acc[r][qq * 4 + 1], (_Float16)acc[r][qq * 4 + 2], (_Float16)acc[r][qq * 4 + 3]};
                *reinterpret_cast<half4_t*>(dst + 8 * qq) = h;
            }
        }
    }
    __syncthreads();
    {
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const bool wide = (P.W & 7) == 0 && (HW & 7) == 0 && (reinterpret_cast<uintptr_t>(ob) & 15) == 0;
#pragma unroll 4
        for (int k = 0; k < HC_BM * H2_TY * (H2_TX / 8) / HC_NT; ++k) {
            const int id = tid + k * HC_NT;
            const int xq = id & 3, row = (id >> 2) & (H2_TY - 1), co = id >> 5;
            const int y = y0 + row, x = x0 + xq * 8;
            const u32x2_t* src = reinterpret_cast<const u32x2_t*>(Ot + co * H2_OST + row * H2_TX + xq * 8);
            const u32x2_t lo = src[0], hi = src[1];
            if (y < P.H && x < P.W) {
                _Float16* dst = ob + (int64_t)co * HW + (int64_t)y * P.W + x;
                if (wide) *reinterpret_cast<u32x4_t*>(dst) = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
                else {
                    const unsigned w4[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (x + e < P.W) dst[e] = __builtin_bit_cast(_Float16, (unsigned short)(w4[e >> 1] >> ((e & 1) * 16)));
                }
            }
        }
    }
}

// =====================================================================================================================================
// ... and for the FORWARD of that stride-2 transposed 3x3 conv (out[o, 2y+ky, 2x+kx] += x[i,y,x] W[o,i,ky,kx]; conv2d_resample.py:114-131).  Output
// parity class (py, px) at class position (Y, X):  out[o, 2Y+py, 2X+px] = sum_{a in A(py), b in B(px)} x[i, Y-a, X-b] W[o,i,py+2a,px+2b],
// A(0) = B(0) = {0, 1}, A(1) = B(1) = {0}: 4 + 2 + 2 + 1 = 9 taps per position, no multiplication by an inserted zero.  The implicit GEMM runs
// the four classes as separate sets of blocks: the two x-parity classes write the ALTERNATE pixels of the same 128-byte lines from different
// blocks -- a third of its time is those stores.  Here a launch handles one row parity PY; a block owns 8 class rows x 32 class
// columns x 128 output channels and BOTH x-parity classes (a wave: 4 rows x 32 columns x 32 channels x 2 classes = 128 accumulators), i.e. whole
// 64-pixel runs of 8 output rows: the two classes are interleaved on their way through LDS and leave as 16-byte pieces.  32-channel chunks
// (36 / 18 MFMAs per wave and barrier); the 9 x 33 input patch of a chunk is staged once for all taps.
constexpr int T2_TY = 8, T2_TX = 32, T2_KC = 32, T2_G = T2_KC / 8;
constexpr int T2_IY = T2_TY + 1, T2_IX = T2_TX + 1;
constexpr int T2_IN_CELLS = T2_G * T2_IY * T2_IX;                       // 1188
constexpr int T2_IN_PASSES = (T2_IN_CELLS + HC_NT - 1) / HC_NT;         // 3
constexpr int T2_OST = T2_TY * 2 * T2_TX + 8;                           // halves per output channel of the epilogue tile (8 rows x 64 pixels, + 16 bytes)
template <int PY> struct T2Cfg {
    static constexpr int NA = PY == 0 ? 2 : 1;                          // row taps (a)
    static constexpr int WT_CELLS = NA * 3 * T2_G * HC_BM;              // weight cells per chunk: [a][kx][g][co]
    static constexpr int WT_PASSES = (WT_CELLS + HC_NT - 1) / HC_NT;    // 6 / 3
    static constexpr int BUF_CELLS = T2_IN_CELLS + WT_CELLS;
    static constexpr int LDS_BYTES = (2 * BUF_CELLS * 16 > HC_BM * T2_OST * 2) ? 2 * BUF_CELLS * 16 : HC_BM * T2_OST * 2;
};

struct HConvT2Params {
    int N, nw, Mo, Ci, H, W, OH, OW;    // input H x W, output OH x OW (= 2H+1, 2W+1)
    int tx, ty;
    int64_t in_bs, out_bs;
    const int32_t* out_flags; int nseg; // needed-output map over flat OUTPUT pixels / 16 (or NULL)
};

// weights -> the fp16 LDS images of both row parities: [nw][mb][py: 6 | 3 taps][chunk][a][kx][g][co][8]  (all of py = 0 first within (nw, mb))
__global__ void __launch_bounds__(256) hconv_t2_weight_kernel(WinoParams P, const float* __restrict__ w, u32x4_t* __restrict__ img, int64_t cells) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= cells) return;
    const int nchunk = P.Ci / T2_KC;
    const int64_t per_mb = (int64_t)9 * T2_G * HC_BM * nchunk;          // cells of one (nw, mb): py = 0 (6 taps) then py = 1 (3 taps)
    const int64_t r = id % per_mb;
    const int64_t mbn = id / per_mb;
    const int mb = (int)(mbn % (P.Mo / HC_BM)), nwi = (int)(mbn / (P.Mo / HC_BM));
    const int64_t n0 = (int64_t)6 * T2_G * HC_BM * nchunk;
    const int py = r < n0 ? 0 : 1;
    const int64_t q = py ? r - n0 : r;
    const int na3 = py ? 3 : 6;
    const int co = (int)(q % HC_BM), g = (int)((q / HC_BM) % T2_G), tap = (int)((q / (T2_G * HC_BM)) % na3), c = (int)(q / ((int64_t)na3 * T2_G * HC_BM));
    const int a = tap / 3, kx = tap % 3, ky = py + 2 * a;
    const float* src = w + (int64_t)nwi * P.wbs + (int64_t)(mb * HC_BM + co) * P.wsm + (int64_t)(c * T2_KC + g * 8) * P.wsc + P.widx[ky * 3 + kx];
    half8_t h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)src[(int64_t)j * P.wsc];
    img[id] = __builtin_bit_cast(u32x4_t, h);
}

struct __attribute__((packed, aligned(2))) T2Piece { unsigned short h[8]; };       // a 16-byte store at 2-byte alignment (output rows are 2 OW bytes long, OW odd)

template <int PY>
__device__ __forceinline__ void hconv_t2_body(const HConvT2Params& P, const _Float16* __restrict__ in, const u32x4_t* __restrict__ wimg,
                                              _Float16* __restrict__ out, u32x4_t* smem, int bx) {
    using Cfg = T2Cfg<PY>;
    constexpr int NA = Cfg::NA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    const int rg = wave & 1, cg = wave >> 1;
    const int ntile = P.tx * P.ty;
    if ((ntile & 7) == 0) bx = (bx & 7) * (ntile >> 3) + (bx >> 3);
    const int tyi = bx / P.tx, txi = bx - tyi * P.tx;
    const int y0 = tyi * T2_TY, x0 = txi * T2_TX;                        // class coordinates
    const int mb = blockIdx.y, n = blockIdx.z;
    const int HW = P.H * P.W;
    const int64_t OHW = (int64_t)P.OH * P.OW;
    _Float16* ob = out + (int64_t)n * P.out_bs + (int64_t)mb * HC_BM * OHW;
    const int nrow = (P.OH - PY + 1) / 2;                               // class rows of this parity
    if (y0 >= nrow) return;

    if (P.out_flags) {                                                  // needed-output map: a tile nobody needs is written as zeros
        const int32_t* fl = P.out_flags + (int64_t)n * P.nseg;
        const int xa = 2 * x0, xb = min(2 * x0 + 2 * T2_TX - 1, P.OW - 1);
        int any = 0;
        for (int idx = tid; idx < T2_TY * 8; idx += HC_NT) {
            const int yo = 2 * (y0 + (idx >> 3)) + PY;
            if (yo < P.OH) {
                const int sg = ((yo * P.OW + xa) >> 4) + (idx & 7);
                if (sg <= ((yo * P.OW + xb) >> 4)) any |= fl[sg];
            }
        }
        if (!__syncthreads_or(any)) {
            for (int e = tid; e < HC_BM * T2_TY * 2 * T2_TX; e += HC_NT) {
                const int x = e % (2 * T2_TX), r = (e / (2 * T2_TX)) % T2_TY, m = e / (T2_TY * 2 * T2_TX);
                const int yo = 2 * (y0 + r) + PY, xo = 2 * x0 + x;
                if (yo < P.OH && xo < P.OW) ob[(int64_t)m * OHW + (int64_t)yo * P.OW + xo] = (_Float16)0.f;
            }
            return;
        }
    }

    auto in_buf = [&](int b) __attribute__((always_inline)) { return smem + b * Cfg::BUF_CELLS; };
    auto wt_buf = [&](int b) __attribute__((always_inline)) { return smem + b * Cfg::BUF_CELLS + T2_IN_CELLS; };
    const _Float16* inb = in + (int64_t)n * P.in_bs;
    const __amdgpu_buffer_rsrc_t rsI = make_rsrc(inb, P.in_bs * 2);
    unsigned ivoff[T2_IN_PASSES];
#pragma unroll
    for (int p = 0; p < T2_IN_PASSES; ++p) {
        const int cell = tid + p * HC_NT;
        const int g = cell / (T2_IY * T2_IX), rem = cell - g * (T2_IY * T2_IX);
        const int pr = rem / T2_IX, pc = rem - pr * T2_IX;
        const int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
        const bool ok = cell < T2_IN_CELLS && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
        ivoff[p] = ok ? (unsigned)((g * 8 * HW + iy * P.W + ix) * 2) : BUF_OOB;
    }
    const int nchunk = P.Ci / T2_KC;
    const int nwi = P.nw > 1 ? n : 0;
    const int64_t per_mb = (int64_t)9 * T2_G * HC_BM * nchunk;
    const u32x4_t* wbase = wimg + ((int64_t)nwi * (P.Mo / HC_BM) + mb) * per_mb + (PY ? (int64_t)6 * T2_G * HC_BM * nchunk : 0);
    const int chs2 = __builtin_amdgcn_readfirstlane(HW * 2);

    unsigned xr[T2_IN_PASSES][8];
    u32x4_t wr[Cfg::WT_PASSES];
    auto issue = [&](int c) __attribute__((always_inline)) {
        c = min(c, nchunk - 1);
        const int soff0 = __builtin_amdgcn_readfirstlane(c * T2_KC * chs2);
#pragma unroll
        for (int p = 0; p < T2_IN_PASSES; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                xr[p][j] = __builtin_bit_cast(unsigned short, __builtin_amdgcn_raw_buffer_load_b16(rsI, (int)ivoff[p], soff0 + j * chs2, 0));
        const u32x4_t* wp = wbase + (int64_t)c * Cfg::WT_CELLS;
#pragma unroll
        for (int p = 0; p < Cfg::WT_PASSES; ++p) wr[p] = wp[min(tid + p * HC_NT, Cfg::WT_CELLS - 1)];
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < T2_IN_PASSES; ++p) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(xr[p][j]));
            const u32x4_t v = {xr[p][0] | (xr[p][1] << 16), xr[p][2] | (xr[p][3] << 16), xr[p][4] | (xr[p][5] << 16), xr[p][6] | (xr[p][7] << 16)};
            if ((p + 1) * HC_NT <= T2_IN_CELLS || tid + p * HC_NT < T2_IN_CELLS) in_buf(buf)[tid + p * HC_NT] = v;
        }
#pragma unroll
        for (int p = 0; p < Cfg::WT_PASSES; ++p)
            if ((p + 1) * HC_NT <= Cfg::WT_CELLS || tid + p * HC_NT < Cfg::WT_CELLS) wt_buf(buf)[tid + p * HC_NT] = wr[p];
    };

    f32x16 acc[2][4];                                                    // [x parity][row]
#pragma unroll
    for (int pxc = 0; pxc < 2; ++pxc)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[pxc][r][q] = 0.f;

    // one K = 16 half of a chunk (channel groups 2 kk + fk) and one column shift b: patch rows rg*4 + r + 1 - a at column fr + 1 - b;
    // b = 0 feeds kx = 0 (x parity 0) and kx = 1 (x parity 1), b = 1 feeds kx = 2 (x parity 0)
    auto step = [&](int buf, int kk, int b) __attribute__((always_inline)) {
        const half8_t* I = reinterpret_cast<const half8_t*>(in_buf(buf)) + ((2 * kk + fk) * T2_IY + rg * 4) * T2_IX + fr + 1 - b;
        const half8_t* Wt = reinterpret_cast<const half8_t*>(wt_buf(buf)) + (2 * kk + fk) * HC_BM + cg * 32 + fr;
        half8_t xf[3 + NA];
#pragma unroll
        for (int q = 0; q < 3 + NA; ++q) xf[q] = I[(q + 2 - NA) * T2_IX];          // patch rows rg*4 + (2 - NA) .. rg*4 + 4
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            if (b == 0) {
                const half8_t w0 = Wt[((a * 3 + 0) * T2_G) * HC_BM], w1 = Wt[((a * 3 + 1) * T2_G) * HC_BM];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const half8_t x = xf[r + 1 - a - (2 - NA)];
                    acc[0][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w0, acc[0][r], 0, 0, 0);
                    acc[1][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w1, acc[1][r], 0, 0, 0);
                }
            } else {
                const half8_t w2 = Wt[((a * 3 + 2) * T2_G) * HC_BM];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[0][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[r + 1 - a - (2 - NA)], w2, acc[0][r], 0, 0, 0);
            }
        }
    };
#define HC_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    issue(0);
    commit(0);
    if (nchunk > 1) issue(1);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        step(buf, 0, 0);
        HC_FENCE();
        if (c + 1 < nchunk) commit(buf ^ 1);
        HC_FENCE();
        step(buf, 0, 1);
        step(buf, 1, 0);
        HC_FENCE();
        if (c + 2 < nchunk) issue(c + 2);
        HC_FENCE();
        step(buf, 1, 1);
        __syncthreads();
    }
#undef HC_FENCE

    // ---- epilogue (no fused tail in transposed mode: the FIR pass owns it): round once; the two x-parity classes are interleaved into whole output
    //      rows in LDS ([channel][8 rows][64 pixels]) and leave in 16-byte pieces
    _Float16* Ot = reinterpret_cast<_Float16*>(smem);
    {
        const int co = cg * 32 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            _Float16* dst = Ot + co * T2_OST + (rg * 4 + r) * 2 * T2_TX + 8 * fk;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                half8_t h;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[2 * e] = (_Float16)acc[0][r][qq * 4 + e]; h[2 * e + 1] = (_Float16)acc[1][r][qq * 4 + e]; }
                *reinterpret_cast<half8_t*>(dst + 16 * qq) = h;
            }
        }
    }
    __syncthreads();
    {
#pragma unroll 4
        for (int k = 0; k < HC_BM * T2_TY * (2 * T2_TX / 8) / HC_NT; ++k) {
            const int id = tid + k * HC_NT;
            const int xq = id & 7, row = (id >> 3) & (T2_TY - 1), co = id >> 6;
            const int yo = 2 * (y0 + row) + PY, xo = 2 * x0 + xq * 8;
            const T2Piece v = __builtin_bit_cast(T2Piece, *reinterpret_cast<const u32x4_t*>(Ot + co * T2_OST + row * 2 * T2_TX + xq * 8));
            if (yo < P.OH && xo < P.OW) {
                _Float16* dst = ob + (int64_t)co * OHW + (int64_t)yo * P.OW + xo;
                if (xo + 8 <= P.OW) {
                    // one 16-byte store at 2-byte alignment (rows are 2 OW bytes, OW odd): the compiler splits a store it knows to be unaligned into
                    // eight 2-byte ones; the hardware (unaligned access mode, the Linux default) takes it whole
                    const u32x4_t vv = __builtin_bit_cast(u32x4_t, v);
                    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(vv) : "memory");
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (xo + e < P.OW) dst[e] = __builtin_bit_cast(_Float16, v.h[e]);
                }
            }
        }
    }
}

// one launch for both row parities: the first gridDim.x / 2 blocks take the even output rows (6 taps: twice the work), the rest the odd ones
__global__ void __launch_bounds__(HC_NT, 2) hconv_t2_kernel(HConvT2Params P, const _Float16* __restrict__ in, const u32x4_t* __restrict__ wimg,
                                                            _Float16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) u32x4_t smem[T2Cfg<0>::LDS_BYTES / 16];
    static_assert(T2Cfg<0>::LDS_BYTES >= T2Cfg<1>::LDS_BYTES, "one LDS block serves both parities");
    const int ntile = P.tx * P.ty;
    if ((int)blockIdx.x < ntile) hconv_t2_body<0>(P, in, wimg, out, smem, (int)blockIdx.x);
    else hconv_t2_body<1>(P, in, wimg, out, smem, (int)blockIdx.x - ntile);
}

// =====================================================================================================================================
// Direct fp16 weight gradient of the same layers: dW[co][tap][ci] += sum_pixels dy[co][y][x] * x[ci][y + ky - 1][x + kx - 1].
// GEMM per tap: m = output channel, n = input channel, K = pixels -- and in NCHW eight consecutive pixels of one channel ARE the 16 bytes an MFMA
// operand lane holds: both tiles go to LDS as straight 16-byte copies (no transpose, no conversion).  The implicit-GEMM wgrad_kernel stages
// x once PER TAP with 2-byte loads and 2-byte LDS stores (0.11 of the fp16 peak); here
//   * a block owns 128 output x 64 input channels x all nine taps (a wave: 32 x 32 x 9 = 144 accumulators) and walks 4-row x 32-pixel tiles:
//     dy tile [co][4][32], x patch [ci][6 rows][2 + 32 + 2 pixels] staged ONCE per tile;
//   * the kx = 0 / 2 operands are the kx = 1 fragment shifted by one pixel: 5 v_alignbit with the dword before / after (two 4-byte LDS reads),
//     instead of shifted copies of the patch; a patch-row fragment triple serves the three dy rows it meets (ky), a dy fragment three patch rows;
//   * one register stage + two LDS buffers, loads and LDS commits interleaved between thirds of a tile's 72 MFMAs (as hconv_kernel);
//   * a block reduces a run of tiles in its accumulators and adds them into the zeroed dW with fp32 atomics (one per element and block).
// Needs W % 32 == 0, H % 4 == 0, Mo % 128 == 0, Ci % 64 == 0 and a dense gradient (masked gradients keep wgrad_kernel's slab list).
constexpr int HW_TY = 4, HW_TX = 32, HW_BM = 128, HW_BN = 64, HW_NT = 512;
constexpr int HW_DY_ST = HW_TY * HW_TX * 2 + 16;       // bytes per output channel of the dy tile (+16: conflict-free 16-byte fragment reads)
constexpr int HW_XROW = 96;                            // bytes per patch row: [12 unused | 4 left halo | 64 main | 4 right halo | 12 unused]
constexpr int HW_X_ST = (HW_TY + 2) * HW_XROW + 16;    // bytes per input channel of the patch
constexpr int HW_DY_BYTES = HW_BM * HW_DY_ST, HW_X_BYTES = HW_BN * HW_X_ST, HW_BUF = HW_DY_BYTES + HW_X_BYTES;

struct HWgradParams {
    int N, Mo, Ci, H, W;
    int tx, tyc, tpb;              // tiles per row, tile rows, tiles per block (a vertical run of one tile column; tyc % tpb == 0)
    int64_t part_stride;           // > 0: partial sums go to `part` (slot stride in floats) with plain stores instead of atomics into dw
    int slots_per_set;             // partial slots per weight set
    int64_t in_bs, out_bs, wbs;    // elements per sample (x, dy), dW elements between samples (0: shared weights)
    int wsm, wsc, widx[9];
};

__global__ void __launch_bounds__(HW_NT, 2) hwgrad_kernel(HWgradParams P, const _Float16* __restrict__ xin, const _Float16* __restrict__ dyin,
                                                          float* __restrict__ dw, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * HW_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fk = lane >> 5;
    const int cof = wave & 3, cih = wave >> 2;
    const int ncib = P.Ci / HW_BN;
    const int co0 = (blockIdx.y / ncib) * HW_BM, ci0 = (blockIdx.y % ncib) * HW_BN;
    const int n = blockIdx.z;
    const int HWp = P.H * P.W;
    // A block walks `tpb` vertically adjacent tiles of one tile column (the two halo rows of the next tile are L2 hits).  Workgroups go to the 8
    // XCDs round-robin: the swizzle gives every XCD a band of whole image rows, so both halves of every 128-byte line (a tile row is 64 bytes) and
    // the halo rows between its blocks are served by ONE L2.
    int cidx = blockIdx.x;
    if ((gridDim.x & 7) == 0) cidx = (cidx & 7) * (gridDim.x >> 3) + (cidx >> 3);
    const int col = cidx % P.tx, seg = cidx / P.tx;
    const int t_beg = 0, t_end = P.tpb;
    const int x0 = col * HW_TX, yseg = seg * P.tpb * HW_TY;
    const _Float16* xb = xin + (int64_t)n * P.in_bs + (int64_t)ci0 * HWp;
    const _Float16* db = dyin + (int64_t)n * P.out_bs + (int64_t)co0 * HWp;

    // ---- staging coordinates relative to the tile origin (constant over the tiles)
    int dy_g[4], dy_l[4];                        // dy: 2048 16-byte pieces
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int id = tid + k * HW_NT, co = id >> 4, row = (id >> 2) & 3, q = id & 3;
        dy_g[k] = co * HWp + row * P.W + q * 8; dy_l[k] = co * HW_DY_ST + row * 64 + q * 16;
    }
    int xm_c[3], xm_r[3], xm_q[3], xm_l[3];      // patch main part: 1536 pieces
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int id = tid + k * HW_NT, ci = id / 24, rem = id - ci * 24;
        xm_c[k] = ci * HWp; xm_r[k] = rem >> 2; xm_q[k] = (rem & 3) * 8; xm_l[k] = HW_DY_BYTES + ci * HW_X_ST + (rem >> 2) * HW_XROW + 16 + (rem & 3) * 16;
    }
    int xh_c[2], xh_r[2], xh_s[2], xh_l[2];      // patch halo dwords: 768 (the second pass is half full)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int id = min(tid + k * HW_NT, 64 * 12 - 1), ci = id / 12, rem = id - ci * 12;
        xh_c[k] = ci * HWp; xh_r[k] = rem >> 1; xh_s[k] = rem & 1; xh_l[k] = HW_DY_BYTES + ci * HW_X_ST + (rem >> 1) * HW_XROW + ((rem & 1) ? 80 : 12);
    }
    u32x4_t rd[4], rm[3];
    unsigned rh[2];
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(xb, (int64_t)HW_BN * HWp * 2);
    auto issue = [&](int t) __attribute__((always_inline)) {
        t = min(t, t_end - 1);
        const int y0 = yseg + t * HW_TY;
        const int org = y0 * P.W + x0;
#pragma unroll
        for (int k = 0; k < 4; ++k) rd[k] = *reinterpret_cast<const u32x4_t*>(db + org + dy_g[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int iy = y0 - 1 + xm_r[k];
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(xb + xm_c[k] + min(max(iy, 0), P.H - 1) * P.W + x0 + xm_q[k]);   // (clamped address, zeroed below:
            const bool ok = iy >= 0 && iy < P.H;                                                                                 //  a predicated load would be waited for at once)
            rm[k] = u32x4_t{ok ? v[0] : 0u, ok ? v[1] : 0u, ok ? v[2] : 0u, ok ? v[3] : 0u};
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int iy = y0 - 1 + xh_r[k];
            const int px = xh_s[k] ? x0 + HW_TX : x0 - 2;
            // (raw buffer load: an out-of-range offset returns 0 in hardware -- a select would make the compiler sink the load into a branch and wait for it there)
            const bool ok = iy >= 0 && iy < P.H && px >= 0 && px < P.W;
            rh[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsX, ok ? (xh_c[k] + iy * P.W + px) * 2 : (int)BUF_OOB, 0, 0);
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        unsigned char* base = smem + buf * HW_BUF;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            asm volatile("" : "+v"(rd[k]));                    // (pins the zero-selects of `issue` and these stores BEHIND the MFMAs they are placed after)
            *reinterpret_cast<u32x4_t*>(base + dy_l[k]) = rd[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            asm volatile("" : "+v"(rm[k]));
            *reinterpret_cast<u32x4_t*>(base + xm_l[k]) = rm[k];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            asm volatile("" : "+v"(rh[k]));
            if (k == 0 || tid < 64 * 12 - HW_NT) *reinterpret_cast<unsigned*>(base + xh_l[k]) = rh[k];
        }
    };

    f32x16 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    // patch rows r' = RA .. RB-1 of one tile: r' meets dy row r = r' - ky
    auto rows = [&](int buf, auto RA_, auto RB_) __attribute__((always_inline)) {
        constexpr int RA = decltype(RA_)::value, RB = decltype(RB_)::value;
        const unsigned char* base = smem + buf * HW_BUF;
        const unsigned char* dyp = base + (cof * 32 + fr) * HW_DY_ST + fk * 16;
        const unsigned char* xp = base + HW_DY_BYTES + (cih * 32 + fr) * HW_X_ST + 16 + fk * 16;
#pragma unroll
        for (int rp = RA; rp < RB; ++rp) {
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
                const unsigned char* xa = xp + rp * HW_XROW + sg * 32;
                const u32x4_t p0 = *reinterpret_cast<const u32x4_t*>(xa);
                const unsigned pv = *reinterpret_cast<const unsigned*>(xa - 4), nx = *reinterpret_cast<const unsigned*>(xa + 16);
                const unsigned s01 = __builtin_amdgcn_alignbit(p0[1], p0[0], 16), s12 = __builtin_amdgcn_alignbit(p0[2], p0[1], 16),
                               s23 = __builtin_amdgcn_alignbit(p0[3], p0[2], 16);
                const u32x4_t xl = {__builtin_amdgcn_alignbit(p0[0], pv, 16), s01, s12, s23};        // pixels x - 1 (kx = 0)
                const u32x4_t xr = {s01, s12, s23, __builtin_amdgcn_alignbit(nx, p0[3], 16)};         // pixels x + 1 (kx = 2)
                const half8_t b0 = __builtin_bit_cast(half8_t, xl), b1 = __builtin_bit_cast(half8_t, p0), b2 = __builtin_bit_cast(half8_t, xr);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int r = rp - ky;
                    if (r < 0 || r >= HW_TY) continue;
                    const half8_t a = *reinterpret_cast<const half8_t*>(dyp + r * 64 + sg * 32);
                    acc[ky][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, acc[ky][0], 0, 0, 0);
                    acc[ky][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, acc[ky][1], 0, 0, 0);
                    acc[ky][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b2, acc[ky][2], 0, 0, 0);
                }
            }
        }
    };
#define HC_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    issue(t_beg);
    commit(0);
    issue(t_beg + 1);
    __syncthreads();
    for (int t = t_beg; t < t_end; ++t) {
        const int buf = (t - t_beg) & 1;
        rows(buf, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
        HC_FENCE();
        if (t + 1 < t_end) commit(buf ^ 1);
        HC_FENCE();
        rows(buf, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
        HC_FENCE();
        if (t + 2 < t_end) issue(t + 2);
        HC_FENCE();
        rows(buf, std::integral_constant<int, 4>{}, std::integral_constant<int, 6>{});
        __syncthreads();
    }
#undef HC_FENCE

    // ---- D layout: lane & 31 = input channel (n), register q = output channel (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
    //      With a partial-sum buffer the block's result is WRITTEN (plain coalesced stores) into its own slot and hwgrad_reduce_kernel sums the
    //      slots: fp32 atomics run at ~0.4 T adds/s whatever the address pattern -- 19 M of them cost a third of the kernel (tools/ubench/hconv_probe.py).
    const bool to_part = P.part_stride > 0;
    const int slot = P.wbs ? (int)blockIdx.x : n * (int)gridDim.x + (int)blockIdx.x;
    float* dwb = (to_part ? part + ((int64_t)(P.wbs ? n : 0) * P.slots_per_set + slot) * P.part_stride : dw + (int64_t)n * P.wbs) + (int64_t)(ci0 + cih * 32 + fr) * P.wsc;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float* dt = dwb + P.widx[ky * 3 + kx];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = co0 + cof * 32 + (q & 3) + 8 * (q >> 2) + 4 * fk;
                if (to_part) dt[(int64_t)co * P.wsm] = acc[ky][kx][q];
                else atomicAdd(dt + (int64_t)co * P.wsm, acc[ky][kx][q]);
            }
        }
}

// dw[set][e] = sum over the set's slots of part[set][slot][e]
__global__ void __launch_bounds__(256) hwgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int64_t E, int slots, int64_t dw_set_stride) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const float* p = part + (int64_t)blockIdx.y * slots * E + e;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int g = 0;
    for (; g + 8 <= slots; g += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += p[(int64_t)(g + j) * E];
    }
    for (; g < slots; ++g) a[0] += p[(int64_t)g * E];
    dw[(int64_t)blockIdx.y * dw_set_stride + e] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
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

bool spi_hwgrad_eligible(const WinoParams& P) {
    return P.Mo % HW_BM == 0 && P.Ci % HW_BN == 0 && P.W % HW_TX == 0 && P.H % HW_TY == 0 && P.H >= 8 && P.in_bs * 2 < (1ll << 31) && P.out_bs * 2 < (1ll << 31) &&
           !P.seg_flags;
}

// grid of the weight-gradient launch: ~256 blocks (one per CU: 145 KB of LDS); a block owns a vertical run of `tpb` tiles, tpb a divisor of the tile rows
static void hwgrad_grid(const WinoParams& Wp, int& gx, int& gy, int& tpb) {
    const int tx = Wp.W / HW_TX, tyc = Wp.H / HW_TY;
    gy = (Wp.Mo / HW_BM) * (Wp.Ci / HW_BN);
    const int target = std::max(1, 256 / std::max(1, gy * Wp.N));
    tpb = tyc;
    for (int d = 2; d <= tyc; ++d)
        if (tyc % d == 0 && tx * (tyc / d) <= target) { tpb = d; break; }                  // smallest run that keeps the grid within the target
    gx = tx * (tyc / tpb);
}

// floats of the partial-sum buffer with which the launch avoids atomics (0 if the shape is not eligible)
int64_t spi_hwgrad_workspace_bytes(const WinoParams& Wp) {
    int gx, gy, tpb; hwgrad_grid(Wp, gx, gy, tpb);
    const int64_t E = (int64_t)Wp.Mo * Wp.Ci * 9;
    return (int64_t)(Wp.nw > 1 ? Wp.N * gx : Wp.N * gx) * E * 4;
}

// x [N, Ci, H, W] and dy [N, Mo, H, W] are fp16 tensors.  With `workspace` (>= spi_hwgrad_workspace_bytes) dw is overwritten with the sum of per-block
// partial sums (deterministic); without it dw must be zeroed and receives fp32 atomics.
int spi_hwgrad_launch(const WinoParams& Wp, const void* x, const void* dy, float* dw, void* workspace, int64_t workspace_bytes, hipStream_t st) {
    HWgradParams P;
    P.N = Wp.N; P.Mo = Wp.Mo; P.Ci = Wp.Ci; P.H = Wp.H; P.W = Wp.W;
    P.tx = Wp.W / HW_TX; P.tyc = Wp.H / HW_TY;
    P.in_bs = Wp.in_bs; P.out_bs = Wp.out_bs; P.wbs = Wp.nw > 1 ? Wp.wbs : 0;
    P.wsm = Wp.wsm; P.wsc = Wp.wsc;
    for (int t = 0; t < 9; ++t) P.widx[t] = Wp.widx[t];
    int gx, gy; hwgrad_grid(Wp, gx, gy, P.tpb);
    const int64_t E = (int64_t)Wp.Mo * Wp.Ci * 9;
    const bool parts = workspace && workspace_bytes >= spi_hwgrad_workspace_bytes(Wp);
    P.part_stride = parts ? E : 0;
    P.slots_per_set = Wp.nw > 1 ? gx : Wp.N * gx;
    hipLaunchKernelGGL(hwgrad_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)Wp.N), dim3(HW_NT), 0, st, P, static_cast<const _Float16*>(x),
                       static_cast<const _Float16*>(dy), dw, static_cast<float*>(workspace));
    if (parts)
        hipLaunchKernelGGL(hwgrad_reduce_kernel, dim3((unsigned)((E + 255) / 256), (unsigned)(Wp.nw > 1 ? Wp.N : 1)), dim3(256), 0, st,
                           static_cast<const float*>(workspace), dw, E, P.slots_per_set, Wp.nw > 1 ? Wp.wbs : 0);
    return SPI_OK;
}

// ---- stride-2 variant: Wp describes the data-gradient problem of a stride-2 transposed 3x3 conv (Wp.H x Wp.W = dx, widx[ky * 3 + kx]); IH x IW = dz
bool spi_hconv_s2_eligible(const WinoParams& P) {
    const int64_t blocks = (int64_t)((P.W + H2_TX - 1) / H2_TX) * ((P.H + H2_TY - 1) / H2_TY) * (P.Mo / HC_BM) * P.N;
    return P.Mo % HC_BM == 0 && P.Ci % HC_KC == 0 && P.in_bs * 2 < (1ll << 31) && P.out_bs * 2 < (1ll << 31) && blocks >= 128 && !P.out_flags;
}

int spi_hconv_s2_launch(const WinoParams& Wp, int IH, int IW, const void* in, const float* w, void* out, void* workspace, hipStream_t st, bool img_ready) {
    HConvS2Params P;
    P.N = Wp.N; P.nw = Wp.nw; P.Mo = Wp.Mo; P.Ci = Wp.Ci; P.H = Wp.H; P.W = Wp.W; P.IH = IH; P.IW = IW;
    P.tx = (Wp.W + H2_TX - 1) / H2_TX; P.ty = (Wp.H + H2_TY - 1) / H2_TY;
    P.in_bs = Wp.in_bs; P.out_bs = Wp.out_bs;
    P.seg_flags = Wp.seg_flags; P.nseg = Wp.nseg;
    u32x4_t* img = static_cast<u32x4_t*>(workspace);
    if (!img_ready) {
        const int64_t cells = (int64_t)Wp.nw * Wp.Mo * Wp.Ci * 9 / 8;
        hipLaunchKernelGGL(hconv_weight_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, Wp, w, img, cells);
    }
    dim3 grid((unsigned)(P.tx * P.ty), (unsigned)(Wp.Mo / HC_BM), (unsigned)Wp.N);
    hipLaunchKernelGGL(hconv_s2_kernel, grid, dim3(HC_NT), 0, st, P, static_cast<const _Float16*>(in), img, static_cast<_Float16*>(out));
    return SPI_OK;
}

// ---- forward of a stride-2 transposed 3x3 conv: Wp.H x Wp.W = INPUT, Wp.Mo / Ci = output / input channels, widx[ky * 3 + kx]
bool spi_hconv_t2_eligible(const WinoParams& P) {
    const int oh = 2 * P.H + 1, ow = 2 * P.W + 1;
    const int64_t blocks = (int64_t)(((ow + 1) / 2 + T2_TX - 1) / T2_TX) * (((oh + 1) / 2 + T2_TY - 1) / T2_TY) * (P.Mo / HC_BM) * P.N;
    return P.Mo % HC_BM == 0 && P.Ci % T2_KC == 0 && P.in_bs * 2 < (1ll << 31) && P.out_bs * 2 < (1ll << 31) && blocks >= 128 && !P.seg_flags;
}

int spi_hconv_t2_launch(const WinoParams& Wp, const void* in, const float* w, void* out, void* workspace, hipStream_t st, bool img_ready) {
    HConvT2Params P;
    P.N = Wp.N; P.nw = Wp.nw; P.Mo = Wp.Mo; P.Ci = Wp.Ci; P.H = Wp.H; P.W = Wp.W; P.OH = 2 * Wp.H + 1; P.OW = 2 * Wp.W + 1;
    P.tx = ((P.OW + 1) / 2 + T2_TX - 1) / T2_TX; P.ty = ((P.OH + 1) / 2 + T2_TY - 1) / T2_TY;
    P.in_bs = Wp.in_bs; P.out_bs = Wp.out_bs;
    P.out_flags = Wp.out_flags; P.nseg = Wp.nseg;
    u32x4_t* img = static_cast<u32x4_t*>(workspace);
    if (!img_ready) {
        const int64_t cells = (int64_t)Wp.nw * Wp.Mo * Wp.Ci * 9 / 8;
        hipLaunchKernelGGL(hconv_t2_weight_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, Wp, w, img, cells);
    }
    dim3 grid((unsigned)(2 * P.tx * P.ty), (unsigned)(Wp.Mo / HC_BM), (unsigned)Wp.N);
    hipLaunchKernelGGL(hconv_t2_kernel, grid, dim3(HC_NT), 0, st, P, static_cast<const _Float16*>(in), img, static_cast<_Float16*>(out));
    return SPI_OK;
}
