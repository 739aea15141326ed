// Dense convolutions on the gfx950 matrix cores: implicit-GEMM forward / data-gradient and
// weight-gradient kernels built on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).
//
// One generic formulation covers every conv on SPI's path (modulated 3x3 / 1x1 convs with
// per-sample weights = "groups = batch" in networks_stylegan2.py:85-88, the stride-2 transposed
// 3x3 of the up-sampling layers, conv2d_resample.py:114-131, their data / weight gradients, and the
// shared-weight VGG convs of the losses):
//
//   Out[n, m, (Y*osy+ooy, X*osx+oox)] = sum_{c < Ci} sum_{t < T} A[n, m, c, t] * In[n, c, Y*isy + dy_t, X*isx + dx_t]
//
// with zero outside the input, A addressed through strides (wsm, wsc, widx_t).  A stride-2
// transposed conv is four such problems (one per output parity class, 4+2+2+1 taps): no
// multiplications by inserted zeros.  GEMM view: M = out channels, N = pixels, K = Ci*T.
//   - block tile BM x BN x 16, waves in a WM x WN grid, each wave TM x TN tiles of 32x32
//   - A/B slabs staged through LDS as [k][m] / [k][p] (row stride +4 floats: <= 2-way write
//     conflicts, conflict-free fragment reads), register-prefetched double buffer
//   - epilogue (noise, bias, activation, gain, clamp) applied to the accumulators in registers
#include "common.hpp"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int MAXT = 9;

struct TapSet { int T; int dy[MAXT]; int dx[MAXT]; int widx[MAXT]; };

struct ClassParams {        // one output parity class
    int OHp, OWp;           // class pixel grid
    int ooy, oox;           // output offset
    TapSet taps;
    unsigned magicT;        // ceil(2^32 / T): k / T == umulhi(k, magicT) for k*T < 2^32
};

struct IGemmParams {
    int N, Mo, Ci;
    int IH, IW, OH, OW;
    int isy, isx, osy, osx;
    int64_t wbs; int wsm, wsc;
    int64_t in_bs, out_bs;
    int64_t w_elems;        // elements of one sample's weight tensor (buffer-descriptor range)
    int ncls;
    ClassParams cls[4];
    // optional zero-segment map of the backward passes' gradient operand (spi_conv_desc.dy_seg_flags): [N, nseg] over flat pixels / 16
    const int32_t* seg_flags; int nseg;
    // optional "needed output" map of the forward pass (spi_conv_desc.out_seg_flags): [N, out_nseg] over flat output pixels / 16
    const int32_t* out_flags; int out_nseg;
};

// k -> (channel, tap) for k = c*T + t.  T == 1 is special-cased (ceil(2^32/1) does not fit 32 bits).
__device__ __forceinline__ void split_k(int k, int T, unsigned magic, int& c, int& t) {
    if (T == 1) { c = k; t = 0; }
    else { c = (int)__umulhi((unsigned)k, magic); t = k - c * T; }
}

__device__ __forceinline__ float epilogue_act(const Epilogue& e, float v) {
    return conv_act_gain_clamp(e.act, e.alpha, e.gain, e.clamp, v);
}

// -------------------------------------------------------------------------------------------------
// forward / dgrad implicit GEMM
//   K is walked tap-major: slab s covers tap t = s / nchunk and 16 input channels c0 = (s % nchunk)*16,
//   so the gather offset (dy,dx) and the bounds test are per-slab constants for a thread's pixel and the
//   8 B-loads of a thread differ only by a channel stride (no per-element index arithmetic).
//   SPLITK: blockIdx.z also enumerates K ranges; partial tiles are atomically added into a zeroed
//   output and the epilogue runs as a separate tiny kernel (used when the tile grid alone cannot
//   fill 256 CUs: the 4^2..32^2 layers whose 4608-deep K loop would otherwise run on a few blocks).
// -------------------------------------------------------------------------------------------------
//   AMF (needs BUF): the A-tile loader runs its lanes along m instead of k -- for the dgrad of tap-major
//   weights (m = input channel is the contiguous axis) that turns 16 scattered 4-byte reads into one line.
//   F16 (needs BUF, no SPLITK): operands are rounded to fp16 (nearest even) when they are staged in LDS -- [k/8][m][8] halves, one 16-byte
//   fragment per lane, the same cell layout as a split-bf16 piece -- and a K = 16 slab is ONE gfx950 v_mfma_f32_32x32x16_f16 per 32 x 32 tile
//   with fp32 accumulation (round 3 issued two CDNA3-style v_mfma_f32_32x32x8_f16 per slab: half the matrix rate, twice the fragment reads):
//   the precision of the reference's `use_fp16` super-resolution blocks (superresolution.py:271), at 16x the fp32 matrix rate.
//   PREC 2 / 3 (needs BUF, no SPLITK): "split bf16" -- every fp32 operand is cut into 2 / 3 bf16 pieces when it is staged in LDS
//   (x = c0 + c1 [+ c2] exactly up to 2^-16 / 2^-24 relative: truncating splits, each residual is exact) and the product is the sum of
//   the 3 / 6 significant piece products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: c0*c0 + c0*c1 + c1*c0 [+ c0*c2 + c2*c0 +
//   c1*c1].  bf16 keeps fp32's exponent range, so no scaling is needed (an fp16 split underflows on small gradients).  The bf16
//   matrix rate is 16x the fp32 one: 3 / 6 MFMAs of K = 16 replace 8 of K = 2 (64 cycles each) -- 5.3x / 2.7x less matrix-pipe time
//   at an error of ~2^-16 / ~2^-23 per product.  LDS layout per piece: [k/8][m][8 bf16] = one 16-byte fragment per lane.
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int prec_pieces(int prec) { return prec == 3 ? 3 : (prec == 4 ? 1 : 2); }
constexpr int prec_lds_factor(int prec) { return prec == 3 ? 3 : (prec == 4 ? 1 : 2); }          // LDS halves per element / 1  (fp32 = 2 halves)

// PREC 4 (forward / dgrad, round 4): the fp16 mode of PREC 1 staged like the split-bf16 modes -- every thread owns a RUN of consecutive ks of one row
// / pixel, rounds it to fp16 (nearest even: two v_cvt_f16_f32 + a pack per pair) and writes it as ONE packed LDS store of up to 16 bytes, where
// PREC 1 writes every element with its own 2-byte store (the fp16 kernels are staging-bound: 0.13 of the fp16 matrix peak).  Needs the
// channel-contiguous weight layouts the split modes need; other layouts keep PREC 1.
template <int PER>
__device__ __forceinline__ void store_f16_run(float* lds, int LD, int row, int k0, const float (&v)[PER]) {
    static_assert(PER == 2 || PER == 4 || PER == 8 || PER == 16, "run length");
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    unsigned pk[PER / 2];
#pragma unroll
    for (int i = 0; i < PER / 2; ++i) {
        const half2_t h = {(_Float16)v[2 * i], (_Float16)v[2 * i + 1]};
        pk[i] = __builtin_bit_cast(unsigned, h);
    }
    unsigned* d = reinterpret_cast<unsigned*>(lds) + ((((k0 >> 3) * LD + row) << 3) + (k0 & 7)) / 2;
    if (PER == 2) d[0] = pk[0];
    else if (PER == 4) *reinterpret_cast<uint2*>(d) = make_uint2(pk[0], pk[1]);
    else if (PER == 8) *reinterpret_cast<uint4*>(d) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    else {
        *reinterpret_cast<uint4*>(d) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint4*>(d + LD * 4) = make_uint4(pk[4 % (PER / 2)], pk[5 % (PER / 2)], pk[6 % (PER / 2)], pk[7 % (PER / 2)]);
    }
}

// the same for a run that already IS fp16 (fp16 activations in HBM, `IOH`): pairs are packed as they are -- no conversion
template <int PER>
__device__ __forceinline__ void store_f16_run(float* lds, int LD, int row, int k0, const _Float16 (&v)[PER]) {
    static_assert(PER == 2 || PER == 4 || PER == 8 || PER == 16, "run length");
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    unsigned pk[PER / 2];
#pragma unroll
    for (int i = 0; i < PER / 2; ++i) {
        const half2_t h = {v[2 * i], v[2 * i + 1]};
        pk[i] = __builtin_bit_cast(unsigned, h);
    }
    unsigned* d = reinterpret_cast<unsigned*>(lds) + ((((k0 >> 3) * LD + row) << 3) + (k0 & 7)) / 2;
    if (PER == 2) d[0] = pk[0];
    else if (PER == 4) *reinterpret_cast<uint2*>(d) = make_uint2(pk[0], pk[1]);
    else if (PER == 8) *reinterpret_cast<uint4*>(d) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    else {
        *reinterpret_cast<uint4*>(d) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint4*>(d + LD * 4) = make_uint4(pk[4 % (PER / 2)], pk[5 % (PER / 2)], pk[6 % (PER / 2)], pk[7 % (PER / 2)]);
    }
}

// x -> up to three bf16 pieces (bit patterns), truncating: x == f(c0) + f(c1) + f(c2) + O(2^-24 |x|)
template <int NS>
__device__ __forceinline__ void split_bf16(float x, unsigned short (&c)[3]) {
    const unsigned b0 = __float_as_uint(x) & 0xffff0000u;
    c[0] = (unsigned short)(b0 >> 16);
    const float r1 = x - __uint_as_float(b0);
    const unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
    c[1] = (unsigned short)(b1 >> 16);
    if (NS == 3) {
        const float r2 = r1 - __uint_as_float(b1);
        c[2] = (unsigned short)(__float_as_uint(r2) >> 16);
    } else c[2] = 0;
}

// Split a run of PER consecutive-k fp32 values (same row) and write each piece as ONE packed LDS store: layout per piece
// [k/8][row][8 bf16], `k0` (a multiple of PER) is the run's first k.  Pairs are packed with v_perm_b32 (the high halves of two registers).
template <int NS, int PER>
__device__ __forceinline__ void store_split_run(float* lds, int LD, int row, int k0, const float (&v)[PER]) {
    static_assert(PER == 2 || PER == 4 || PER == 8 || PER == 16, "run length");
    unsigned pk[3][PER / 2];
#pragma unroll
    for (int i = 0; i < PER / 2; ++i) {
        const float x0 = v[2 * i], x1 = v[2 * i + 1];
        pk[0][i] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
        const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u), r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
        pk[1][i] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
        if (NS == 3) {
            const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
            pk[2][i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
        }
    }
    unsigned* base = reinterpret_cast<unsigned*>(lds) + ((((k0 >> 3) * LD + row) << 3) + (k0 & 7)) / 2;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        unsigned* d = base + q * LD * 8;                          // one piece = 2 k-cells x LD rows x 8 halves = LD * 8 dwords
        if (PER == 2) d[0] = pk[q][0];
        else if (PER == 4) *reinterpret_cast<uint2*>(d) = make_uint2(pk[q][0], pk[q][1]);
        else if (PER == 8) *reinterpret_cast<uint4*>(d) = make_uint4(pk[q][0], pk[q][1], pk[q][2], pk[q][3]);
        else {
            *reinterpret_cast<uint4*>(d) = make_uint4(pk[q][0], pk[q][1], pk[q][2], pk[q][3]);
            *reinterpret_cast<uint4*>(d + LD * 4) = make_uint4(pk[q][4 % (PER / 2)], pk[q][5 % (PER / 2)], pk[q][6 % (PER / 2)], pk[q][7 % (PER / 2)]);
        }
    }
}

// piece products in accumulation order (small terms first): NS = 3: (2,0) (0,2) (1,1) (1,0) (0,1) (0,0); NS = 2: (1,0) (0,1) (0,0)
template <int NS> __device__ __forceinline__ constexpr int split_pa(int pi) { return NS == 3 ? (pi == 0 ? 2 : pi == 1 ? 0 : pi == 2 ? 1 : pi == 3 ? 1 : 0) : (pi == 0 ? 1 : 0); }
template <int NS> __device__ __forceinline__ constexpr int split_pb(int pi) { return NS == 3 ? (pi == 0 ? 0 : pi == 1 ? 2 : pi == 2 ? 1 : pi == 3 ? 0 : pi == 4 ? 1 : 0) : (pi == 1 ? 1 : 0); }

// accumulate the significant piece products of one K = 16 slab for one 32x32 tile, small terms first
template <int NS>
__device__ __forceinline__ f32x16 mfma_split(const bf16x8_t (&a)[3], const bf16x8_t (&b)[3], f32x16 acc) {
    if (NS == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

//   IOH (PREC 1 / 4 only, round 5): the ACTIVATIONS -- `in` and `out` -- are fp16 tensors in HBM (the reference's use_fp16 blocks keep them in
//   half precision end to end, networks_stylegan2.py:421-436); weights, bias, noise and the accumulators stay fp32.  Loads are 2-byte
//   buffer loads, a run is packed as it arrives (no conversion), the epilogue rounds once on its way out.
template <int WM, int WN, int TM, int TN, bool SPLITK, bool BUF, bool AMF = false, int PREC = 0, bool IOH = false>
__global__ void __launch_bounds__(64 * WM * WN) igemm_kernel(IGemmParams P, const float* __restrict__ in_,
                                                            const float* __restrict__ wgt, float* __restrict__ out_,
                                                            Epilogue ep, int nsplit) {
    static_assert(!IOH || PREC == 1 || PREC == 4, "fp16 activation tensors go with the fp16 operand modes");
    using AT = std::conditional_t<IOH, _Float16, float>;             // element type of the activation tensors
    constexpr int ES = IOH ? 2 : 4;
    const AT* __restrict__ in = reinterpret_cast<const AT*>(in_);
    AT* __restrict__ out = reinterpret_cast<AT*>(out_);
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int A_PER = BM * BK / NT, B_PER = BN * BK / NT;
    constexpr int A_MSTEP = NT / BK;          // rows of m covered per pass (k fastest)
    constexpr int B_KSTEP = NT / BN;          // k rows covered per pass (p fastest)
    static_assert(NT >= BN && NT % BK == 0 && NT % BN == 0 && A_PER >= 1 && B_PER >= 1, "tile/threads mismatch");
    static_assert(!AMF || (BUF && NT % BM == 0), "m-fast A loads need the buffer path");
    constexpr bool F16 = (PREC == 1);
    constexpr bool F16R = (PREC == 4);                                 // fp16, run-staged (see store_f16_run)
    constexpr bool SPL = (PREC >= 2);                                  // run-staged operands: the split-bf16 modes and F16R
    constexpr int NS = prec_pieces(PREC);
    static_assert(PREC == 0 || (BUF && !SPLITK), "fp16 / split-bf16 operands: buffer path, no split-K");
    constexpr int A_KSTEP = AMF ? NT / BM : 0; // AMF: k rows covered per pass (m fastest)
    constexpr int LDSF = SPL ? prec_lds_factor(PREC) : 2;
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA * LDSF / 2];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB * LDSF / 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int zi = blockIdx.z;
    const int split = SPLITK ? zi % nsplit : 0;
    if (SPLITK) zi /= nsplit;
    const int n = zi / P.ncls, ci = zi % P.ncls;
    const ClassParams& C = P.cls[ci];
    const int npix = C.OHp * C.OWp;
    const int p0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    if (p0 >= npix) return;
    const int T = C.taps.T;
    const int nchunk = (P.Ci + BK - 1) / BK;
    const int nslab_all = T * nchunk;
    int s_beg = 0, s_end = nslab_all;
    if (SPLITK) {
        const int per = (nslab_all + nsplit - 1) / nsplit;
        s_beg = split * per; s_end = min(s_beg + per, nslab_all);
        if (s_beg >= s_end) return;
    }
    const AT* inb = in + (int64_t)n * P.in_bs;
    const float* wb = wgt + (int64_t)n * P.wbs;
    const int64_t chs = (int64_t)P.IH * P.IW;

    // ---- needed-output map (forward only; host guarantees no split-K): tiles without a flagged output segment are not computed
    if (!SPLITK && P.out_flags) {
        const int pl = min(p0 + BN, npix) - 1;
        const int Y0 = p0 / C.OWp, Y1 = pl / C.OWp;
        const int32_t* fl = P.out_flags + (int64_t)n * P.out_nseg;
        int any = 0;
        for (int Yr = Y0; Yr <= Y1; ++Yr) {
            const int xa = (Yr == Y0) ? p0 - Y0 * C.OWp : 0, xb = (Yr == Y1) ? pl - Y1 * C.OWp : C.OWp - 1;
            const int row = (Yr * P.osy + C.ooy) * P.OW + C.oox;
            const int s1 = (row + xb * P.osx) >> 4;
            for (int sg = ((row + xa * P.osx) >> 4) + tid; sg <= s1; sg += NT) any |= fl[sg];
        }
        if (!__syncthreads_or(any)) {
            AT* ob = out + (int64_t)n * P.out_bs;
            for (int e = tid; e < BM * BN; e += NT) {
                const int m = m0 + e / BN, pp = p0 + e % BN;
                if (m < P.Mo && pp < npix) {
                    const int Yo = pp / C.OWp, Xo = pp - Yo * C.OWp;
                    ob[(int64_t)m * P.OH * P.OW + (int64_t)(Yo * P.osy + C.ooy) * P.OW + (Xo * P.osx + C.oox)] = (AT)0.f;
                }
            }
            return;
        }
    }

    // ---- sparse gradient operand (dgrad only; host guarantees ncls == 1, unit output strides, no split-K): when no flagged segment
    //      intersects the tile's receptive field the accumulators would stay exactly 0 -> write the zeros and leave.
    if (!SPLITK && P.seg_flags) {
        const int pl = min(p0 + BN, npix) - 1;
        const int Y0 = p0 / C.OWp, Y1 = pl / C.OWp;
        int dymin = 0, dymax = 0, dxmin = 0, dxmax = 0;
        for (int t = 0; t < T; ++t) {
            dymin = min(dymin, C.taps.dy[t]); dymax = max(dymax, C.taps.dy[t]);
            dxmin = min(dxmin, C.taps.dx[t]); dxmax = max(dxmax, C.taps.dx[t]);
        }
        int xlo = 0, xhi = P.IW - 1;
        if (Y0 == Y1) { xlo = max((p0 - Y0 * C.OWp) * P.isx + dxmin, 0); xhi = min((pl - Y0 * C.OWp) * P.isx + dxmax, P.IW - 1); }
        const int ylo = max(Y0 * P.isy + dymin, 0), yhi = min(Y1 * P.isy + dymax, P.IH - 1);
        const int32_t* fl = P.seg_flags + (int64_t)n * P.nseg;
        int any = 0;
        for (int iy = ylo; iy <= yhi; ++iy) {
            const int s1 = (iy * P.IW + xhi) >> 4;
            for (int sg = ((iy * P.IW + xlo) >> 4) + tid; sg <= s1; sg += NT) any |= fl[sg];
        }
        if (!__syncthreads_or(any)) {
            AT* ob = out + (int64_t)n * P.out_bs;
            for (int e = tid; e < BM * BN; e += NT) {
                const int m = m0 + e / BN, pp = p0 + e % BN;
                if (m < P.Mo && pp < npix) ob[(int64_t)m * npix + pp] = (AT)0.f;
            }
            return;
        }
    }

    // ---- per-thread load coordinates
    // fp32 / fp16: a thread's elements are strided (A: one k, rows A_MSTEP apart -- or with AMF one row, ks A_KSTEP apart; B: one pixel, ks
    // B_KSTEP apart).  Split-bf16: every thread owns a RUN of consecutive ks of one row / pixel (A_PER resp. B_PER long), so that a piece
    // of the run is one packed LDS store.
    constexpr int RA = BK / A_PER;                                   // SPL: k-runs per A row
    const int a_k = SPL ? (AMF ? (tid / BM) * A_PER : (tid % RA) * A_PER) : (AMF ? tid / BM : tid % BK);
    const int a_m = SPL ? (AMF ? tid % BM : tid / RA) : (AMF ? tid % BM : tid / BK);
    const int b_p = tid % BN, b_k = SPL ? (tid / BN) * B_PER : tid / BN;
    const int p = p0 + b_p;
    const bool pv = p < npix;
    const int Y = pv ? p / C.OWp : 0, X = pv ? p - Y * C.OWp : 0;
    const int iy0 = Y * P.isy, ix0 = X * P.isx;

    // Software pipeline (two register sets, two LDS buffers).  During iteration s:
    //   - the loads of slab s+2 are ISSUED piecewise right after the first MFMA groups, so their address
    //     arithmetic runs in the shadow of the matrix pipe (an MFMA occupies the pipe for 64 cycles after a
    //     4-cycle issue) and they have a whole iteration to land;
    //   - slab s+1 (requested one iteration ago) is masked and WRITTEN to the other LDS buffer behind the
    //     last MFMA groups.
    // Loads are unconditional (indices clamped into range, validity kept in a bit mask that is applied at the
    // LDS write): predicated loads turn into exec-masked branch regions that hipcc waits for immediately.
    // BUF (Ci % 16 == 0): loads go through raw buffer descriptors -- out-of-range offsets return 0 in hardware,
    // so rows beyond Mo and taps that fall outside the image need no clamps, masks or selects, and the per-slab
    // part of every address is a SCALAR offset.  That matters because on gfx950 the fp32 MFMA shares the FP32
    // lanes with the VALU: every vector instruction in the loop costs ~2.6 matrix-pipe cycles
    // (tools/ubench/mfma_valu.hip: 157 TF with no VALU beside the MFMAs, 118 TF with 8 per MFMA).
    struct Stage { float ra[A_PER]; AT rb[B_PER]; unsigned am, bm; };
    Stage st0, st1;
    constexpr unsigned OOB = 0x80000000u;
    __amdgpu_buffer_rsrc_t rsA, rsB;
    unsigned voffA[A_PER];
    if (BUF) {
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)wb, 0, (int)(P.w_elems * 4), 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)inb, 0, (int)(P.in_bs * ES), 0x00020000);
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int m = m0 + a_m + ((AMF || SPL) ? 0 : j * A_MSTEP);
            const int k = a_k + (SPL ? j : (AMF ? j * A_KSTEP : 0));
            voffA[j] = m < P.Mo ? (unsigned)((m * P.wsm + k * P.wsc) * 4) : OOB;
        }
    }
    // per-slab addressing state (scalars + one vector offset), computed once per slab by slab_setup()
    const int chs4 = __builtin_amdgcn_readfirstlane((int)chs * ES);              // bytes between input channels (scalar)
    const int kstr4 = __builtin_amdgcn_readfirstlane((SPL ? 1 : B_KSTEP) * (int)chs * ES);    // bytes between the channels one thread loads
    struct SlabAddr { int t, c0; int soffA; unsigned voffB; int soffB; bool okB; int iy, ix; };
    auto slab_setup = [&](int s) {
        SlabAddr q;
        s = min(s, s_end - 1);
        q.t = __builtin_amdgcn_readfirstlane(s / nchunk); q.c0 = __builtin_amdgcn_readfirstlane((s - q.t * nchunk) * BK);
        q.iy = iy0 + C.taps.dy[q.t]; q.ix = ix0 + C.taps.dx[q.t];
        q.okB = pv && q.iy >= 0 && q.iy < P.IH && q.ix >= 0 && q.ix < P.IW;
        q.soffA = __builtin_amdgcn_readfirstlane((q.c0 * P.wsc + C.taps.widx[q.t]) * 4);     // wave-uniform by construction: keep it in an SGPR
        q.voffB = q.okB ? (unsigned)((q.iy * P.IW + q.ix + b_k * (int)chs) * ES) : OOB;
        q.soffB = __builtin_amdgcn_readfirstlane(q.c0 * chs4);
        return q;
    };
    // Split-bf16 forward (k = channel is the contiguous axis of tap-major weights, wsc == 1): a thread's run of A_PER consecutive ks is A_PER / 4
    // 16-byte global loads (rows beyond Mo read row Mo-1 and are zeroed afterwards; Ci % 16 == 0 keeps every k in range).
    constexpr int AVW = A_PER >= 4 ? 4 : A_PER;                      // floats per vector load
    const bool a_row_ok = (m0 + a_m) < P.Mo;
    const float* a_ptr = wb + (int64_t)min(m0 + a_m, P.Mo - 1) * P.wsm + a_k;
    auto load_a = [&](Stage& S, const SlabAddr& q, int j) {
        if constexpr (SPL && !AMF) {
            if (j % AVW == 0) {
                const float* src = a_ptr + (q.soffA >> 2) + j;
                if constexpr (AVW == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(src);
                    S.ra[j] = a_row_ok ? v.x : 0.f; S.ra[j + 1] = a_row_ok ? v.y : 0.f; S.ra[j + 2] = a_row_ok ? v.z : 0.f; S.ra[j + 3] = a_row_ok ? v.w : 0.f;
                } else {
                    const float2 v = *reinterpret_cast<const float2*>(src);
                    S.ra[j] = a_row_ok ? v.x : 0.f; S.ra[j + 1] = a_row_ok ? v.y : 0.f;
                }
            }
        } else if (BUF) {
            S.ra[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, (int)voffA[j], q.soffA, 0));
        } else {
            const int c = q.c0 + a_k;
            const int koff = min(c, P.Ci - 1) * P.wsc + C.taps.widx[q.t];
            const int m = m0 + a_m + j * A_MSTEP;
            S.ra[j] = wb[(int64_t)min(m, P.Mo - 1) * P.wsm + koff];
            S.am |= (unsigned)(c < P.Ci && m < P.Mo) << j;
        }
    };
    auto load_b = [&](Stage& S, const SlabAddr& q, int j) {
        if (BUF) {
            if constexpr (IOH) S.rb[j] = __builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(rsB, (int)q.voffB, q.soffB + j * kstr4, 0));
            else S.rb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, (int)q.voffB, q.soffB + j * kstr4, 0));
        } else {
            const int64_t poff = (int64_t)min(max(q.iy, 0), P.IH - 1) * P.IW + min(max(q.ix, 0), P.IW - 1);
            const int c = q.c0 + b_k + j * B_KSTEP;
            S.rb[j] = inb[(int64_t)min(c, P.Ci - 1) * chs + poff];
            S.bm |= (unsigned)(q.okB && c < P.Ci) << j;
        }
    };
    auto load_all = [&](Stage& S, int s) {
        S.am = S.bm = 0;
        const SlabAddr q = slab_setup(s);
#pragma unroll
        for (int j = 0; j < A_PER; ++j) load_a(S, q, j);
#pragma unroll
        for (int j = 0; j < B_PER; ++j) load_b(S, q, j);
    };
    auto store_a = [&](const Stage& S, int buf, int j) {
        if constexpr (F16R) {
            if (j == 0) store_f16_run<A_PER>(As[buf], LDA, a_m, a_k, S.ra);
        } else if constexpr (SPL) {
            if (j == 0) store_split_run<NS, A_PER>(As[buf], LDA, a_m, a_k, S.ra);
        } else if constexpr (F16) {
            const int k = AMF ? a_k + j * A_KSTEP : a_k, m = AMF ? a_m : a_m + j * A_MSTEP;
            reinterpret_cast<_Float16*>(As[buf])[((k >> 3) * LDA + m) * 8 + (k & 7)] = (_Float16)S.ra[j];
        } else if (AMF) As[buf][(a_k + j * A_KSTEP) * LDA + a_m] = S.ra[j];
        else As[buf][a_k * LDA + a_m + j * A_MSTEP] = (BUF || ((S.am >> j) & 1u)) ? S.ra[j] : 0.f;
    };
    auto store_b = [&](const Stage& S, int buf, int j) {
        if constexpr (F16R) {
            if (j == 0) store_f16_run<B_PER>(Bs[buf], LDB, b_p, b_k, S.rb);
        } else if constexpr (SPL) {
            if (j == 0) store_split_run<NS, B_PER>(Bs[buf], LDB, b_p, b_k, S.rb);
        } else if constexpr (F16) {
            const int k = b_k + j * B_KSTEP;
            reinterpret_cast<_Float16*>(Bs[buf])[((k >> 3) * LDB + b_p) * 8 + (k & 7)] = (_Float16)S.rb[j];
        } else Bs[buf][(b_k + j * B_KSTEP) * LDB + b_p] = (BUF || ((S.bm >> j) & 1u)) ? (float)S.rb[j] : 0.f;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fk = lane >> 5;
    // one pipeline step: MFMAs of slab s from LDS[buf]; issue loads of slab s+2 into L; write slab s+1 from W to LDS[buf^1]
    auto step = [&](int s, int buf, Stage& L, const Stage& W) {
        L.am = L.bm = 0;
        const SlabAddr q2 = slab_setup(s + 2);
        if constexpr (F16R) {
            static_assert(BK == 16, "one v_mfma_f32_32x32x16_f16 per slab");
            const half8_t* Ab = reinterpret_cast<const half8_t*>(As[buf]) + fk * LDA + wm * TM * 32 + fr;
            const half8_t* Bb = reinterpret_cast<const half8_t*>(Bs[buf]) + fk * LDB + wn * TN * 32 + fr;
            half8_t af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = Ab[i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bb[j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i == 0) {                                    // slab s+2 requested behind the first tile row
#pragma unroll
                    for (int j = 0; j < A_PER; ++j) load_a(L, q2, j);
#pragma unroll
                    for (int j = 0; j < B_PER; ++j) load_b(L, q2, j);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < A_PER; ++j) store_a(W, buf ^ 1, j);       // slab s+1: rounded, packed, written to the other buffer
#pragma unroll
            for (int j = 0; j < B_PER; ++j) store_b(W, buf ^ 1, j);
            __syncthreads();
            return;
        }
        if constexpr (SPL) {
            // one K = 16 step per slab: all piece fragments of the wave's tiles (16 B each), then per tile NS*(NS+1)/2 MFMAs.  The loads
            // of slab s+2 are issued behind the first tile row, slab s+1 is split and written to the other LDS buffer behind the last.
            const bf16x8_t* Ab = reinterpret_cast<const bf16x8_t*>(As[buf]) + fk * LDA + wm * TM * 32 + fr;
            const bf16x8_t* Bb = reinterpret_cast<const bf16x8_t*>(Bs[buf]) + fk * LDB + wn * TN * 32 + fr;
            bf16x8_t af[TM][3], bf[TN][3];
#pragma unroll
            for (int q = 0; q < NS; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i][q] = Ab[q * 2 * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j][q] = Bb[q * 2 * LDB + j * 32];
            }
            // product-major order: consecutive MFMAs go to DIFFERENT accumulators (TM*TN of them between two MFMAs on the same one), so
            // the dependent-accumulator latency of the matrix pipe is hidden; small products first.
            constexpr int NP = NS * (NS + 1) / 2;
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) {
                const int ca = split_pa<NS>(pi), cb = split_pb<NS>(pi);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][ca], bf[j][cb], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (pi == 0) {
#pragma unroll
                    for (int j = 0; j < A_PER; ++j) load_a(L, q2, j);
#pragma unroll
                    for (int j = 0; j < B_PER; ++j) load_b(L, q2, j);
                }
                if (pi == NP / 2) {
#pragma unroll
                    for (int j = 0; j < A_PER; ++j) store_a(W, buf ^ 1, j);
                }
                if (pi == NP / 2 + 1) {
#pragma unroll
                    for (int j = 0; j < B_PER; ++j) store_b(W, buf ^ 1, j);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            return;
        }
        if constexpr (F16) {
            // one K = 16 MFMA per tile and slab: lane half fk holds k = 8 fk .. 8 fk + 7 of its row / pixel (one 16-byte LDS read per fragment)
            static_assert(BK == 16, "one v_mfma_f32_32x32x16_f16 per slab");
            const half8_t* Ab = reinterpret_cast<const half8_t*>(As[buf]) + fk * LDA + wm * TM * 32 + fr;
            const half8_t* Bb = reinterpret_cast<const half8_t*>(Bs[buf]) + fk * LDB + wn * TN * 32 + fr;
            half8_t af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = Ab[i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bb[j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < A_PER; ++j) load_a(L, q2, j);              // slab s+2 requested behind the MFMAs
#pragma unroll
            for (int j = 0; j < B_PER; ++j) load_b(L, q2, j);
#pragma unroll
            for (int j = 0; j < A_PER; ++j) store_a(W, buf ^ 1, j);         // slab s+1 rounded and written to the other buffer
#pragma unroll
            for (int j = 0; j < B_PER; ++j) store_b(W, buf ^ 1, j);
            __syncthreads();
            return;
        }
        constexpr int NK = BK / 2;
        using FragT = float;
        const FragT* Ab = reinterpret_cast<const FragT*>(As[buf]) + wm * TM * 32 + fr;
        const FragT* Bb = reinterpret_cast<const FragT*>(Bs[buf]) + wn * TN * 32 + fr;
        // fragments are read one MFMA group ahead (two register sets), so the lgkmcnt wait in front of a
        // group never exposes the LDS latency
        FragT af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = Ab[fk * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = Bb[fk * LDB + j * 32];
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            if (kk + 1 < NK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(kk + 1) & 1][i] = Ab[(2 * (kk + 1) + fk) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[(kk + 1) & 1][j] = Bb[(2 * (kk + 1) + fk) * LDB + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk < NK / 2) {               // first half of the MFMA groups: request slab s+2
#pragma unroll
                for (int j = 0; j < A_PER; ++j) if (j * (NK / 2) / A_PER == kk) load_a(L, q2, j);
#pragma unroll
                for (int j = 0; j < B_PER; ++j) if (j * (NK / 2) / B_PER == kk) load_b(L, q2, j);
            } else {                          // second half: slab s+1 goes to the other LDS buffer
#pragma unroll
                for (int j = 0; j < A_PER; ++j) if (j * (NK / 2) / A_PER == kk - NK / 2) store_a(W, buf ^ 1, j);
#pragma unroll
                for (int j = 0; j < B_PER; ++j) if (j * (NK / 2) / B_PER == kk - NK / 2) store_b(W, buf ^ 1, j);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    // prologue: slab s_beg -> LDS[0]; slab s_beg+1 in flight in st1
    load_all(st0, s_beg);
#pragma unroll
    for (int j = 0; j < A_PER; ++j) store_a(st0, 0, j);
#pragma unroll
    for (int j = 0; j < B_PER; ++j) store_b(st0, 0, j);
    load_all(st1, s_beg + 1);
    __syncthreads();
    for (int s = s_beg; s < s_end; s += 2) {
        step(s, 0, st0, st1);                          // loads s+2 -> st0, writes s+1 (st1) -> LDS[1]
        if (s + 1 < s_end) step(s + 1, 1, st1, st0);   // loads s+3 -> st1, writes s+2 (st0) -> LDS[0]
    }

    // ---- epilogue: C/D layout col = lane & 31 (pixel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (channel)
    const float ng = (!SPLITK && ep.noise) ? (ep.noise_gain ? ep.noise_gain[0] : 1.f) : 0.f;
    AT* ob = out + (int64_t)n * P.out_bs;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int pp = p0 + (wn * TN + j) * 32 + fr;
        if (pp >= npix) continue;
        const int Yo = pp / C.OWp, Xo = pp - Yo * C.OWp;
        const int64_t opix = (int64_t)(Yo * P.osy + C.ooy) * P.OW + (Xo * P.osx + C.oox);
        const float nz = (!SPLITK && ep.noise) ? ep.noise[opix] * ng : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (m < P.Mo) {
                    AT* dst = ob + (int64_t)m * P.OH * P.OW + opix;
                    if constexpr (SPLITK) atomicAdd(dst, acc[i][j][r]);
                    else {
                        float v = acc[i][j][r] + nz;
                        if (ep.bias) v += ep.bias[m];
                        if (ep.act) v = epilogue_act(ep, v);
                        *dst = (AT)v;
                    }
                }
            }
        }
    }
}

// epilogue for the split-K path: y = act(y + noise*gain + bias) over [N, O, HW]
__global__ void conv_epilogue_kernel(float* __restrict__ y, int64_t total, int O, int64_t HW, Epilogue ep) {
    const float ng = ep.noise ? (ep.noise_gain ? ep.noise_gain[0] : 1.f) : 0.f;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = g % HW;
        const int m = (int)((g / HW) % O);
        float v = y[g];
        if (ep.noise) v += ep.noise[pix] * ng;
        if (ep.bias) v += ep.bias[m];
        if (ep.act) v = epilogue_act(ep, v);
        y[g] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// weight gradient: dA[n, m, c, t] = sum_pixels dOut[n, m, out(Y,X)] * In[n, c, in(Y,X,t)]
//   GEMM: rows m, columns j = t*Ci + c (tap-major, so a 128-column tile of a >=128-channel layer has ONE
//   tap and its gather offset is a per-slab constant), reduction over class pixels in slabs of 16,
//   split across blockIdx.x with fp32 atomics into a zeroed dA.  Same machinery as igemm_kernel:
//   raw-buffer loads (hardware zero-fill out of range, no masks), two register stages, loads of slab
//   s+2 issued and slab s+1 written to LDS behind the MFMA groups, fragments read one group ahead.
// -------------------------------------------------------------------------------------------------
//   SPARSE: P.seg_flags marks the 16-pixel segments of dOut that hold a non-zero; a block first compacts the slabs of its pixel
//   range whose dOut segments are flagged into an LDS list and then runs the pipeline over that list only (masked losses: most
//   slabs multiply by an all-zero A operand).
constexpr int WG_LISTMAX = 2048;
//   IOH (PREC 1 only, round 5): `in` and `dout` are fp16 tensors in HBM (see igemm_kernel); dw stays fp32.
template <int WM, int WN, int TM, int TN, int PREC = 0, bool FAST = false, bool SPARSE = false, bool IOH = false>
__global__ void __launch_bounds__(64 * WM * WN) wgrad_kernel(IGemmParams P, const float* __restrict__ in_,
                                                            const float* __restrict__ dout_, float* __restrict__ dw,
                                                            int pix_per_block) {
    static_assert(!IOH || PREC == 1, "fp16 activation tensors go with the fp16 operand mode");
    using AT = std::conditional_t<IOH, _Float16, float>;
    constexpr int ES = IOH ? 2 : 4;
    const AT* __restrict__ in = reinterpret_cast<const AT*>(in_);
    const AT* __restrict__ dout = reinterpret_cast<const AT*>(dout_);
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int A_PER = BM * BK / NT, B_PER = BN * BK / NT;
    constexpr int ROWSTEP = NT / BK;
    constexpr unsigned OOB = 0x40000000u;          // two of them still add up to an out-of-range offset
    constexpr bool F16 = (PREC == 1);
    constexpr bool SPL = (PREC >= 2);
    constexpr int NS = prec_pieces(PREC);
    constexpr int LDSF = SPL ? prec_lds_factor(PREC) : 2;
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA * LDSF / 2];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB * LDSF / 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n = blockIdx.z / P.ncls, ci = blockIdx.z % P.ncls;
    const ClassParams& C = P.cls[ci];
    const int npix = C.OHp * C.OWp;
    const int T = C.taps.T;
    const int Kc = P.Ci * T;                               // columns of this class
    const int ntile_n = (Kc + BN - 1) / BN;
    const int m0 = (blockIdx.y / ntile_n) * BM, j0 = (blockIdx.y % ntile_n) * BN;
    if (m0 >= P.Mo) return;                                 // a class with fewer column tiles than the widest one (transposed convs)
    // dense: blockIdx.x owns a pixel range; SPARSE: an equal share of the class's flagged slabs (ranked below)
    const int pbeg = SPARSE ? 0 : blockIdx.x * pix_per_block, pend = SPARSE ? npix : min(pbeg + pix_per_block, npix);
    if (pbeg >= npix) return;
    const AT* inb = in + (int64_t)n * P.in_bs;
    const AT* dob = dout + (int64_t)n * P.out_bs;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)dob, 0, (int)(P.out_bs * ES), 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)inb, 0, (int)(P.in_bs * ES), 0x00020000);
    auto ld = [&](const __amdgpu_buffer_rsrc_t& rs, int voff, int soff) -> AT {
        if constexpr (IOH) return __builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0));
        else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
    };

    const int l_p = tid % BK;                  // pixel within slab
    const int l_r = tid / BK;                  // first row (m for A, column j for B)
    // rows / columns are fixed per thread for the whole reduction
    unsigned rowA[A_PER], chanB[B_PER];
    int bdy[B_PER], bdx[B_PER];
#pragma unroll
    for (int q = 0; q < A_PER; ++q) {
        const int m = m0 + l_r + q * ROWSTEP;
        rowA[q] = m < P.Mo ? (unsigned)(m * P.OH * P.OW * ES) : OOB;
    }
#pragma unroll
    for (int q = 0; q < B_PER; ++q) {
        const int j = j0 + l_r + q * ROWSTEP;
        const int t = j < Kc ? j / P.Ci : 0;
        const int c = j < Kc ? j - t * P.Ci : 0;
        chanB[q] = j < Kc ? (unsigned)(c * P.IH * P.IW * ES) : OOB;
        bdy[q] = C.taps.dy[t]; bdx[q] = C.taps.dx[t];
    }
    // one tap for the whole tile?  (block-uniform: channel counts that are multiples of the tile width)
    const bool uni = (P.Ci % BN == 0);
    // FAST (host: Ci % BN == 0, Mo % BM == 0, every class row holds >= BK pixels): the per-q part of every address is a multiple
    // of ROWSTEP planes, i.e. wave-uniform -> it rides in the buffer instruction's scalar offset, and the pixel cursor advances
    // incrementally (no division per slab): ~25 vector instructions per slab instead of ~100, all branch-free selects (on
    // gfx950 each VALU instruction beside the fp32 MFMAs costs ~2.6 matrix-pipe cycles).  The scalar offset is not
    // range-checked by the hardware, hence the "all rows / channels exist" condition.
    const int sstepA = __builtin_amdgcn_readfirstlane(ROWSTEP * P.OH * P.OW * ES);
    const int sstepB = __builtin_amdgcn_readfirstlane(ROWSTEP * P.IH * P.IW * ES);
    struct Stage { AT ra[A_PER]; AT rb[B_PER]; };
    Stage st0, st1;
    int curY = 0, curX = 0;                    // FAST: pixel of the next slab to load (load_all is called with pk = pbeg, +BK, +2BK, ...)
    if (FAST) { const int p0 = min(pbeg + l_p, npix - 1); curY = p0 / C.OWp; curX = p0 - curY * C.OWp; }
    auto load_all = [&](Stage& S, int pk) {
        const int p = pk + l_p;
        const bool pvld = p < pend;
        if constexpr (FAST) {
            const int Y = curY, X = curX;
            const int nx = curX + BK;
            const bool wrap = nx >= C.OWp;
            curX = wrap ? nx - C.OWp : nx;
            curY += wrap ? 1 : 0;
            const unsigned pixA = pvld ? (unsigned)(((Y * P.osy + C.ooy) * P.OW + (X * P.osx + C.oox)) * ES) : OOB;
            const int iy = Y * P.isy + bdy[0], ix = X * P.isx + bdx[0];
            const bool inb = pvld & (iy >= 0) & (iy < P.IH) & (ix >= 0) & (ix < P.IW);
            const unsigned pixB = inb ? (unsigned)((iy * P.IW + ix) * ES) : OOB;
            const unsigned voA = pixA + rowA[0], voB = pixB + chanB[0];
#pragma unroll
            for (int q = 0; q < A_PER; ++q)
                S.ra[q] = ld(rsA, (int)voA, q * sstepA);
#pragma unroll
            for (int q = 0; q < B_PER; ++q)
                S.rb[q] = ld(rsB, (int)voB, q * sstepB);
        } else {
        const int pc = min(p, npix - 1);
        const int Y = pc / C.OWp, X = pc - Y * C.OWp;
        const unsigned pixA = pvld ? (unsigned)(((Y * P.osy + C.ooy) * P.OW + (X * P.osx + C.oox)) * ES) : OOB;
#pragma unroll
        for (int q = 0; q < A_PER; ++q)
            S.ra[q] = ld(rsA, (int)(rowA[q] + pixA), 0);
        const int iyb = Y * P.isy, ixb = X * P.isx;
        if (uni) {
            const int iy = iyb + bdy[0], ix = ixb + bdx[0];
            const unsigned pixB = (pvld && iy >= 0 && iy < P.IH && ix >= 0 && ix < P.IW) ? (unsigned)((iy * P.IW + ix) * ES) : OOB;
#pragma unroll
            for (int q = 0; q < B_PER; ++q)
                S.rb[q] = ld(rsB, (int)(chanB[q] + pixB), 0);
        } else {
#pragma unroll
            for (int q = 0; q < B_PER; ++q) {
                const int iy = iyb + bdy[q], ix = ixb + bdx[q];
                const unsigned pixB = (pvld && iy >= 0 && iy < P.IH && ix >= 0 && ix < P.IW) ? (unsigned)((iy * P.IW + ix) * ES) : OOB;
                S.rb[q] = ld(rsB, (int)(chanB[q] + pixB), 0);
            }
        }
        }
    };
    auto store_a = [&](const Stage& S, int buf, int q) {
        if constexpr (SPL) {
            unsigned short c[3];
            split_bf16<NS>((float)S.ra[q], c);
            unsigned short* base = reinterpret_cast<unsigned short*>(As[buf]) + (((l_p >> 3) * LDA + l_r + q * ROWSTEP) << 3) + (l_p & 7);
#pragma unroll
            for (int z = 0; z < NS; ++z) base[z * 2 * LDA * 8] = c[z];
        } else if constexpr (F16) reinterpret_cast<_Float16*>(As[buf])[((l_p >> 3) * LDA + l_r + q * ROWSTEP) * 8 + (l_p & 7)] = (_Float16)S.ra[q];
        else As[buf][l_p * LDA + l_r + q * ROWSTEP] = (float)S.ra[q];
    };
    auto store_b = [&](const Stage& S, int buf, int q) {
        if constexpr (SPL) {
            unsigned short c[3];
            split_bf16<NS>((float)S.rb[q], c);
            unsigned short* base = reinterpret_cast<unsigned short*>(Bs[buf]) + (((l_p >> 3) * LDB + l_r + q * ROWSTEP) << 3) + (l_p & 7);
#pragma unroll
            for (int z = 0; z < NS; ++z) base[z * 2 * LDB * 8] = c[z];
        } else if constexpr (F16) reinterpret_cast<_Float16*>(Bs[buf])[((l_p >> 3) * LDB + l_r + q * ROWSTEP) * 8 + (l_p & 7)] = (_Float16)S.rb[q];
        else Bs[buf][l_p * LDB + l_r + q * ROWSTEP] = (float)S.rb[q];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    static_assert(!(SPARSE && FAST), "the incremental cursor of FAST needs consecutive slabs");
    int nslab = (pend - pbeg + BK - 1) / BK;
    __shared__ unsigned short s_list[SPARSE ? WG_LISTMAX : 1];
    __shared__ int s_scan[SPARSE ? NT : 1];
    if constexpr (SPARSE) {
        // Every block ranks ALL slabs of its (sample, class) that touch a flagged dOut segment (the flag map is a few KB, L2-resident)
        // and takes an equal share of the ranked list: blockIdx.x-th of gridDim.x parts.  Splitting the pixel RANGE instead leaves the
        // blocks of the masked-out rows idle and the ones inside the mask with the whole work.
        const int32_t* fl = P.seg_flags + (int64_t)n * P.nseg;
        auto range_nz = [&](int lo, int hi) { int a = 0; for (int sg = lo >> 4; sg <= (hi >> 4); ++sg) a |= fl[sg]; return a != 0; };
        const int nsl = (npix + BK - 1) / BK;
        const int per = (nsl + NT - 1) / NT;                     // host: <= 64
        const int k0 = tid * per;
        unsigned long long bits = 0ull;
        for (int j = 0; j < per; ++j) {
            const int k = k0 + j;
            bool nz = false;
            if (k < nsl) {
                const int pa = k * BK, pb = min(pa + BK, npix) - 1;
                const int Ya = pa / C.OWp, Xa = pa - Ya * C.OWp, Yb = pb / C.OWp, Xb = pb - Yb * C.OWp;
                const int ra = (Ya * P.osy + C.ooy) * P.OW + C.oox, rb = (Yb * P.osy + C.ooy) * P.OW + C.oox;
                if (Ya == Yb) nz = range_nz(ra + Xa * P.osx, ra + Xb * P.osx);
                else if (Yb == Ya + 1) nz = range_nz(ra + Xa * P.osx, ra + (C.OWp - 1) * P.osx) || range_nz(rb, rb + Xb * P.osx);
                else nz = true;                                  // rows shorter than a slab: no filtering
            }
            bits |= (unsigned long long)nz << j;
        }
        s_scan[tid] = __popcll(bits);
        __syncthreads();
        for (int off = 1; off < NT; off <<= 1) {                 // inclusive scan of the per-thread counts
            const int v = tid >= off ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const int total = s_scan[NT - 1];
        const int beg = (int)((int64_t)blockIdx.x * total / gridDim.x), end = (int)((int64_t)(blockIdx.x + 1) * total / gridDim.x);
        int rank = s_scan[tid] - __popcll(bits);
        while (bits) {
            const int j = __builtin_ctzll(bits);
            bits &= bits - 1ull;
            if (rank >= beg && rank < end) s_list[rank - beg] = (unsigned short)(k0 + j);
            ++rank;
        }
        __syncthreads();
        nslab = end - beg;
        if (nslab == 0) return;                                  // dw was zeroed by the host
    }
    // pixel index of the i-th slab this block reduces (past the end: pend -> every element out of range -> zeros)
    auto slab_pk = [&](int i) {
        if constexpr (SPARSE) return i < nslab ? (int)s_list[i] * BK : pend;
        else return pbeg + i * BK;
    };
    const int fr = lane & 31, fk = lane >> 5;
    auto step = [&](int s, int buf, Stage& L, const Stage& W) {
        if constexpr (SPL) {
            const bf16x8_t* Ab = reinterpret_cast<const bf16x8_t*>(As[buf]) + fk * LDA + wm * TM * 32 + fr;
            const bf16x8_t* Bb = reinterpret_cast<const bf16x8_t*>(Bs[buf]) + fk * LDB + wn * TN * 32 + fr;
            bf16x8_t af[TM][3], bf[TN][3];
#pragma unroll
            for (int z = 0; z < NS; ++z) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i][z] = Ab[z * 2 * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j][z] = Bb[z * 2 * LDB + j * 32];
            }
            constexpr int NP = NS * (NS + 1) / 2;
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) {
                const int ca = split_pa<NS>(pi), cb = split_pb<NS>(pi);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][ca], bf[j][cb], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (pi == 0) load_all(L, slab_pk(s + 2));
                if (pi == NP / 2) {
#pragma unroll
                    for (int q = 0; q < A_PER; ++q) store_a(W, buf ^ 1, q);
                }
                if (pi == NP / 2 + 1) {
#pragma unroll
                    for (int q = 0; q < B_PER; ++q) store_b(W, buf ^ 1, q);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            return;
        }
        if constexpr (F16) {
            static_assert(BK == 16, "one v_mfma_f32_32x32x16_f16 per slab");
            const half8_t* Ab = reinterpret_cast<const half8_t*>(As[buf]) + fk * LDA + wm * TM * 32 + fr;
            const half8_t* Bb = reinterpret_cast<const half8_t*>(Bs[buf]) + fk * LDB + wn * TN * 32 + fr;
            half8_t af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = Ab[i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bb[j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_all(L, slab_pk(s + 2));
#pragma unroll
            for (int q = 0; q < A_PER; ++q) store_a(W, buf ^ 1, q);
#pragma unroll
            for (int q = 0; q < B_PER; ++q) store_b(W, buf ^ 1, q);
            __syncthreads();
            return;
        }
        constexpr int NK = BK / 2;
        using FragT = float;
        const FragT* Ab = reinterpret_cast<const FragT*>(As[buf]) + wm * TM * 32 + fr;
        const FragT* Bb = reinterpret_cast<const FragT*>(Bs[buf]) + wn * TN * 32 + fr;
        FragT af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = Ab[fk * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = Bb[fk * LDB + j * 32];
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            if (kk + 1 < NK) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[(kk + 1) & 1][i] = Ab[(2 * (kk + 1) + fk) * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[(kk + 1) & 1][j] = Bb[(2 * (kk + 1) + fk) * LDB + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk == 0) load_all(L, slab_pk(s + 2));           // slab s+2 (beyond pend: every element out of range -> zeros)
            if (kk >= NK / 2) {
#pragma unroll
                for (int q = 0; q < A_PER; ++q) if (q * (NK / 2) / A_PER == kk - NK / 2) store_a(W, buf ^ 1, q);
#pragma unroll
                for (int q = 0; q < B_PER; ++q) if (q * (NK / 2) / B_PER == kk - NK / 2) store_b(W, buf ^ 1, q);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    load_all(st0, slab_pk(0));
#pragma unroll
    for (int q = 0; q < A_PER; ++q) store_a(st0, 0, q);
#pragma unroll
    for (int q = 0; q < B_PER; ++q) store_b(st0, 0, q);
    load_all(st1, slab_pk(1));
    __syncthreads();
    for (int s = 0; s < nslab; s += 2) {
        step(s, 0, st0, st1);
        if (s + 1 < nslab) step(s + 1, 1, st1, st0);
    }
    float* dwb = dw + (int64_t)n * P.wbs;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = j0 + (wn * TN + j) * 32 + fr;
        if (col >= Kc) continue;
        const int t = col / P.Ci, c = col - t * P.Ci;
        const int koff = c * P.wsc + C.taps.widx[t];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (m < P.Mo) atomicAdd(dwb + (int64_t)m * P.wsm + koff, acc[i][j][r]);
            }
        }
    }
}

// =================================================================================================
// host side: descriptor -> generic problem
// =================================================================================================
static unsigned magic_for(int T) { return (unsigned)(((1ull << 32) + (unsigned)T - 1) / (unsigned)T); }

static int validate(const spi_conv_desc* d, const char* who) {
    SPI_REQUIRE(d != nullptr, "%s: null descriptor", who);
    SPI_REQUIRE(d->N > 0 && d->I > 0 && d->O > 0 && d->H > 0 && d->W > 0, "%s: bad sizes", who);
    SPI_REQUIRE((d->kh == 1 || d->kh == 3) && d->kh == d->kw, "%s: only 1x1 and 3x3 kernels are supported (got %dx%d)", who, d->kh, d->kw);
    SPI_REQUIRE(d->transposed == 0 || d->transposed == 1, "%s: bad transposed flag", who);
    SPI_REQUIRE(d->transposed == 0 || d->pad == 0, "%s: transposed mode takes padding 0", who);
    SPI_REQUIRE(d->pad >= 0 && d->pad < d->kh, "%s: bad padding", who);
    SPI_REQUIRE((int64_t)d->I * d->kh * d->kw < 65536 && (int64_t)d->O * d->kh * d->kw < 65536, "%s: channel count too large", who);
    SPI_REQUIRE(d->compute_f16 >= 0 && d->compute_f16 <= 3, "%s: compute_f16 must be 0 (fp32 MFMA), 1 (fp16), 2 (bf16 x3 split) or 3 (bf16 x6 split)", who);
    SPI_REQUIRE(d->act_dtype == SPI_DTYPE_F32 || (d->act_dtype == SPI_DTYPE_F16 && d->compute_f16 == 1), "%s: act_dtype must be fp32, or fp16 together with compute_f16 = 1", who);
    return SPI_OK;
}

static void out_dims(const spi_conv_desc* d, int& OH, int& OW) {
    if (d->transposed) { OH = 2 * d->H + d->kh - 2; OW = 2 * d->W + d->kw - 2; }
    else { OH = d->H + 2 * d->pad - d->kh + 1; OW = d->W + 2 * d->pad - d->kw + 1; }
}

// widx of tap (ky,kx) honouring the flip flag
static inline int tap_w(const spi_conv_desc* d, int ky, int kx) {
    const int tap = d->flip ? (d->kh - 1 - ky) * d->kw + (d->kw - 1 - kx) : ky * d->kw + kx;
    return d->w_tap_major ? tap * d->I : tap;           // [O, T, I] puts a whole channel row behind every tap
}

// forward problem (also the shape of the weight-gradient problem)
static void make_forward(const spi_conv_desc* d, IGemmParams& P) {
    int OH, OW; out_dims(d, OH, OW);
    const int kk = d->kh * d->kw;
    P.N = d->N; P.Mo = d->O; P.Ci = d->I; P.IH = d->H; P.IW = d->W; P.OH = OH; P.OW = OW;
    P.seg_flags = nullptr; P.nseg = 0;
    P.out_flags = nullptr; P.out_nseg = 0;
    P.wbs = d->w_batch_stride; P.in_bs = (int64_t)d->I * d->H * d->W; P.out_bs = (int64_t)d->O * OH * OW;
    P.w_elems = (int64_t)d->O * d->I * kk;
    if (!d->transposed) {
        P.isy = P.isx = P.osy = P.osx = 1; P.ncls = 1;
        P.wsm = d->I * kk; P.wsc = d->w_tap_major ? 1 : kk;
        ClassParams& C = P.cls[0];
        C.OHp = OH; C.OWp = OW; C.ooy = C.oox = 0; C.taps.T = kk; C.magicT = magic_for(kk);
        for (int ky = 0; ky < d->kh; ++ky) for (int kx = 0; kx < d->kw; ++kx) {
            const int t = ky * d->kw + kx;
            C.taps.dy[t] = ky - d->pad; C.taps.dx[t] = kx - d->pad; C.taps.widx[t] = tap_w(d, ky, kx);
        }
    } else {
        // out[o,Y,X] = sum in[i,y,x] W[o,i,ky,kx], Y = 2y + ky: class (py,px) holds the taps with ky%2==py, kx%2==px
        P.isy = P.isx = 1; P.osy = P.osx = 2; P.ncls = 0;
        P.wsm = d->I * kk; P.wsc = d->w_tap_major ? 1 : kk;
        for (int py = 0; py < 2; ++py) for (int px = 0; px < 2; ++px) {
            ClassParams C; C.taps.T = 0;
            for (int ky = py; ky < d->kh; ky += 2) for (int kx = px; kx < d->kw; kx += 2) {
                const int t = C.taps.T++;
                C.taps.dy[t] = -(ky - py) / 2; C.taps.dx[t] = -(kx - px) / 2; C.taps.widx[t] = tap_w(d, ky, kx);
            }
            C.OHp = (OH - py + 1) / 2; C.OWp = (OW - px + 1) / 2; C.ooy = py; C.oox = px;
            if (C.taps.T == 0 || C.OHp <= 0 || C.OWp <= 0) continue;
            C.magicT = magic_for(C.taps.T);
            P.cls[P.ncls++] = C;
        }
    }
}

// data-gradient problem: input = dy (O channels, OH x OW), output = dx (I channels, H x W)
static void make_dgrad(const spi_conv_desc* d, IGemmParams& P) {
    int OH, OW; out_dims(d, OH, OW);
    const int kk = d->kh * d->kw;
    P.N = d->N; P.Mo = d->I; P.Ci = d->O; P.IH = OH; P.IW = OW; P.OH = d->H; P.OW = d->W;
    P.seg_flags = d->dy_seg_flags; P.nseg = (int)(((int64_t)OH * OW + SPI_SEG_PIXELS - 1) / SPI_SEG_PIXELS);
    P.out_flags = nullptr; P.out_nseg = 0;
    P.wbs = d->w_batch_stride; P.in_bs = (int64_t)d->O * OH * OW; P.out_bs = (int64_t)d->I * d->H * d->W;
    P.w_elems = (int64_t)d->O * d->I * kk;
    P.osy = P.osx = 1; P.ncls = 1;
    ClassParams& C = P.cls[0];
    C.OHp = d->H; C.OWp = d->W; C.ooy = C.oox = 0; C.taps.T = kk; C.magicT = magic_for(kk);
    if (!d->transposed) {
        // dx[i,y,x] = sum_{o,ky,kx} W[o,i,ky,kx] dy[o, y - ky + pad, x - kx + pad]
        P.isy = P.isx = 1; P.wsm = d->w_tap_major ? 1 : kk; P.wsc = d->I * kk;
        for (int ky = 0; ky < d->kh; ++ky) for (int kx = 0; kx < d->kw; ++kx) {
            const int t = ky * d->kw + kx;
            C.taps.dy[t] = d->pad - ky; C.taps.dx[t] = d->pad - kx; C.taps.widx[t] = tap_w(d, ky, kx);
        }
    } else {
        // dx[i,y,x] = sum_{o,ky,kx} W[o,i,ky,kx] dz[o, 2y + ky, 2x + kx]
        P.isy = P.isx = 2; P.wsm = d->w_tap_major ? 1 : kk; P.wsc = d->I * kk;
        for (int ky = 0; ky < d->kh; ++ky) for (int kx = 0; kx < d->kw; ++kx) {
            const int t = ky * d->kw + kx;
            C.taps.dy[t] = ky; C.taps.dx[t] = kx; C.taps.widx[t] = tap_w(d, ky, kx);
        }
    }
}

template <int WM, int WN, int TM, int TN>
static void launch_igemm(const IGemmParams& P, const float* in, const float* w, float* out, const Epilogue& ep, int nsplit, hipStream_t st, int prec, bool ioh) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    int maxpix = 0;
    for (int c = 0; c < P.ncls; ++c) maxpix = std::max(maxpix, P.cls[c].OHp * P.cls[c].OWp);
    dim3 grid((unsigned)((maxpix + BN - 1) / BN), (unsigned)((P.Mo + BM - 1) / BM), (unsigned)(P.N * P.ncls * nsplit));
    // buffer-descriptor fast path: whole channel chunks, and every byte offset fits a signed 32-bit field
    const bool buf = (P.Ci % BK == 0) && (P.in_bs * 4 < (1ll << 31)) && (P.w_elems * 4 < (1ll << 31));
    if (nsplit > 1) {
        if (buf && P.wsm == 1) hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, true, true, true>), grid, dim3(64 * WM * WN), 0, st, P, in, w, out, ep, nsplit);
        else if (buf) hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, true, true>), grid, dim3(64 * WM * WN), 0, st, P, in, w, out, ep, nsplit);
        else hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, true, false>), grid, dim3(64 * WM * WN), 0, st, P, in, w, out, ep, nsplit);
    } else if (prec && buf && (prec == 1 || P.wsm == 1 || P.wsc == 1)) {      // split-bf16 without channels-innermost weights: exact fp32 below
#define SPI_IG_PREC(PR, IOF) do { if (P.wsm == 1) hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, false, true, true, PR, IOF>), grid, dim3(64 * WM * WN), 0, st, P, in, w, out, ep, 1); \
                             else hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, false, true, false, PR, IOF>), grid, dim3(64 * WM * WN), 0, st, P, in, w, out, ep, 1); } while (0)
        if (prec == 1 && ioh) { if (P.wsm == 1 || P.wsc == 1) SPI_IG_PREC(4, true); else SPI_IG_PREC(1, true); }      // fp16 activation tensors
        else if (prec == 1) { if (P.wsm == 1 || P.wsc == 1) SPI_IG_PREC(4, false); else SPI_IG_PREC(1, false); }        // fp16: run-staged when the weights are channel-contiguous
        else if (prec == 2) SPI_IG_PREC(2, false); else SPI_IG_PREC(3, false);
#undef SPI_IG_PREC
    } else {
        if (buf && P.wsm == 1) hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, false, true, true>), grid, dim3(64 * WM * WN), 0, st, P, in, w, out, ep, 1);
        else if (buf) hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, false, true>), grid, dim3(64 * WM * WN), 0, st, P, in, w, out, ep, 1);
        else hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, false, false>), grid, dim3(64 * WM * WN), 0, st, P, in, w, out, ep, 1);
    }
}

// ------------------------------------------------------------------------------------------------
// Round 6: weight gradient of the stride-2 TRANSPOSED 3x3 convolutions (conv0 of every up-sampling block; conv2d_resample.py:114-131), fp32.
//     dW[o][i][ky][kx] = sum_{y, x} in[i][y][x] * dY[o][2 y + ky][2 x + kx]                    (in: H x W, dY: (2 H + 1) x (2 W + 1))
// wgrad_kernel runs this as four parity-class GEMMs whose dY operand is read with stride 2 (half of every sector wasted, `in` read four times):
// 0.50 - 0.57 of the fp32 peak, the single most expensive conv launch of the loop.  Here one block owns 64 output x 64 input channels x ALL NINE
// taps (144 accumulators per lane, VGPR-form MFMAs, two blocks per CU) and walks input rows of a 16-pixel column strip:
//   * per step the three dY rows 2 y .. 2 y + 2 (33 columns, contiguous 132-byte runs) and the input row piece arrive through plain buffer loads --
//     per-thread constant vector offsets, the row in the scalar offset: no address arithmetic -- one step ahead of the MFMAs;
//   * on their way into LDS the dY rows are DE-INTERLEAVED into the operand streams of the three kx taps: even columns E[j] = dY[2 j], odd columns
//     O[j] = dY[2 j + 1]; tap kx = 0 contracts in[x] with E[x], kx = 1 with O[x], kx = 2 with E[x + 1].  An MFMA's two K slots are the lane halves, so
//     each stream is stored per half: Eh0 / Eh1 / Oh0 / Oh1 (x = 2 s + h) and the shifted Eh0s[s] = E[2 s + 2] -- five arrays per row, every A fragment
//     one 16-byte LDS read of four K steps, no shuffles (the fp16 kernel hwgrad_kernel needs v_alignbit for the same shift);
//   * 72 MFMAs per step and wave (9 taps x 8 K steps), three accumulators interleaved.
// The pixel reduction is split over blocks (column strips x row chunks x samples); partial sums meet in the zeroed dW through fp32 atomics like the
// other weight-gradient kernels.  A masked gradient (dy_seg_flags) skips the steps whose dY rows are flagged zero.
// ------------------------------------------------------------------------------------------------
struct TWgradParams {
    int N, Mo, Ci, H, W, OW;
    int64_t in_bs, out_bs, wbs;
    int wsm, wsc, widx9[9];
    int xsegs, rows_per_chunk, ncib;
    const int32_t* seg_flags; int nseg;
};
constexpr int TW_PX = 16;                       // input pixels (K) per step
constexpr int TW_ARR = 64 * 8 + 8 * 4;          // floats of one operand array [64 channels][8 K steps], 16 bytes of padding per 8 channels (conflict-free 16-byte reads)
constexpr int TW_DY = 15 * TW_ARR;              // one dY stage: [row 3][stream 5]
constexpr int TW_X = 2 * TW_ARR;                // one input stage: [half 2]
constexpr int TW_STAGE = TW_DY + TW_X;          // 9248 floats
constexpr int TW_MAXROWS = 256;

__global__ void __launch_bounds__(256, 2) twgrad_kernel(TWgradParams P, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw) {
    __shared__ __attribute__((aligned(16))) float lds[2 * TW_STAGE];
    __shared__ unsigned char live[TW_MAXROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int cw = wave >> 1, iw = wave & 1;                          // MFMA quadrant: 32 output x 32 input channels
    const int n = blockIdx.z;
    const int cob = (blockIdx.y / P.ncib) * 64, cib = (blockIdx.y % P.ncib) * 64;
    const int xseg = blockIdx.x % P.xsegs, rchunk = blockIdx.x / P.xsegs;
    const int ix0 = xseg * TW_PX;
    const int r0 = rchunk * P.rows_per_chunk, r1 = min(r0 + P.rows_per_chunk, P.H);
    const int nrows = r1 - r0;
    if (nrows <= 0) return;
    const int OHW = (2 * P.H + 1) * P.OW;
    // ---- which steps carry a gradient (masked launches): the three dY rows of a step over the strip's 33 columns
    if (P.seg_flags) {
        const int32_t* fl = P.seg_flags + (int64_t)n * P.nseg;
        for (int j = tid; j < nrows; j += 256) {
            int any = 0;
            for (int r = 0; r < 3; ++r) {
                const int base = (2 * (r0 + j) + r) * P.OW + 2 * ix0;
                for (int sg = base >> 4; sg <= (base + 32) >> 4; ++sg) any |= fl[sg];
            }
            live[j] = any != 0;
        }
    } else {
        for (int j = tid; j < nrows; j += 256) live[j] = 1;
    }
    __syncthreads();
    auto next_live = [&](int j) { while (j < nrows && !live[j]) ++j; return j; };     // (block-uniform)
    int cur = next_live(0);
    if (cur >= nrows) return;

    // ---- loaders: per-thread constant vector offsets, the step / row in the scalar offset
    const __amdgpu_buffer_rsrc_t rsY = make_rsrc(dy + (int64_t)n * P.out_bs + (int64_t)cob * OHW, (int64_t)64 * OHW * 4);
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(x + (int64_t)n * P.in_bs + (int64_t)cib * P.H * P.W, (int64_t)64 * P.H * P.W * 4);
    const int w8 = tid >> 5, e = tid & 31;                           // dY loads: column e of 8 channel rows per pass, 24 passes = [row 3][8 channel groups]
    const int voffY = (w8 * OHW + e) * 4;
    // destinations of column e: even e = E[j], j = e / 2 -> stream j & 1 (Eh0 / Eh1) at s = j >> 1, and for even j >= 2 also Eh0s at s = j / 2 - 1;
    //                           odd e = O[j], j = e / 2 -> stream 2 + (j & 1) at s = j >> 1
    const int jj = e >> 1;
    const int arr0 = (e & 1) ? 2 + (jj & 1) : (jj & 1), s0 = jj >> 1;
    const bool two = !(e & 1) && !(jj & 1) && jj >= 2;
    const int ldsY0 = arr0 * TW_ARR + w8 * 8 + s0;                    // + (row * 5) * TW_ARR + (8 g) * 8 + g * 4 for channel group g = i & 7 (co = 8 g + w8)
    const int ldsY1 = 4 * TW_ARR + w8 * 8 + (jj >> 1) - 1;
    // the 33rd column E[16] of every (row, channel): threads 0..191, stream Eh0s at s = 7
    const int xr = tid >> 6, xc = tid & 63;                          // (tid < 192)
    const int voffE = (xc * OHW + 32) * 4;
    const int ldsE = (xr * 5 + 4) * TW_ARR + xc * 8 + (xc >> 3) * 4 + 7;
    // input row piece: 64 channels x 16 pixels = 256 float4
    const int xci = tid >> 2, xp4 = (tid & 3) * 4;
    const int voffX = (xci * P.H * P.W + xp4) * 4;
    const int ldsX = TW_DY + xci * 8 + (xci >> 3) * 4 + (xp4 >> 1);

    float ry[24], re = 0.f;
    f32x4_t rx;
    auto load_step = [&](int j) {
        const int iy = r0 + j;
        const int sbase = ((2 * iy) * P.OW + 2 * ix0) * 4;
#pragma unroll
        for (int i = 0; i < 24; ++i)
            ry[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsY, voffY, sbase + ((8 * (i & 7)) * OHW + (i >> 3) * P.OW) * 4, 0));
        if (tid < 192) re = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsY, voffE, sbase + xr * P.OW * 4, 0));
        rx = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const char*>(x + (int64_t)n * P.in_bs + (int64_t)cib * P.H * P.W) + voffX + (iy * P.W + ix0) * 4);
    };
    // one dY pass (8 channel rows) of the staged registers -> LDS; the passes are issued one per MFMA behind the last tap row of a step
    auto store_pass = [&](float* st, int i) __attribute__((always_inline)) {
        const int off = ((i >> 3) * 5) * TW_ARR + (8 * (i & 7)) * 8 + (i & 7) * 4;
        st[ldsY0 + off] = ry[i];
        if (two) st[ldsY1 + off] = ry[i];
    };
    auto store_tail = [&](float* st) __attribute__((always_inline)) {
        if (tid < 192) st[ldsE] = re;
        *reinterpret_cast<float2*>(st + ldsX) = make_float2(rx.x, rx.z);                 // half 0: pixels xp4, xp4 + 2
        *reinterpret_cast<float2*>(st + ldsX + TW_ARR) = make_float2(rx.y, rx.w);        // half 1: pixels xp4 + 1, xp4 + 3
    };
    auto store_step = [&](float* st) {
#pragma unroll
        for (int i = 0; i < 24; ++i) store_pass(st, i);
        store_tail(st);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // operand fragment bases (floats): A = dY stream of tap kx for this lane's half, B = the input row
    const int co_l = cw * 32 + l32, ci_l = iw * 32 + l32;
    const int aoff = co_l * 8 + (co_l >> 3) * 4;
    const int a_kx[3] = {(h ? 1 : 0) * TW_ARR + aoff, (h ? 3 : 2) * TW_ARR + aoff, (h ? 4 : 1) * TW_ARR + aoff};
    const int boff = TW_DY + h * TW_ARR + ci_l * 8 + (ci_l >> 3) * 4;

    load_step(cur);
    store_step(lds);
    __syncthreads();
    int buf = 0;
    while (true) {
        const int nxt = next_live(cur + 1);
        const bool has_next = nxt < nrows;
        if (has_next) load_step(nxt);
        const float* st = lds + buf * TW_STAGE;
        const float4 b0 = *reinterpret_cast<const float4*>(st + boff), b1 = *reinterpret_cast<const float4*>(st + boff + 4);
        const float bk[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float ak[3][8];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 a0 = *reinterpret_cast<const float4*>(st + ky * 5 * TW_ARR + a_kx[kx]), a1 = *reinterpret_cast<const float4*>(st + ky * 5 * TW_ARR + a_kx[kx] + 4);
                ak[kx][0] = a0.x; ak[kx][1] = a0.y; ak[kx][2] = a0.z; ak[kx][3] = a0.w; ak[kx][4] = a1.x; ak[kx][5] = a1.y; ak[kx][6] = a1.z; ak[kx][7] = a1.w;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[kx][k], bk[k], acc[ky * 3 + kx], 0, 0, 0);
                    // the next step's rows (requested at the top of this step, ~48 MFMAs ago) go to the other stage behind the last tap row's MFMAs, one
                    // pass per MFMA: the LDS writes run in the matrix pipe's shadow instead of after it
                    if (ky == 2 && has_next) {
                        __builtin_amdgcn_sched_barrier(0);
                        store_pass(lds + (buf ^ 1) * TW_STAGE, k * 3 + kx);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
        if (!has_next) break;
        store_tail(lds + (buf ^ 1) * TW_STAGE);
        __syncthreads();
        buf ^= 1; cur = nxt;
    }
    // ---- partial sums -> dW (zeroed by the caller side): C/D layout col = lane & 31 (input channel), row = (r & 3) + 8 (r >> 2) + 4 h (output channel)
    float* dwn = dw + (P.wbs ? (int64_t)n * P.wbs : 0) + (int64_t)(cib + ci_l) * P.wsc;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = cob + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            atomicAdd(dwn + (int64_t)m * P.wsm + P.widx9[t], acc[t][r]);
        }
}

// eligibility + launch of twgrad_kernel (P = make_forward(d) of a transposed conv); returns false when the problem keeps the generic kernel
static bool twgrad_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SPI_CONV_TWGRAD"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}
static bool launch_twgrad(const spi_conv_desc* d, const IGemmParams& P, const float* x, const float* dy, float* dw, hipStream_t st) {
    if (!twgrad_enabled() || !d->transposed || d->kh != 3 || d->kw != 3 || d->compute_f16 != 0 || d->act_dtype != 0) return false;
    if (P.Mo % 64 || P.Ci % 64 || P.IW % TW_PX || P.OH != 2 * P.IH + 1 || P.OW != 2 * P.IW + 1 || P.ncls != 4) return false;
    if ((int64_t)64 * P.OH * P.OW * 4 >= (1ll << 31) || (int64_t)64 * P.IH * P.IW * 4 >= (1ll << 31)) return false;
    if ((int64_t)P.IH * P.IW < 16384) return false;          // small planes: the blocks' 36 864 atomics each outweigh the MFMAs, the generic split keeps them
    TWgradParams T;
    T.N = P.N; T.Mo = P.Mo; T.Ci = P.Ci; T.H = P.IH; T.W = P.IW; T.OW = P.OW;
    T.in_bs = P.in_bs; T.out_bs = P.out_bs; T.wbs = P.wbs; T.wsm = P.wsm; T.wsc = P.wsc;
    for (int t = 0; t < 9; ++t) T.widx9[t] = -1;
    for (int c = 0; c < P.ncls; ++c) {
        const ClassParams& C = P.cls[c];
        for (int t = 0; t < C.taps.T; ++t) {
            const int ky = C.ooy - 2 * C.taps.dy[t], kx = C.oox - 2 * C.taps.dx[t];      // make_forward: dy = -(ky - py) / 2
            if (ky < 0 || ky > 2 || kx < 0 || kx > 2) return false;
            T.widx9[ky * 3 + kx] = C.taps.widx[t];
        }
    }
    for (int t = 0; t < 9; ++t) if (T.widx9[t] < 0) return false;
    T.xsegs = P.IW / TW_PX; T.ncib = P.Ci / 64;
    const int64_t base = (int64_t)T.xsegs * (P.Mo / 64) * T.ncib * P.N;
    // ~512 blocks (two per CU, one round: measured best of 256 .. 2048 on 256 -> 128 at 256^2), at least 8 rows per block (below that the 36 864
    // atomics of a block cost more than its MFMAs)
    static int target = 0;
    if (!target) { const char* e = getenv("SPI_TWGRAD_BLOCKS"); target = e ? atoi(e) : 512; if (target < 64) target = 512; }
    int chunks = (int)std::max<int64_t>(1, std::min<int64_t>((target + base - 1) / base, P.IH / 8));
    T.rows_per_chunk = std::min((P.IH + chunks - 1) / chunks, TW_MAXROWS);
    chunks = (P.IH + T.rows_per_chunk - 1) / T.rows_per_chunk;
    T.seg_flags = d->dy_seg_flags; T.nseg = (int)(((int64_t)P.OH * P.OW + SPI_SEG_PIXELS - 1) / SPI_SEG_PIXELS);
    dim3 grid((unsigned)(T.xsegs * chunks), (unsigned)((P.Mo / 64) * T.ncib), (unsigned)P.N);
    hipLaunchKernelGGL(twgrad_kernel, grid, dim3(256), 0, st, T, x, dy, dw);
    return true;
}

// tile configuration and number of K ranges of a forward / dgrad implicit GEMM (host logic shared by the launch and spi_conv2d_out_accumulates)
struct IGemmPlan { int cfg, nsplit; };
static IGemmPlan plan_igemm(const IGemmParams& P, int f16) {
    int maxpix = 0, maxT = 0;
    for (int c = 0; c < P.ncls; ++c) { maxpix = std::max(maxpix, P.cls[c].OHp * P.cls[c].OWp); maxT = std::max(maxT, P.cls[c].taps.T); }
    auto blocks = [&](int bm, int bn) { return (int64_t)((maxpix + bn - 1) / bn) * ((P.Mo + bm - 1) / bm) * P.N * P.ncls; };
    const int nslab = maxT * ((P.Ci + BK - 1) / BK);
    int cfg;                    // 0: 32x128, 1: 128x128, 2: 64x64, 3: 32x32, 4: 64x256 (64 output channels: same 64x64 wave tile as cfg 1)
    if (P.Mo <= 32) cfg = 0;
    else if (P.Mo <= 64 && blocks(64, 256) >= 384) cfg = 4;
    else if (P.Mo > 32 && P.Mo <= 64 && maxpix >= 4096) cfg = 2;       // 64 output channels, too few 64x256 tiles: 64x64 tiles waste nothing (a 128-row tile is half empty)
    else if (blocks(128, 128) >= 384) cfg = 1;
    else if (P.Mo >= 64 && maxpix >= 64) cfg = 2;
    else cfg = 3;
    // 2-piece split-bf16: the matrix pipe is 5.3x faster, so the per-slab staging work (splitting, LDS stores) weighs more: a 128 x 256
    // tile halves the A-operand work per MFMA (measured: +10..20 % at 256 channels; the 3-piece mode loses a third with it -- registers)
    if (f16 == 2 && cfg == 1 && blocks(128, 256) >= 512) cfg = 5;
    // fp16 operands: the matrix pipe is 16x faster than fp32's, the kernel is bound by moving fp32 operands from L2 into registers (64 B / clk / CU:
    // a 128 x 128 tile loads 16 KB per K = 16 slab for 128 matrix-pipe cycles per wave) -- the 128 x 256 tile loads 1.5x the bytes for 2x the MFMAs
    if (f16 == 1 && cfg == 1 && blocks(128, 256) >= 512 && (P.wsm == 1 || P.wsc == 1)) cfg = 5;
    static const int bm_of[6] = {32, 128, 64, 32, 64, 128}, bn_of[6] = {128, 128, 64, 32, 256, 256};
    const int64_t nb = blocks(bm_of[cfg], bn_of[cfg]);
    int nsplit = 1;
    const bool f16_ok = f16 == 1 && (P.Ci % BK == 0) && (P.in_bs * 4 < (1ll << 31)) && (P.w_elems * 4 < (1ll << 31));
    // fp16 operands: never split (one precision per call).  Split-bf16 requests on grids that need split-K run the exact fp32 kernels
    // (the 4^2..32^2 layers: a higher precision than asked for, on a small share of the FLOPs).
    // (~1024 blocks, >= 4 slabs per range: targets of 512 / 256 blocks and >= 8 slabs were measured within +-10 % on the 4^2..32^2 layers, round 3)
    if (nb < 512 && nslab >= 8 && !f16_ok) nsplit = (int)std::min<int64_t>(nslab / 4, (1024 + nb - 1) / nb);
    return IGemmPlan{cfg, nsplit};
}

static int dispatch_igemm(const IGemmParams& P, const float* in, const float* w, float* out, const Epilogue& ep, hipStream_t st, int f16 = 0,
                          bool out_zeroed = false, bool ioh = false) {
    const IGemmPlan plan = plan_igemm(P, f16);
    const int cfg = plan.cfg, nsplit = plan.nsplit;
    if (ioh) {
        // fp16 activation tensors exist on the buffer-descriptor path of the fp16 operand mode only (whole 16-channel slabs, < 2 GiB per sample)
        // (the same size predicate as plan_igemm's `f16_ok` and launch_igemm's `buf`: a sample that passes here takes the buffer-descriptor kernel)
        SPI_REQUIRE(f16 == 1 && nsplit == 1 && (P.Ci % BK == 0) && (P.in_bs * 4 < (1ll << 31)) && (P.w_elems * 4 < (1ll << 31)),
                    "spi_conv2d: act_dtype = fp16 needs compute_f16 = 1 and a reduction channel count that is a multiple of 16 (got %d)", P.Ci);
    }
    if (nsplit > 1 && !out_zeroed) {
        spi_zero_async(out, (int64_t)P.N * P.out_bs, st);
    }
    switch (cfg) {
    case 0: launch_igemm<1, 4, 1, 1>(P, in, w, out, ep, nsplit, st, f16, ioh); break;
    case 1: launch_igemm<2, 2, 2, 2>(P, in, w, out, ep, nsplit, st, f16, ioh); break;
    case 2: launch_igemm<2, 2, 1, 1>(P, in, w, out, ep, nsplit, st, f16, ioh); break;
    case 4: launch_igemm<1, 4, 2, 2>(P, in, w, out, ep, nsplit, st, f16, ioh); break;
    case 5: launch_igemm<2, 2, 2, 4>(P, in, w, out, ep, nsplit, st, f16, ioh); break;
    default: launch_igemm<1, 1, 1, 1>(P, in, w, out, ep, nsplit, st, f16, ioh); break;
    }
    if (nsplit > 1 && (ep.bias || ep.noise || ep.act)) {
        const int64_t total = (int64_t)P.N * P.out_bs;
        const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(conv_epilogue_kernel, dim3(grid), dim3(256), 0, st, out, total, P.Mo, (int64_t)P.OH * P.OW, ep);
    }
    return SPI_OK;
}

// Winograd eligibility of a forward / dgrad problem: 3x3, stride 1, pad 1, exact fp32, whole 8-channel slabs, and enough 16 x 16 x 64
// blocks to fill the 256 CUs once (smaller layers keep the implicit GEMM, whose split-K fills the chip).
// F(4x4, 3x3) switch: on unless SPI_CONV_WINO_F4=0 (A/B measurements, the precision table of DESIGN.md) or spi_conv_wino_f4_set(0)
static int g_wino_f4 = -1;
static bool wino_f4_enabled() {
    if (g_wino_f4 < 0) { const char* e = getenv("SPI_CONV_WINO_F4"); g_wino_f4 = (e && e[0] == '0') ? 0 : 1; }
    return g_wino_f4 != 0;
}
extern "C" void spi_conv_wino_f4_set(int on) { g_wino_f4 = on ? 1 : 0; }

static bool make_wino(const spi_conv_desc* d, const IGemmParams& P, WinoParams& Wp) {
    Wp.f4 = 0;
    // (compute_f16 = 3, the 6-product bf16 split, asks for fp32-equivalent products: the fp32 Winograd path is at least that precise and faster
    //  than the split implicit GEMM on these layers, so it serves that mode too; modes 1 and 2 trade precision for speed and keep their kernels)
    if (d->transposed || d->kh != 3 || d->pad != 1 || (d->compute_f16 != 0 && d->compute_f16 != 3) || P.ncls != 1 || P.cls[0].taps.T != 9) return false;
    if (P.Ci % 8 != 0 || P.in_bs * 4 >= (1ll << 31) || P.IH != P.OH || P.IW != P.OW) return false;
    Wp.N = P.N; Wp.nw = d->w_batch_stride ? P.N : 1; Wp.Mo = P.Mo; Wp.Ci = P.Ci; Wp.H = P.OH; Wp.W = P.OW;
    Wp.bx = (P.OW + 15) / 16; Wp.by = (P.OH + 15) / 16; Wp.ocp = (P.Mo + 63) / 64 * 64;
    Wp.in_bs = P.in_bs; Wp.out_bs = P.out_bs; Wp.wbs = P.wbs; Wp.u_bs = 0; Wp.wsm = P.wsm; Wp.wsc = P.wsc;
    const TapSet& T = P.cls[0].taps;
    for (int t = 0; t < 9; ++t) {
        if (T.dy[t] < -1 || T.dy[t] > 1 || T.dx[t] < -1 || T.dx[t] > 1) return false;
        Wp.widx[(T.dy[t] + 1) * 3 + (T.dx[t] + 1)] = T.widx[t];
    }
    Wp.seg_flags = P.seg_flags; Wp.out_flags = P.out_flags; Wp.nseg = P.seg_flags ? P.nseg : P.out_nseg;
    const int64_t blocks = (int64_t)Wp.bx * Wp.by * (Wp.ocp / 64) * P.N;
    // 128 .. 255 blocks (512 channels at 64^2 / N = 1, the 512-channel VGG layers at 32^2 / N = 4, 256 channels at 64^2 / N = 2): half the CUs
    // would idle -- cut the channel reduction in two ranges of >= 16 slabs; the partial outputs meet through atomics and the epilogue runs as its
    // own small kernel (0.17 -> 0.13 ms on the first).  Finer splits of smaller layers were measured SLOWER than the split-K implicit GEMM
    // (32 .. 64 blocks x 4 .. 8 ranges: the zero / weight-transform / epilogue launches and the atomics outweigh the 2.25x fewer MFMAs).
    // (not with a needed-output map: the split's separate epilogue pass would write act(bias + noise) into the tiles the kernel skipped, where
    //  the unsplit path leaves exact zeros -- the contract of out_seg_flags must not depend on the block count)
    Wp.ksplit = (blocks >= 128 && blocks < 256 && P.Ci / 8 >= 32 && !P.out_flags) ? 2 : 1;
    // Round 6: F(4x4, 3x3) where its 16 x 32-pixel x 64-channel blocks still fill the 256 CUs (the >= 256^2 layers): 1.78x fewer MFMAs than F(2x2, 3x3)
    // (>= 128 reduction channels at >= 256^2, >= 256 at 128^2: the generator's b128 (batched) / b256 / super-resolution layers.  The VGG layers of the LPIPS / BoxCX losses -- 64 channels at
    //  256^2, 128 at 128^2 -- keep F(2x2): their ReLU / max-pool decisions amplify rounding noise into isolated gradient elements, tests/test_hip_losses_gpu.py)
    Wp.f4 = (wino_f4_enabled() && P.OH % 16 == 0 && P.OW % 32 == 0 && ((P.Ci >= 128 && (int64_t)P.OH * P.OW >= 65536) || (P.Ci >= 256 && (int64_t)P.OH * P.OW >= 16384)) &&
             (int64_t)(P.OH / 16) * (P.OW / 32) * (Wp.ocp / 64) * P.N >= 256 && Wp.ksplit == 1) ? 1 : 0;
    return blocks >= 128 && P.Mo >= 48;
}

// channel-split Winograd: zero the output first, run the shared epilogue kernel afterwards
static int wino_conv(WinoParams& Wp, const IGemmParams& P, const float* in, const float* w, float* out, const Epilogue& ep, void* ws, hipStream_t st,
                     bool out_zeroed, bool u_ready = false) {
    if (Wp.ksplit > 1 && !out_zeroed) {
        int rc = spi_zero_async(out, (int64_t)P.N * P.out_bs, st); if (rc) return rc;
    }
    int rc = spi_wino_launch(Wp, in, w, out, ep, ws, st, u_ready); if (rc) return rc;
    if (Wp.ksplit > 1 && (ep.bias || ep.noise || ep.act)) {
        const int64_t total = (int64_t)P.N * P.out_bs;
        const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(conv_epilogue_kernel, dim3(grid), dim3(256), 0, st, out, total, P.Mo, (int64_t)P.OH * P.OW, ep);
    }
    return SPI_OK;
}

// Winograd F(3x3, 2x2) eligibility of a weight-gradient problem (P = make_forward(d)): 3x3, stride 1, pad 1, exact fp32, rows of at least one
// 32-pixel strip, and enough tile rows per block for the 64 x 64 x 16 accumulators' prologue / 144-atomic epilogue to amortise.  A masked
// gradient (dy_seg_flags) is handled inside the kernel: tile rows without a flagged segment are skipped.
static bool make_wino_wgrad(const spi_conv_desc* d, const IGemmParams& P, WinoParams& Wp) {
    Wp.f4 = 0;
    if (d->transposed || d->kh != 3 || d->pad != 1 || (d->compute_f16 != 0 && d->compute_f16 != 3) || P.ncls != 1 || P.cls[0].taps.T != 9) return false;
    if (P.IH != P.OH || P.IW != P.OW || P.OW < 32 || P.OH < 16) return false;
    if ((int64_t)P.OH * P.OW * 64 * 4 >= (1ll << 31) || P.Mo < 32 || P.Ci < 32) return false;
    Wp.N = P.N; Wp.nw = d->w_batch_stride ? P.N : 1; Wp.Mo = P.Mo; Wp.Ci = P.Ci; Wp.H = P.OH; Wp.W = P.OW;
    Wp.bx = Wp.by = 0; Wp.ocp = 0;
    Wp.in_bs = P.in_bs; Wp.out_bs = P.out_bs; Wp.wbs = P.wbs; Wp.u_bs = 0; Wp.wsm = P.wsm; Wp.wsc = P.wsc;
    const TapSet& T = P.cls[0].taps;
    for (int t = 0; t < 9; ++t) {
        if (T.dy[t] < -1 || T.dy[t] > 1 || T.dx[t] < -1 || T.dx[t] > 1) return false;
        Wp.widx[(T.dy[t] + 1) * 3 + (T.dx[t] + 1)] = T.widx[t];
    }
    Wp.ksplit = 1;
    Wp.seg_flags = d->dy_seg_flags; Wp.out_flags = nullptr;
    Wp.nseg = d->dy_seg_flags ? (int)(((int64_t)P.OH * P.OW + SPI_SEG_PIXELS - 1) / SPI_SEG_PIXELS) : 0;
    return true;
}
// Direct fp16 conv (hconv.hip) eligibility of a forward / dgrad problem: fp16 activation tensors, fp16 operands, 3x3, stride 1, pad 1
static bool make_hconv(const spi_conv_desc* d, const IGemmParams& P, WinoParams& Wp) {
    if (d->act_dtype != SPI_DTYPE_F16 || d->compute_f16 != 1 || d->transposed || d->kh != 3 || d->pad != 1 || P.ncls != 1 || P.cls[0].taps.T != 9) return false;
    if (P.IH != P.OH || P.IW != P.OW) return false;
    Wp.N = P.N; Wp.nw = d->w_batch_stride ? P.N : 1; Wp.Mo = P.Mo; Wp.Ci = P.Ci; Wp.H = P.OH; Wp.W = P.OW;
    Wp.bx = Wp.by = 0; Wp.ocp = 0;
    Wp.in_bs = P.in_bs; Wp.out_bs = P.out_bs; Wp.wbs = P.wbs; Wp.u_bs = 0; Wp.wsm = P.wsm; Wp.wsc = P.wsc;
    const TapSet& T = P.cls[0].taps;
    for (int t = 0; t < 9; ++t) {
        if (T.dy[t] < -1 || T.dy[t] > 1 || T.dx[t] < -1 || T.dx[t] > 1) return false;
        Wp.widx[(T.dy[t] + 1) * 3 + (T.dx[t] + 1)] = T.widx[t];
    }
    Wp.seg_flags = P.seg_flags; Wp.out_flags = P.out_flags; Wp.nseg = P.seg_flags ? P.nseg : P.out_nseg;
    Wp.ksplit = 1;
    return spi_hconv_eligible(Wp);
}
// Direct fp16 stride-2 conv (hconv.hip) eligibility of the DATA GRADIENT of a stride-2 transposed 3x3 conv (P = make_dgrad(d))
static bool make_hconv_s2(const spi_conv_desc* d, const IGemmParams& P, WinoParams& Wp) {
    if (d->act_dtype != SPI_DTYPE_F16 || d->compute_f16 != 1 || !d->transposed || d->kh != 3 || P.ncls != 1 || P.cls[0].taps.T != 9 || P.isy != 2 || P.isx != 2) return false;
    Wp.N = P.N; Wp.nw = d->w_batch_stride ? P.N : 1; Wp.Mo = P.Mo; Wp.Ci = P.Ci; Wp.H = P.OH; Wp.W = P.OW;
    Wp.bx = Wp.by = 0; Wp.ocp = 0;
    Wp.in_bs = P.in_bs; Wp.out_bs = P.out_bs; Wp.wbs = P.wbs; Wp.u_bs = 0; Wp.wsm = P.wsm; Wp.wsc = P.wsc;
    const TapSet& T = P.cls[0].taps;
    for (int t = 0; t < 9; ++t) {
        if (T.dy[t] < 0 || T.dy[t] > 2 || T.dx[t] < 0 || T.dx[t] > 2) return false;
        Wp.widx[T.dy[t] * 3 + T.dx[t]] = T.widx[t];
    }
    Wp.seg_flags = P.seg_flags; Wp.out_flags = nullptr; Wp.nseg = P.nseg; Wp.ksplit = 1;
    return spi_hconv_s2_eligible(Wp);
}
// Direct fp16 forward of a stride-2 transposed 3x3 conv (hconv.hip; P = make_forward(d), whose four parity classes the kernel re-derives itself)
static bool make_hconv_t2(const spi_conv_desc* d, const IGemmParams& P, WinoParams& Wp) {
    if (d->act_dtype != SPI_DTYPE_F16 || d->compute_f16 != 1 || !d->transposed || d->kh != 3 || d->kw != 3) return false;
    Wp.N = P.N; Wp.nw = d->w_batch_stride ? P.N : 1; Wp.Mo = P.Mo; Wp.Ci = P.Ci; Wp.H = d->H; Wp.W = d->W;
    Wp.bx = Wp.by = 0; Wp.ocp = 0;
    Wp.in_bs = P.in_bs; Wp.out_bs = P.out_bs; Wp.wbs = P.wbs; Wp.u_bs = 0; Wp.wsm = P.wsm; Wp.wsc = P.wsc;
    for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) Wp.widx[ky * 3 + kx] = tap_w(d, ky, kx);
    Wp.seg_flags = nullptr; Wp.out_flags = P.out_flags; Wp.nseg = P.out_nseg; Wp.ksplit = 1;
    return spi_hconv_t2_eligible(Wp);
}
// Direct fp16 weight gradient (hconv.hip) eligibility (P = make_forward(d)): as make_hconv, plus whole 4 x 32-pixel tiles and a dense gradient
static bool make_hwgrad(const spi_conv_desc* d, const IGemmParams& P, WinoParams& Wp) {
    if (d->act_dtype != SPI_DTYPE_F16 || d->compute_f16 != 1 || d->transposed || d->kh != 3 || d->pad != 1 || P.ncls != 1 || P.cls[0].taps.T != 9) return false;
    if (P.IH != P.OH || P.IW != P.OW) return false;
    Wp.N = P.N; Wp.nw = d->w_batch_stride ? P.N : 1; Wp.Mo = P.Mo; Wp.Ci = P.Ci; Wp.H = P.OH; Wp.W = P.OW;
    Wp.bx = Wp.by = 0; Wp.ocp = 0;
    Wp.in_bs = P.in_bs; Wp.out_bs = P.out_bs; Wp.wbs = P.wbs; Wp.u_bs = 0; Wp.wsm = P.wsm; Wp.wsc = P.wsc;
    const TapSet& T = P.cls[0].taps;
    for (int t = 0; t < 9; ++t) {
        if (T.dy[t] < -1 || T.dy[t] > 1 || T.dx[t] < -1 || T.dx[t] > 1) return false;
        Wp.widx[(T.dy[t] + 1) * 3 + (T.dx[t] + 1)] = T.widx[t];
    }
    Wp.seg_flags = d->dy_seg_flags; Wp.out_flags = nullptr; Wp.nseg = 0; Wp.ksplit = 1;
    return spi_hwgrad_eligible(Wp);
}
constexpr int64_t WINO_WGRAD_WS = 16;      // the pass needs no scratch; a (nominal) workspace is the caller's opt-in, as for the other passes

extern "C" {

int64_t spi_conv2d_workspace_bytes(const spi_conv_desc* d, int pass) {
    if (validate(d, "spi_conv2d_workspace_bytes") || pass < 0 || pass > 2) return 0;
    IGemmParams P; WinoParams Wp;
    if (pass == 2) {
        make_forward(d, P);
        if (make_hwgrad(d, P, Wp)) return spi_hwgrad_workspace_bytes(Wp);         // (any smaller non-null workspace still opts in: atomics instead of partial sums)
        return make_wino_wgrad(d, P, Wp) ? WINO_WGRAD_WS : 0;
    }
    if (pass == 0) make_forward(d, P); else make_dgrad(d, P);
    if (make_hconv(d, P, Wp)) return spi_hconv_workspace_bytes(Wp);
    if (pass == 1 && make_hconv_s2(d, P, Wp)) return spi_hconv_workspace_bytes(Wp);
    if (pass == 0 && make_hconv_t2(d, P, Wp)) return spi_hconv_workspace_bytes(Wp);
    return make_wino(d, P, Wp) ? spi_wino_workspace_bytes(Wp) : 0;
}

int spi_conv2d_out_accumulates(const spi_conv_desc* d, int pass) {
    if (validate(d, "spi_conv2d_out_accumulates")) return SPI_ERR_BAD_ARG;
    if (pass < 0 || pass > 1) { spi_set_error("spi_conv2d_out_accumulates: pass must be 0 (forward) or 1 (dgrad)"); return SPI_ERR_BAD_ARG; }
    IGemmParams P; WinoParams Wp;
    if (pass == 0) {
        make_forward(d, P);
        if (d->out_seg_flags) { P.out_flags = d->out_seg_flags; P.out_nseg = (int)(((int64_t)P.OH * P.OW + SPI_SEG_PIXELS - 1) / SPI_SEG_PIXELS); }
    } else {
        make_dgrad(d, P);
    }
    if (d->workspace && make_hconv(d, P, Wp) && d->workspace_bytes >= spi_hconv_workspace_bytes(Wp)) return 0;
    if (d->workspace && pass == 1 && make_hconv_s2(d, P, Wp) && d->workspace_bytes >= spi_hconv_workspace_bytes(Wp)) return 0;
    if (d->workspace && pass == 0 && make_hconv_t2(d, P, Wp) && d->workspace_bytes >= spi_hconv_workspace_bytes(Wp)) return 0;
    if (d->workspace && make_wino(d, P, Wp) && d->workspace_bytes >= spi_wino_workspace_bytes(Wp)) return Wp.ksplit > 1 ? 1 : 0;
    return plan_igemm(P, d->compute_f16).nsplit > 1 ? 1 : 0;
}

int spi_conv2d_plan(const spi_conv_desc* d, int pass, int32_t* out8) {
    int rc = validate(d, "spi_conv2d_plan"); if (rc) return rc;
    SPI_REQUIRE(out8 && pass >= 0 && pass <= 2, "spi_conv2d_plan: pass must be 0, 1 or 2 and out8 non-null");
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    IGemmParams P; WinoParams Wp;
    if (pass == 2) {
        make_forward(d, P);
        if (d->workspace && d->workspace_bytes >= WINO_WGRAD_WS && make_hwgrad(d, P, Wp)) {
            out8[0] = 2; out8[1] = 128; out8[2] = 64; out8[3] = 1; out8[4] = -1; out8[5] = 512;        // (grid: see spi_hwgrad_launch)
            return SPI_OK;
        }
        if (d->workspace && d->workspace_bytes >= WINO_WGRAD_WS && make_wino_wgrad(d, P, Wp)) {
            out8[0] = 1; out8[1] = 64; out8[2] = 64; out8[3] = 1; out8[4] = -1; out8[5] = 256;        // (grid: see spi_wino_wgrad_launch)
            return SPI_OK;
        }
        const int BN = 128, BM = P.Mo <= 32 ? 32 : 128;
        int maxpix = 0, maxcols = 0;
        for (int c = 0; c < P.ncls; ++c) { maxpix = std::max(maxpix, P.cls[c].OHp * P.cls[c].OWp); maxcols = std::max(maxcols, P.Ci * P.cls[c].taps.T); }
        const int tiles = ((P.Mo + BM - 1) / BM) * ((maxcols + BN - 1) / BN);
        int64_t active = 0;
        for (int c = 0; c < P.ncls; ++c) active += (int64_t)((P.Mo + BM - 1) / BM) * ((P.Ci * P.cls[c].taps.T + BN - 1) / BN);
        const int64_t splits = std::max<int64_t>(1, 1024 / std::max<int64_t>(1, active * P.N));
        int ppb = (int)((maxpix + splits - 1) / splits);
        ppb = std::max(256, ((ppb + BK - 1) / BK) * BK);
        const int nrange = (maxpix + ppb - 1) / ppb;
        out8[0] = 0; out8[1] = BM; out8[2] = BN; out8[3] = nrange; out8[4] = (int32_t)((int64_t)nrange * tiles * P.N * P.ncls); out8[5] = 256;
        return SPI_OK;
    }
    if (pass == 0) {
        make_forward(d, P);
        if (d->out_seg_flags) { P.out_flags = d->out_seg_flags; P.out_nseg = (int)(((int64_t)P.OH * P.OW + SPI_SEG_PIXELS - 1) / SPI_SEG_PIXELS); }
    } else make_dgrad(d, P);
    if (d->workspace && pass == 0 && make_hconv_t2(d, P, Wp) && d->workspace_bytes >= spi_hconv_workspace_bytes(Wp)) {
        out8[0] = 2; out8[1] = 128; out8[2] = 512; out8[3] = 1; out8[4] = (int32_t)((int64_t)(((P.OW + 1) / 2 + 31) / 32) * (((P.OH + 1) / 2 + 7) / 8) * (P.Mo / 128) * P.N * 2); out8[5] = 512;
        return SPI_OK;
    }
    if (d->workspace && pass == 1 && make_hconv_s2(d, P, Wp) && d->workspace_bytes >= spi_hconv_workspace_bytes(Wp)) {
        out8[0] = 2; out8[1] = 128; out8[2] = 256; out8[3] = 1; out8[4] = (int32_t)((int64_t)((P.OW + 31) / 32) * ((P.OH + 7) / 8) * (P.Mo / 128) * P.N); out8[5] = 512;
        return SPI_OK;
    }
    if (d->workspace && make_hconv(d, P, Wp) && d->workspace_bytes >= spi_hconv_workspace_bytes(Wp)) {
        out8[0] = 2; out8[1] = 128; out8[2] = 512; out8[3] = 1; out8[4] = (int32_t)((int64_t)((P.OW + 31) / 32) * ((P.OH + 15) / 16) * (P.Mo / 128) * P.N); out8[5] = 512;
        return SPI_OK;
    }
    if (d->workspace && make_wino(d, P, Wp) && d->workspace_bytes >= spi_wino_workspace_bytes(Wp)) {
        out8[0] = 1; out8[1] = 64; out8[2] = 256; out8[3] = Wp.ksplit; out8[4] = (int32_t)((int64_t)Wp.bx * Wp.by * (Wp.ocp / 64) * P.N * Wp.ksplit); out8[5] = 256;
        return SPI_OK;
    }
    const IGemmPlan plan = plan_igemm(P, d->compute_f16);
    static const int bm_of[6] = {32, 128, 64, 32, 64, 128}, bn_of[6] = {128, 128, 64, 32, 256, 256}, nt_of[6] = {256, 256, 256, 64, 256, 256};
    int maxpix = 0;
    for (int c = 0; c < P.ncls; ++c) maxpix = std::max(maxpix, P.cls[c].OHp * P.cls[c].OWp);
    const int bm = bm_of[plan.cfg], bn = bn_of[plan.cfg];
    out8[0] = 0; out8[1] = bm; out8[2] = bn; out8[3] = plan.nsplit;
    out8[4] = (int32_t)((int64_t)((maxpix + bn - 1) / bn) * ((P.Mo + bm - 1) / bm) * P.N * P.ncls * plan.nsplit); out8[5] = nt_of[plan.cfg];
    return SPI_OK;
}

int spi_conv2d_fwd(const spi_conv_desc* d, const float* x, const float* w, float* y, spi_stream_t stream) {
    int rc = validate(d, "spi_conv2d_fwd"); if (rc) return rc;
    SPI_REQUIRE(x && w && y, "spi_conv2d_fwd: null tensor");
    SPI_REQUIRE(d->act == 0 || d->act == SPI_ACT_LINEAR || d->act == SPI_ACT_RELU || d->act == SPI_ACT_LRELU,
                "spi_conv2d_fwd: fused epilogue supports linear / relu / lrelu only");
    SPI_REQUIRE(!(d->transposed && (d->bias || d->noise || d->act > SPI_ACT_LINEAR)),
                "spi_conv2d_fwd: no fused epilogue in transposed mode (the FIR pass owns it)");
    IGemmParams P; make_forward(d, P);
    if (d->out_seg_flags) { P.out_flags = d->out_seg_flags; P.out_nseg = (int)(((int64_t)P.OH * P.OW + SPI_SEG_PIXELS - 1) / SPI_SEG_PIXELS); }
    Epilogue ep{d->bias, d->noise, d->noise_gain, d->act, d->alpha, d->act ? d->gain : 1.f, d->act ? d->clamp : -1.f};
    WinoParams Wp;
    if (d->workspace && make_hconv_t2(d, P, Wp) && d->workspace_bytes >= spi_hconv_workspace_bytes(Wp)) {
        SPI_REQUIRE((reinterpret_cast<uintptr_t>(d->workspace) & 15) == 0, "spi_conv2d_fwd: workspace must be 16-byte aligned");
        rc = spi_hconv_t2_launch(Wp, x, w, y, d->workspace, as_stream(stream), d->workspace_ready != 0); if (rc) return rc;
        SPI_LAUNCH_CHECK("spi_conv2d_fwd (direct fp16, transposed)");
        return SPI_OK;
    }
    if (d->workspace && make_hconv(d, P, Wp) && d->workspace_bytes >= spi_hconv_workspace_bytes(Wp)) {
        SPI_REQUIRE((reinterpret_cast<uintptr_t>(d->workspace) & 15) == 0, "spi_conv2d_fwd: workspace must be 16-byte aligned");
        rc = spi_hconv_launch(Wp, x, w, y, ep, d->workspace, as_stream(stream), d->workspace_ready != 0); if (rc) return rc;
        SPI_LAUNCH_CHECK("spi_conv2d_fwd (direct fp16)");
        return SPI_OK;
    }
    if (d->workspace && make_wino(d, P, Wp) && d->workspace_bytes >= spi_wino_workspace_bytes(Wp)) {
        SPI_REQUIRE((reinterpret_cast<uintptr_t>(d->workspace) & 15) == 0, "spi_conv2d_fwd: workspace must be 16-byte aligned");
        rc = wino_conv(Wp, P, x, w, y, ep, d->workspace, as_stream(stream), d->out_zeroed != 0, d->workspace_ready != 0); if (rc) return rc;
        SPI_LAUNCH_CHECK("spi_conv2d_fwd (winograd)");
        return SPI_OK;
    }
    rc = dispatch_igemm(P, x, w, y, ep, as_stream(stream), d->compute_f16, d->out_zeroed != 0, d->act_dtype == SPI_DTYPE_F16); if (rc) return rc;
    SPI_LAUNCH_CHECK("spi_conv2d_fwd");
    return SPI_OK;
}

int spi_conv2d_dgrad(const spi_conv_desc* d, const float* dy, const float* w, float* dx, spi_stream_t stream) {
    int rc = validate(d, "spi_conv2d_dgrad"); if (rc) return rc;
    SPI_REQUIRE(dy && w && dx, "spi_conv2d_dgrad: null tensor");
    IGemmParams P; make_dgrad(d, P);
    Epilogue ep{nullptr, nullptr, nullptr, 0, 0.f, 1.f, -1.f};
    WinoParams Wp;
    if (d->workspace && make_hconv_s2(d, P, Wp) && d->workspace_bytes >= spi_hconv_workspace_bytes(Wp)) {
        SPI_REQUIRE((reinterpret_cast<uintptr_t>(d->workspace) & 15) == 0, "spi_conv2d_dgrad: workspace must be 16-byte aligned");
        rc = spi_hconv_s2_launch(Wp, P.IH, P.IW, dy, w, dx, d->workspace, as_stream(stream), d->workspace_ready != 0); if (rc) return rc;
        SPI_LAUNCH_CHECK("spi_conv2d_dgrad (direct fp16, stride 2)");
        return SPI_OK;
    }
    if (d->workspace && make_hconv(d, P, Wp) && d->workspace_bytes >= spi_hconv_workspace_bytes(Wp)) {
        SPI_REQUIRE((reinterpret_cast<uintptr_t>(d->workspace) & 15) == 0, "spi_conv2d_dgrad: workspace must be 16-byte aligned");
        rc = spi_hconv_launch(Wp, dy, w, dx, ep, d->workspace, as_stream(stream), d->workspace_ready != 0); if (rc) return rc;
        SPI_LAUNCH_CHECK("spi_conv2d_dgrad (direct fp16)");
        return SPI_OK;
    }
    if (d->workspace && make_wino(d, P, Wp) && d->workspace_bytes >= spi_wino_workspace_bytes(Wp)) {
        SPI_REQUIRE((reinterpret_cast<uintptr_t>(d->workspace) & 15) == 0, "spi_conv2d_dgrad: workspace must be 16-byte aligned");
        rc = wino_conv(Wp, P, dy, w, dx, ep, d->workspace, as_stream(stream), d->out_zeroed != 0, d->workspace_ready != 0); if (rc) return rc;
        SPI_LAUNCH_CHECK("spi_conv2d_dgrad (winograd)");
        return SPI_OK;
    }
    rc = dispatch_igemm(P, dy, w, dx, ep, as_stream(stream), d->compute_f16, d->out_zeroed != 0, d->act_dtype == SPI_DTYPE_F16); if (rc) return rc;
    SPI_LAUNCH_CHECK("spi_conv2d_dgrad");
    return SPI_OK;
}

int spi_conv2d_wgrad(const spi_conv_desc* d, const float* x, const float* dy, float* dw, spi_stream_t stream) {
    int rc = validate(d, "spi_conv2d_wgrad"); if (rc) return rc;
    SPI_REQUIRE(x && dy && dw, "spi_conv2d_wgrad: null tensor");
    SPI_REQUIRE(d->w_batch_stride == (int64_t)d->O * d->I * d->kh * d->kw || d->N == 1 || d->w_batch_stride == 0,
                "spi_conv2d_wgrad: w_batch_stride must be 0 or O*I*kh*kw");
    IGemmParams P; make_forward(d, P);
    SPI_REQUIRE(P.in_bs * 4 < (1ll << 30) && P.out_bs * 4 < (1ll << 30), "spi_conv2d_wgrad: a per-sample activation must be < 1 GiB");
    const int64_t wsz = (int64_t)d->O * d->I * d->kh * d->kw;
    const int64_t nw = (d->w_batch_stride == 0) ? 1 : d->N;
    {
        WinoParams Wp;
        if (d->workspace && d->workspace_bytes >= WINO_WGRAD_WS && make_hwgrad(d, P, Wp)) {
            const bool parts = d->workspace_bytes >= spi_hwgrad_workspace_bytes(Wp) && (reinterpret_cast<uintptr_t>(d->workspace) & 15) == 0;
            if (!parts && !d->dw_zeroed) spi_zero_async(dw, nw * wsz, as_stream(stream));          // (the partial-sum path overwrites dw)
            rc = spi_hwgrad_launch(Wp, x, dy, dw, parts ? d->workspace : nullptr, d->workspace_bytes, as_stream(stream)); if (rc) return rc;
            SPI_LAUNCH_CHECK("spi_conv2d_wgrad (direct fp16)");
            return SPI_OK;
        }
    }
    if (!d->dw_zeroed) {
        spi_zero_async(dw, nw * wsz, as_stream(stream));
    }
    if (launch_twgrad(d, P, x, dy, dw, as_stream(stream))) {            // stride-2 transposed 3x3, fp32: the nine-tap kernel
        SPI_LAUNCH_CHECK("spi_conv2d_wgrad (transposed, direct)");
        return SPI_OK;
    }
    {
        WinoParams Wp;
        if (d->workspace && d->workspace_bytes >= WINO_WGRAD_WS && make_wino_wgrad(d, P, Wp)) {
            rc = spi_wino_wgrad_launch(Wp, x, dy, dw, as_stream(stream)); if (rc) return rc;
            SPI_LAUNCH_CHECK("spi_conv2d_wgrad (winograd)");
            return SPI_OK;
        }
    }
    // tile 128 x 128; split the pixel reduction so the grid has ~>= 1024 blocks
    // (a 32-row tile for layers with <= 32 output channels -- the 3-channel torgb layers: a 128-row tile spends 97 % of its MFMAs on
    //  rows that do not exist and the kernel should be memory-bound there)
    constexpr int BN = 128;
    const int BM = P.Mo <= 32 ? 32 : 128;
    int maxpix = 0, maxcols = 0;
    for (int c = 0; c < P.ncls; ++c) { maxpix = std::max(maxpix, P.cls[c].OHp * P.cls[c].OWp); maxcols = std::max(maxcols, P.Ci * P.cls[c].taps.T); }
    const int tiles = ((P.Mo + BM - 1) / BM) * ((maxcols + BN - 1) / BN);
    // blocks that do work: the classes of a transposed conv hold 4 / 2 / 2 / 1 taps, i.e. different numbers of column tiles
    int64_t active = 0;
    for (int c = 0; c < P.ncls; ++c) active += (int64_t)((P.Mo + BM - 1) / BM) * ((P.Ci * P.cls[c].taps.T + BN - 1) / BN);
    int64_t splits = std::max<int64_t>(1, 1024 / std::max<int64_t>(1, active * P.N));
    int ppb = (int)((maxpix + splits - 1) / splits);
    ppb = std::max(256, ((ppb + BK - 1) / BK) * BK);     // >= 16 slabs per block: below that the 16 K atomics of a tile cost more than its MFMAs
    dim3 grid((unsigned)((maxpix + ppb - 1) / ppb), (unsigned)tiles, (unsigned)(P.N * P.ncls));
    // measured: the scalar-offset variant wins when a tap is ONE column tile (Ci == 128: +6 %) and loses for Ci >= 256 (-10 %)
    bool fastw = (P.Ci == BN) && (P.Mo % 128 == 0);
    for (int c = 0; c < P.ncls; ++c) fastw = fastw && P.cls[c].OWp >= BK;
    const bool sparse = d->dy_seg_flags != nullptr && ppb / BK + 1 <= WG_LISTMAX && (maxpix + BK - 1) / BK <= 64 * 256;
    if (sparse) { P.seg_flags = d->dy_seg_flags; P.nseg = (int)(((int64_t)P.OH * P.OW + SPI_SEG_PIXELS - 1) / SPI_SEG_PIXELS); }
#define SPI_WG_LAUNCH(PR, IOF, FASTF, SPF) hipLaunchKernelGGL((wgrad_kernel<2, 2, 2, 2, PR, FASTF, SPF, IOF>), grid, dim3(256), 0, as_stream(stream), P, x, dy, dw, ppb)
#define SPI_WG_SKINNY(PR, IOF, SPF) hipLaunchKernelGGL((wgrad_kernel<1, 4, 1, 1, PR, false, SPF, IOF>), grid, dim3(256), 0, as_stream(stream), P, x, dy, dw, ppb)
#define SPI_WG_BY_PREC(CALL, ...) do { switch (prec) { case 1: if (ioh) CALL(1, true, __VA_ARGS__); else CALL(1, false, __VA_ARGS__); break; case 2: CALL(2, false, __VA_ARGS__); break; \
                                                         case 3: CALL(3, false, __VA_ARGS__); break; default: CALL(0, false, __VA_ARGS__); } } while (0)
    const int prec = d->compute_f16;
    const bool ioh = d->act_dtype == SPI_DTYPE_F16;
    SPI_REQUIRE(!ioh || (P.in_bs * 2 < (1ll << 30) && P.out_bs * 2 < (1ll << 30)), "spi_conv2d_wgrad: a per-sample activation must be < 1 GiB");
    if (BM == 32) { if (sparse) SPI_WG_BY_PREC(SPI_WG_SKINNY, true); else SPI_WG_BY_PREC(SPI_WG_SKINNY, false); }
    else if (sparse) SPI_WG_BY_PREC(SPI_WG_LAUNCH, false, true);
    else if (fastw) SPI_WG_BY_PREC(SPI_WG_LAUNCH, true, false);
    else SPI_WG_BY_PREC(SPI_WG_LAUNCH, false, false);
#undef SPI_WG_BY_PREC
#undef SPI_WG_SKINNY
#undef SPI_WG_LAUNCH
    SPI_LAUNCH_CHECK("spi_conv2d_wgrad");
    return SPI_OK;
}

}  // extern "C"
