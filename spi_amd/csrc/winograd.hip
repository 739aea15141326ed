// Winograd F(2x2, 3x3) convolutions on the gfx950 matrix cores, fp32 throughout.
//
// The 3x3 / stride-1 / pad-1 convolutions of the generator's 128^2 .. 512^2 layers (and their data gradients, which are
// convolutions of the same shape with flipped, transposed weights) carry two thirds of the step's FLOPs.  Minimal filtering
// (Lavin & Gray) computes a 2x2 output tile from a 4x4 input patch with 16 multiplications per (in, out) channel pair
// instead of 36:
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A           g: 3x3 taps, d: 4x4 patch, Y: 2x2 outputs
//
// so the channel reduction becomes 16 independent GEMMs  M_f[oc, tile] = sum_c U_f[oc, c] V_f[c, tile]  (f = the 16
// "frequencies"), 2.25x fewer MFMAs than the implicit GEMM of conv.hip, still v_mfma_f32_32x32x2_f32 with fp32 operands and
// fp32 accumulation (the transforms add and halve only: the result differs from the direct sum by a few fp32 roundings --
// this is the algorithm cuDNN picks for the reference's fp32 3x3 convolutions).
//
// One block = 8 x 8 tiles (16 x 16 output pixels) x 64 output channels x all 16 frequencies = 65536 accumulators
// = 256 per lane (AGPRs; one wave per SIMD).  Per slab of 8 input channels:
//   * U (transformed weights, produced once per call by wino_weight_kernel in exactly the LDS image) and the slab's raw
//     8 x 18 x 18 input window go global -> LDS directly (global_load_lds_dwordx4 / buffer_load_dword ... lds: no registers
//     in between, out-of-image pixels arrive as 0), one slab resp. two slabs ahead;
//   * every thread reads the 4x4 patches of one tile for two channels from the raw window, transforms them (32 adds per
//     patch) and writes V to LDS -- in the shadow of the MFMAs of the previous slab;
//   * every wave runs 64 MFMAs (16 frequencies x 4 k-steps) on its 32 oc x 32 tile quadrant, operands read as one
//     16-byte LDS fragment per (frequency, operand) = 4 k-steps.
// The output transform (A^T M A), noise / bias / activation epilogue and the 2x2 stores run on the accumulators in registers.
#include "common.hpp"

// LDS-DMA loads take their LDS destination from M0.  M0 is a register the compiler manages itself (LDS instruction bounds, other DMA
// builtins): an inline-asm statement that names it as a clobber is flagged by hipcc ("reserved register ... undefined behaviour"), because the
// compiler may keep a value of its own in M0 across the statement.  So M0 is SAVED and RESTORED inside the same statement (two extra
// scalar moves per transfer, off the vector issue port): whatever the compiler had there is intact when the statement ends.
#define SPI_LDS_DMA_GLOBAL_X4(dst, voff, sbase)                                                                                   \
    do { unsigned m0_keep_;                                                                                                      \
         asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"                \
                      : "=&s"(m0_keep_) : "s"(dst), "v"(voff), "s"(sbase) : "memory"); } while (0)
#define SPI_LDS_DMA_BUFFER_X1(dst, voff, rsrc, soff)                                                                              \
    do { unsigned m0_keep_;                                                                                                      \
         asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"        \
                      : "=&s"(m0_keep_) : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory"); } while (0)


typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WKC = 8;                 // input channels per slab
constexpr int WOC = 64;                // output channels per block
constexpr int WSLAB = 16 * 2 * 64 * 4; // floats of one operand slab in LDS: [f 16][h 2][row 64][j 4], channel k = 2j + h
constexpr int WRAW = 41 * 64;          // floats of one raw input window in LDS: [8 channels][18][18] = 2592, rounded up to whole wave transfers

// ---- weights: U[n][slab][f][h][ocp][j] = (G g G^T)[f] of channel c = slab*8 + 2j + h, output channel oc (zero rows up to ocp)
__global__ void __launch_bounds__(256) wino_weight_kernel(WinoParams P, const float* __restrict__ w, float* __restrict__ U) {
    const int64_t total = (int64_t)P.nw * (P.Ci / WKC) * 2 * P.ocp;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        const int oc = (int)(g % P.ocp);
        int64_t r = g / P.ocp;
        const int h = (int)(r & 1); r >>= 1;
        const int slab = (int)(r % (P.Ci / WKC));
        const int n = (int)(r / (P.Ci / WKC));
        float u[4][16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gk[3][3];
#pragma unroll
            for (int t = 0; t < 9; ++t) gk[t / 3][t % 3] = 0.f;
            if (oc < P.Mo) {
                const float* wp = w + (int64_t)n * P.wbs + (int64_t)oc * P.wsm + (int64_t)(slab * WKC + 2 * j + h) * P.wsc;
#pragma unroll
                for (int t = 0; t < 9; ++t) gk[t / 3][t % 3] = wp[P.widx[t]];       // widx[(dy+1)*3 + (dx+1)]
            }
            float t4[4][3];                                                          // G g
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                t4[0][s] = gk[0][s];
                t4[1][s] = 0.5f * (gk[0][s] + gk[1][s] + gk[2][s]);
                t4[2][s] = 0.5f * (gk[0][s] - gk[1][s] + gk[2][s]);
                t4[3][s] = gk[2][s];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {                                            // (G g) G^T
                u[j][a * 4 + 0] = t4[a][0];
                u[j][a * 4 + 1] = 0.5f * (t4[a][0] + t4[a][1] + t4[a][2]);
                u[j][a * 4 + 2] = 0.5f * (t4[a][0] - t4[a][1] + t4[a][2]);
                u[j][a * 4 + 3] = t4[a][2];
            }
        }
        float4* dst = reinterpret_cast<float4*>(U + (int64_t)n * P.u_bs) + ((int64_t)(slab * 16) * 2 + h) * P.ocp + oc;
#pragma unroll
        for (int f = 0; f < 16; ++f) dst[(int64_t)f * 2 * P.ocp] = make_float4(u[0][f], u[1][f], u[2][f], u[3][f]);
    }
}

__device__ __forceinline__ float wino_act(const WinoEpilogue& e, float v) {
    return conv_act_gain_clamp(e.act, e.alpha, e.gain, e.clamp, v);
}

__global__ void __launch_bounds__(256, 1) wino_conv_kernel(WinoParams P, const float* __restrict__ in, const float* __restrict__ U,
                                                           float* __restrict__ out, WinoEpilogue ep) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // Us[2][WSLAB] | Vs[2][WSLAB] | Rs[2][WRAW] = 148.5 KB
    float* Us = lds;
    float* Vs = lds + 2 * WSLAB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31;
    const int ocw = wave >> 1, tw = wave & 1;                         // MFMA quadrant: 32 output channels x 32 tiles
    const int ksplit = P.ksplit > 1 ? P.ksplit : 1;
    const int n = blockIdx.z / ksplit, ks = blockIdx.z - n * ksplit, oc0 = blockIdx.y * WOC;
    const int bxi = blockIdx.x % P.bx, byi = blockIdx.x / P.bx;
    const int oy0 = byi * 16, ox0 = bxi * 16;
    const int64_t HW = (int64_t)P.H * P.W;
    float* ob = out + (int64_t)n * P.out_bs;

    // ---- needed-output map (forward) / zero-segment map of the gradient operand (dgrad): nothing flagged in the block's
    //      output rows resp. receptive field -> the result is exactly zero: write it and leave.
    if (P.out_flags || P.seg_flags) {
        const int32_t* fl = (P.out_flags ? P.out_flags : P.seg_flags) + (int64_t)n * P.nseg;
        const int halo = P.out_flags ? 0 : 1;
        const int ylo = max(oy0 - halo, 0), yhi = min(oy0 + 15 + halo, P.H - 1);
        const int xlo = max(ox0 - halo, 0), xhi = min(ox0 + 15 + halo, P.W - 1);
        int any = 0;
        const int rows = yhi - ylo + 1;
        for (int e = tid; e < rows * 4; e += 256) {                  // <= 4 segments per row (18 pixels span at most 3)
            const int iy = ylo + (e >> 2);
            const int sg = ((iy * P.W + xlo) >> 4) + (e & 3);
            if (sg <= ((iy * P.W + xhi) >> 4)) any |= fl[sg];
        }
        if (!__syncthreads_or(any)) {
            if (ksplit > 1) return;                                  // split reduction: out was zeroed by the launcher
            for (int e = tid; e < WOC * 256; e += 256) {
                const int m = oc0 + (e >> 8), yy = oy0 + ((e >> 4) & 15), xx = ox0 + (e & 15);
                if (m < P.Mo && yy < P.H && xx < P.W) ob[(int64_t)m * HW + (int64_t)yy * P.W + xx] = 0.f;
            }
            return;
        }
    }

    // ---- data movement: no global load of the main loop goes through registers.
    //  * raw input: the slab's 8 x 18 x 18 window (16 x 16 pixels + halo) lands in LDS slot by slot (buffer_load_dword ... lds: lane l of
    //    a wave writes LDS[M0 + 4 l] from its OWN global address, out-of-image offsets return 0): 41 wave-instructions per slab,
    //    per-lane offsets fixed for the whole kernel, the slab's channel base is the scalar offset;
    //  * U: eight 16-byte transfers per thread (global_load_lds_dwordx4).
    // Inline assembly on purpose: through the builtins hipcc treats every later LDS access as a possible alias of a transfer in flight
    // and waits vmcnt(0) in front of it.  Here every transfer of an iteration is issued at its top and awaited once, before its barrier.
    const __amdgpu_buffer_rsrc_t rsI = make_rsrc(in + (int64_t)n * P.in_bs, P.in_bs * 4);
    const int chs4 = __builtin_amdgcn_readfirstlane((int)HW * 4);
    const int nslab = P.Ci / WKC;
    // this block's range of slabs (the whole reduction unless the layer is split)
    const int s_beg = (int)((int64_t)nslab * ks / ksplit), s_end = (int)((int64_t)nslab * (ks + 1) / ksplit);
    unsigned voffR[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        // LDS image of the raw window: [4 channel pairs][18 rows][18 columns][2 channels of the pair] -- a thread's patch row of BOTH its channels is
        // two 16-byte reads and the transform runs on (channel 0, channel 1) pairs (v_pk_add_f32) without shuffling registers
        const int slot = (i * 4 + wave) * 64 + lane;
        const int pairq = slot / 648, rem2 = slot - pairq * 648, r = rem2 / 36, c = (rem2 - r * 36) >> 1;
        const int ch = (pairq >> 1) * 4 + (pairq & 1) + 2 * (slot & 1);      // pair p = wave p's two channels: (0,2) (1,3) (4,6) (5,7)
        const int iy = oy0 - 1 + r, ix = ox0 - 1 + c;
        voffR[i] = (slot < WKC * 324 && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W) ? (unsigned)(ch * chs4 + (iy * P.W + ix) * 4) : BUF_OOB;
    }
    unsigned uvoff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) uvoff[i] = (unsigned)((((int64_t)(i * 4 + wave) * P.ocp + oc0 + lane) * 16));
    const char* Ubase = reinterpret_cast<const char*>(U + (int64_t)n * P.u_bs);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    auto copy_u = [&](int buf, int s, int i) {                        // segment i (0..7) of U[s]
        s = min(s, nslab - 1);
        const char* src = Ubase + (int64_t)s * 32 * P.ocp * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * WSLAB + (((i * 4 + wave) * 64) << 2)) * 4));
        SPI_LDS_DMA_GLOBAL_X4(dst, uvoff[i], src);
    };
    auto copy_raw = [&](int rbuf, int s, int i) {                     // wave transfer i (0..10) of the raw window of slab s
        s = min(s, nslab - 1);
        const int soff = __builtin_amdgcn_readfirstlane(s * WKC * chs4);
        if (i < 10 || wave == 0) {                                   // 41 wave transfers: the last round is wave 0's alone
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((4 * WSLAB + rbuf * WRAW + (i * 4 + wave) * 64) * 4));
            SPI_LDS_DMA_BUFFER_X1(dst, voffR[i], rsI, soff);
        }
    };

    // ---- transform: tile = lane (8 x 8 tiles), channels k = 2*(jb + q) + hch, q = 0, 1: the 4x4 patches of BOTH channels as (q = 0, q = 1) pairs;
    //      the raw window stores exactly these pairs interleaved (pair index = wave: channels (0,2) (1,3) (4,6) (5,7), see voffR above)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int hch = wave & 1, jb = (wave >> 1) * 2;
    const int ty = lane >> 3, tx = lane & 7;
    f32x2 d2[16];
    auto read_patch_row = [&](const float* Rb, int r) {               // row r of both channels' patches: two 16-byte reads
        const float* pr = Rb + ((((jb >> 1) * 2 + hch) * 18 + (2 * ty + r)) * 18 + 2 * tx) * 2;
        const float4 lo = *reinterpret_cast<const float4*>(pr), hi = *reinterpret_cast<const float4*>(pr + 4);
        d2[r * 4 + 0] = f32x2{lo.x, lo.y}; d2[r * 4 + 1] = f32x2{lo.z, lo.w}; d2[r * 4 + 2] = f32x2{hi.x, hi.y}; d2[r * 4 + 3] = f32x2{hi.z, hi.w};
    };
    // B^T d B: row group a (4 of the 16 frequencies) of both channels' patches: 8 packed adds (v_pk_add_f32 spelled out -- left to itself the
    // compiler scalarises about half of these vector adds again) ...
    auto pk_add = [](f32x2 x, f32x2 y) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
    auto pk_sub = [](f32x2 x, f32x2 y) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y)); return r; };
    f32x2 vt2[4];
    auto transform_rows = [&](int a, int) {
        f32x2 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            t[c] = a == 0 ? pk_sub(d2[0 + c], d2[8 + c]) : a == 1 ? pk_add(d2[4 + c], d2[8 + c]) : a == 2 ? pk_sub(d2[8 + c], d2[4 + c]) : pk_sub(d2[4 + c], d2[12 + c]);
        vt2[0] = pk_sub(t[0], t[2]); vt2[1] = pk_add(t[1], t[2]); vt2[2] = pk_sub(t[2], t[1]); vt2[3] = pk_sub(t[1], t[3]);
    };
    // ... and the two channels of a frequency written as one 8-byte store
    auto store_rows = [&](float* Vb, int a) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
            *reinterpret_cast<f32x2*>(Vb + ((((a * 4 + b) * 2 + hch) * 64 + lane) << 2) + jb) = vt2[b];
    };
    auto transform_store = [&](float* Vb, int a) { transform_rows(a, 0); store_rows(Vb, a); };

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    float* Rs = lds + 4 * WSLAB;
    // prologue: U[0], raw[0], raw[1] -> LDS; V[0] from raw[0]
#pragma unroll
    for (int i = 0; i < 11; ++i) { if (i < 8) copy_u(0, s_beg, i); copy_raw(0, s_beg, i); copy_raw(1, s_beg + 1, i); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) read_patch_row(Rs, r);
#pragma unroll
    for (int a = 0; a < 4; ++a) transform_store(Vs, a);
    __syncthreads();

    const int aoff = ((h * 64 + ocw * 32 + l32) << 2), boff = ((h * 64 + tw * 32 + l32) << 2);
    for (int s = s_beg; s < s_end; ++s) {
        const int buf = (s - s_beg) & 1;
        const float* Ub = Us + buf * WSLAB + aoff;
        const float* Vb = Vs + buf * WSLAB + boff;
        float* Vw = Vs + (buf ^ 1) * WSLAB;
        // 4 frequencies per group, k-step outermost: consecutive MFMAs go to different accumulators.  Behind EVERY MFMA one micro-slot of
        // side work is issued (64 per slab; an MFMA occupies the pipe for 64 cycles after a 4-cycle issue, so a handful of
        // instructions per slot run in its shadow, while a clump of 20 behind four MFMAs would leave the pipe idle):
        //   * the next group's operand fragments (first 8 slots of a group),
        //   * this iteration's transfers: raw[s+2] -> the raw buffer slab s lived in (11), U[s+1] -> the other U buffer (8),
        //   * raw[s+1] -> V[s+1]: patch rows (4 slots), then per row group of the transform: channel 0, channel 1, the stores.
        const float* Rn = Rs + (buf ^ 1) * WRAW;
        float4 af[2][4], bf[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[0][i] = *reinterpret_cast<const float4*>(Ub + i * 512);
            bf[0][i] = *reinterpret_cast<const float4*>(Vb + i * 512);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = g * 16 + j * 4 + i, t = m & 15;
            acc[g * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][j], bf[g & 1][i][j], acc[g * 4 + i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < 4 && t < 8) {
                if (t < 4) af[(g + 1) & 1][t] = *reinterpret_cast<const float4*>(Ub + ((g + 1) * 4 + t) * 512);
                else bf[(g + 1) & 1][t - 4] = *reinterpret_cast<const float4*>(Vb + ((g + 1) * 4 + t - 4) * 512);
            }
            if ((m & 3) == 1 && (m >> 2) < 11) copy_raw(buf, s + 2, m >> 2);
            if ((m & 3) == 3 && (m >> 2) < 8) copy_u(buf ^ 1, s + 1, m >> 2);
            if ((m & 3) == 2 && (m >> 2) < 4) read_patch_row(Rn, m >> 2);
            if (m >= 24 && m < 40) {
                const int a = (m - 24) >> 2, ph = (m - 24) & 3;
                if (ph == 0) transform_rows(a, 0);
                else if (ph == 2) store_rows(Vw, a);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this iteration's transfers have landed
        __syncthreads();
    }

    // ---- output transform A^T M A + epilogue; C/D layout: col = lane & 31 (tile), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int tile = tw * 32 + l32;
    const int oy = oy0 + 2 * (tile >> 3), ox = ox0 + 2 * (tile & 7);
    if (oy >= P.H || ox >= P.W) return;
    const bool y1 = oy + 1 < P.H, x1 = ox + 1 < P.W;
    const int64_t pix = (int64_t)oy * P.W + ox;
    float nz[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (ep.noise) {
        const float ng = ep.noise_gain ? ep.noise_gain[0] : 1.f;
        nz[0][0] = ep.noise[pix] * ng;
        if (x1) nz[0][1] = ep.noise[pix + 1] * ng;
        if (y1) { nz[1][0] = ep.noise[pix + P.W] * ng; if (x1) nz[1][1] = ep.noise[pix + P.W + 1] * ng; }
    }
    if (ksplit > 1) {                                                 // partial sums of a channel range: no epilogue here, four atomics per tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = oc0 + ocw * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= P.Mo) continue;
            float s0[4], s1[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                s0[a] = acc[a * 4 + 0][r] + acc[a * 4 + 1][r] + acc[a * 4 + 2][r];
                s1[a] = acc[a * 4 + 1][r] - acc[a * 4 + 2][r] - acc[a * 4 + 3][r];
            }
            float* dst = ob + (int64_t)m * HW + pix;
            atomicAdd(dst, s0[0] + s0[1] + s0[2]);
            if (x1) atomicAdd(dst + 1, s1[0] + s1[1] + s1[2]);
            if (y1) { atomicAdd(dst + P.W, s0[1] - s0[2] - s0[3]); if (x1) atomicAdd(dst + P.W + 1, s1[1] - s1[2] - s1[3]); }
        }
        return;
    }
    // interior blocks (all 64 channels, all 16 x 16 pixels, even row length): no per-element bounds tests.  The activation is the same
    //   conv_act_gain_clamp as the border blocks and the implicit GEMM: NaN and +-inf propagate identically in every block of a layer,
    //   so a diverging run shows up in the loss wherever the bad value sits.
    if (oc0 + WOC <= P.Mo && oy0 + 16 <= P.H && ox0 + 16 <= P.W && (P.W & 1) == 0) {
        const bool has_epi = ep.act != 0 || ep.bias || ep.noise;
        float* dst0 = ob + (int64_t)(oc0 + ocw * 32 + 4 * h) * HW + pix;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mo = (r & 3) + 8 * (r >> 2);
            float s0[4], s1[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                s0[a] = acc[a * 4 + 0][r] + acc[a * 4 + 1][r] + acc[a * 4 + 2][r];
                s1[a] = acc[a * 4 + 1][r] - acc[a * 4 + 2][r] - acc[a * 4 + 3][r];
            }
            float y[2][2];
            y[0][0] = s0[0] + s0[1] + s0[2]; y[0][1] = s1[0] + s1[1] + s1[2];
            y[1][0] = s0[1] - s0[2] - s0[3]; y[1][1] = s1[1] - s1[2] - s1[3];
            if (has_epi) {
                const float bv = ep.bias ? ep.bias[oc0 + ocw * 32 + 4 * h + mo] : 0.f;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        float v = y[a][b] + nz[a][b] + bv;
                        y[a][b] = conv_act_gain_clamp(ep.act, ep.alpha, ep.gain, ep.clamp, v);      // block-uniform branches
                    }
            }
            float* dst = dst0 + (int64_t)mo * HW;
            *reinterpret_cast<float2*>(dst) = make_float2(y[0][0], y[0][1]);
            *reinterpret_cast<float2*>(dst + P.W) = make_float2(y[1][0], y[1][1]);
        }
    } else {
    const bool vec = x1 && ((P.W & 1) == 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = oc0 + ocw * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= P.Mo) continue;
        float s0[4], s1[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            s0[a] = acc[a * 4 + 0][r] + acc[a * 4 + 1][r] + acc[a * 4 + 2][r];
            s1[a] = acc[a * 4 + 1][r] - acc[a * 4 + 2][r] - acc[a * 4 + 3][r];
        }
        float y[2][2];
        y[0][0] = s0[0] + s0[1] + s0[2]; y[0][1] = s1[0] + s1[1] + s1[2];
        y[1][0] = s0[1] - s0[2] - s0[3]; y[1][1] = s1[1] - s1[2] - s1[3];
        const float bv = ep.bias ? ep.bias[m] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float v = y[a][b] + nz[a][b] + bv;
                if (ep.act) v = wino_act(ep, v);
                y[a][b] = v;
            }
        float* dst = ob + (int64_t)m * HW + pix;
        if (vec) {
            *reinterpret_cast<float2*>(dst) = make_float2(y[0][0], y[0][1]);
            if (y1) *reinterpret_cast<float2*>(dst + P.W) = make_float2(y[1][0], y[1][1]);
        } else {
            dst[0] = y[0][0];
            if (x1) dst[1] = y[0][1];
            if (y1) { dst[P.W] = y[1][0]; if (x1) dst[P.W + 1] = y[1][1]; }
        }
    }
    }
}

// =================================================================================================
// Weight gradient of the same convolutions: minimal filtering F(3x3, 2x2).
//
//     dW[ky][kx] = sum over 2x2 tiles of  sum_{jy,jx} dY[jy][jx] * X[jy + ky][jx + kx]        (X: the tile's 4x4 input patch)
//                = A^T [ sum_tiles (G dY G^T) (.) (B^T X B) ] A        A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 1]   (3x4)
//                                                                      G   = [1 0; 1/2 1/2; 1/2 -1/2; 0 1]
//                                                                      B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 -1 0 1]
// 16 multiplications per (in, out) channel pair and tile instead of 36, as 16 independent GEMMs M_f[oc][ic] = sum_t D_f[oc][t] X_f[ic][t]
// whose reduction index is the TILE: v_mfma_f32_32x32x2_f32 with k = 2 tiles.  The halves of G are folded into the output transform
// (G' = [1 0; 1 1; 1 -1; 0 1], M_f scaled by s_a s_b, s = [1, 1/2, 1/2, 1]), so both operand transforms only add.
//
// One block = 64 output x 64 input channels x 16 frequencies = 65536 accumulators (256 per lane, AGPRs, one wave per SIMD) over a
// 32-pixel-wide strip of the image, walked one tile row (16 tiles) per step.  Per step:
//   * raw rows go global -> LDS directly (buffer_load_dword ... lds, out-of-image / out-of-range channels arrive as 0): the 2 new input rows
//     and the 2 gradient rows of the NEXT step into ring buffers (6 input rows, 4 gradient rows, [row][64 channels][34 floats]);
//   * a lane of MFMA row / column c and k-half h reads the raw 2x2 gradient tile resp. 4x4 input patch of ITS channel and tile from LDS
//     (8-byte reads, channel stride 34 floats: conflict-free), transforms it in registers (12 resp. 32 adds) and the 16 results ARE its
//     A resp. B operands of the 16 frequencies -- the transformed operands never touch LDS;
//   * every wave runs 128 MFMAs (16 frequencies x 8 tile pairs) on its 32 x 32 channel quadrant; the reads and transforms of tile pair
//     p + 1 and the step's 34 row transfers are issued in micro-slots behind the MFMAs of pair p.
// The reduction over the image is split across blocks (strips x row chunks x samples); each block applies the output transform to its
// partial sums in registers and adds the 9 taps into dW with fp32 atomics (dW zeroed by the caller side, like the implicit-GEMM kernel).
// =================================================================================================
constexpr int GPX = 34;                 // floats per LDS row of a channel: 32 strip pixels + 2 (input rows: the halo pair; gradient rows: unused)
constexpr int GROW = 64 * GPX;          // floats of one ring row [64 channels][34] = 34 wave transfers
constexpr int GXR = 6, GDR = 4;         // ring depths: input rows (4 in use + 2 arriving), gradient rows (2 + 2)
constexpr int GLDS = (GXR + GDR) * GROW; // 21760 floats = 85 KB

struct WinoWgradParams {
    int N, nw, Mo, Ci, H, W;
    int strips, nci;                    // 32-pixel strips per row; 64-channel input blocks
    int tiles_per_chunk;                // tile rows per block
    int64_t in_bs, out_bs, wbs;
    int wsm, wsc, widx[9];              // dw[n*wbs + m*wsm + c*wsc + widx[ky*3 + kx]]
    const int32_t* seg_flags; int nseg; // optional zero-segment map of dy ([N, nseg] over flat pixels / 16, spi_conv_desc.dy_seg_flags) or NULL
};

__global__ void __launch_bounds__(256, 1) wino_wgrad_kernel(WinoWgradParams P, const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dw) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // X ring [6][64][34] | D ring [4][64][34]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31;
    const int ocw = wave >> 1, icw = wave & 1;                         // MFMA quadrant: 32 output x 32 input channels
    const int n = blockIdx.z;
    const int cob = (blockIdx.y / P.nci) * 64, cib = (blockIdx.y % P.nci) * 64;
    const int strip = blockIdx.x % P.strips, chunk = blockIdx.x / P.strips;
    const int px0 = strip * 32;
    const int tiles_y = (P.H + 1) >> 1;
    const int ty_beg = chunk * P.tiles_per_chunk, ty_end = min(ty_beg + P.tiles_per_chunk, tiles_y);
    if (ty_beg >= ty_end) return;
    const int HW = P.H * P.W;
    // the block's 64 channels of each operand as their own buffers: channels past the tensor's end are out of range = 0
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(x + (int64_t)n * P.in_bs + (int64_t)cib * HW, (int64_t)min(64, P.Ci - cib) * HW * 4);
    const __amdgpu_buffer_rsrc_t rsD = make_rsrc(dy + (int64_t)n * P.out_bs + (int64_t)cob * HW, (int64_t)min(64, P.Mo - cob) * HW * 4);

    // ---- row transfers.  A row PAIR (two consecutive image rows into two consecutive ring rows) is 68 wave transfers, transfer t = 4 i + wave
    //      (i = 0..16) of it is this wave's: ring row r = t / 34, slots (t % 34) * 64 + lane = (channel, pixel) of that row.  The per-lane
    //      offsets are fixed for the whole kernel; the image row rides in the scalar offset (not range-checked by the hardware, so rows outside
    //      the image select the out-of-range lane offset instead).
    unsigned patX[17], patD[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        const int t = 4 * i + wave, q = t % 34;
        const int slot = q * 64 + lane, ch = slot / GPX, px = slot - ch * GPX;
        const int gx = px0 - 1 + px, gd = px0 + px;
        patX[i] = (gx >= 0 && gx < P.W) ? (unsigned)((ch * HW + gx) * 4) : BUF_OOB;
        patD[i] = (px < 32 && gd < P.W) ? (unsigned)((ch * HW + gd) * 4) : BUF_OOB;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    // transfer i of the pair whose first image row is y0, into ring rows slot0, slot0 + 1 of the X (kind 0) or D (kind 1) ring
    auto xfer = [&](int kind, int y0, int slot0, int i) {
        const int t = 4 * i + wave;
        const int r = t >= 34 ? 1 : 0;                                 // (wave-uniform: t is)
        const int q = t - 34 * r;
        const int y = y0 + r;
        const bool rok = y >= 0 && y < P.H;
        const unsigned vo = rok ? (kind ? patD[i] : patX[i]) : BUF_OOB;
        const int soff = __builtin_amdgcn_readfirstlane(rok ? y * P.W * 4 : 0);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(((kind ? GXR * GROW : 0) + (slot0 + r) * GROW + q * 64) * 4));
        if (kind) SPI_LDS_DMA_BUFFER_X1(dst, vo, rsD, soff);
        else      SPI_LDS_DMA_BUFFER_X1(dst, vo, rsX, soff);
    };

    // ---- operands: raw patch of (channel l32 of the quadrant, tile 2 p + h) -> 16 frequencies, in registers
    typedef float f32x2w __attribute__((ext_vector_type(2)));
    f32x2w rx2[8], rd2[2];                                            // raw patch rows as (column 0, column 1) / (column 2, column 3) pairs
    f32x2w opa[2][8], opb[2][8];                                      // operand f = 4 a + b lives in pair [2 a + (b >> 1)], half b & 1
    const float* xrow[4]; const float* drow[2];
    auto set_rows = [&](int ty) {                                     // ring rows of tile row ty: input rows 2 ty - 1 .. 2 ty + 2, gradient rows 2 ty, 2 ty + 1
        const int xs = (2 * ty) % GXR, ds = (2 * ty) % GDR;
#pragma unroll
        for (int k = 0; k < 4; ++k) xrow[k] = lds + ((xs + k) % GXR) * GROW + (icw * 32 + l32) * GPX + 2 * h;
#pragma unroll
        for (int k = 0; k < 2; ++k) drow[k] = lds + (GXR + ds + k) * GROW + (ocw * 32 + l32) * GPX + 2 * h;
    };
    auto read_x = [&](int p, int k) {                                 // row k of the input patch of tile pair p
        rx2[k * 2 + 0] = *reinterpret_cast<const f32x2w*>(xrow[k] + 4 * p);
        rx2[k * 2 + 1] = *reinterpret_cast<const f32x2w*>(xrow[k] + 4 * p + 2);
    };
    auto read_d = [&](int p, int k) { rd2[k] = *reinterpret_cast<const f32x2w*>(drow[k] + 4 * p); };
    // packed adds with half selects, spelled out (the compiler scalarises vector expressions of this shape and adds register moves).  Volatile: they
    // stay in the micro-slot they are written in (LLVM would sink pure adds to their use, in front of the NEXT pair's MFMAs).
    auto pk_add = [](f32x2w x, f32x2w y) { f32x2w r; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
    auto pk_sub = [](f32x2w x, f32x2w y) { f32x2w r; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y)); return r; };
    // (x.lo - y.lo, x.hi + y.lo)
    auto pk_mlo_plo = [](f32x2w x, f32x2w y) { f32x2w r; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(x), "v"(y)); return r; };
    // (x.lo - y.hi, x.hi - y.hi)
    auto pk_sub_hi = [](f32x2w x, f32x2w y) { f32x2w r; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y)); return r; };
    // (x.lo + x.hi, x.lo - x.hi)
    auto pk_sum_diff = [](f32x2w x) { f32x2w r; asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x)); return r; };
    // B^T X B, row group a (4 of the 16 frequencies): rows x0 - x2, x1 + x2, x2 - x1, x3 - x1 on both column pairs, then the same along the columns
    auto xform_x = [&](f32x2w (&o)[8], int a) {
        const int r0 = a == 0 ? 0 : a == 1 ? 1 : a == 2 ? 2 : 3, r1 = a == 0 ? 2 : a == 1 ? 2 : 1;
        const f32x2w tl = a == 1 ? pk_add(rx2[r0 * 2], rx2[r1 * 2]) : pk_sub(rx2[r0 * 2], rx2[r1 * 2]);
        const f32x2w th = a == 1 ? pk_add(rx2[r0 * 2 + 1], rx2[r1 * 2 + 1]) : pk_sub(rx2[r0 * 2 + 1], rx2[r1 * 2 + 1]);
        o[a * 2 + 0] = pk_mlo_plo(tl, th);                            // t0 - t2, t1 + t2
        o[a * 2 + 1] = pk_sub_hi(th, tl);                             // t2 - t1, t3 - t1
    };
    // G' dY G'^T (the halves of G live in the output transform): rows d0, d0 + d1, d0 - d1, d1; columns u0, u0 + u1, u0 - u1, u1
    auto xform_d = [&](f32x2w (&o)[8]) {
        const f32x2w u[4] = {rd2[0], pk_add(rd2[0], rd2[1]), pk_sub(rd2[0], rd2[1]), rd2[1]};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            o[a * 2 + 0] = u[a];                                      // (b = 0, b = 3)   -- rows 0 and 3 are copies of the raw pairs (the raw registers are re-read next)
            o[a * 2 + 1] = pk_sum_diff(u[a]);                         // (b = 1, b = 2)
        }
    };
    // operand of frequency f = 4 a + b out of the pair arrays
    auto fa = [](const f32x2w (&o)[8], int f) { const int a = f >> 2, b = f & 3; return b == 0 ? o[a * 2].x : b == 3 ? o[a * 2].y : b == 1 ? o[a * 2 + 1].x : o[a * 2 + 1].y; };
    auto fb = [](const f32x2w (&o)[8], int f) { const int a = f >> 2, b = f & 3; return (b & 1) ? o[a * 2 + (b >> 1)].y : o[a * 2 + (b >> 1)].x; };

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // ---- (re)start of the pipeline at tile row ty: its 4 input rows and 2 gradient rows, and its first tile pair transformed
    auto prime = [&](int ty) {
        const int y = 2 * ty;
#pragma unroll
        for (int i = 0; i < 17; ++i) { xfer(0, y - 1, y % GXR, i); xfer(0, y + 1, (y + 2) % GXR, i); xfer(1, y, y % GDR, i); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        set_rows(ty);
#pragma unroll
        for (int k = 0; k < 4; ++k) read_x(0, k);
        read_d(0, 0); read_d(0, 1);
#pragma unroll
        for (int a = 0; a < 4; ++a) xform_x(opb[0], a);
        xform_d(opa[0]);
    };
    // masked losses: a tile row whose two gradient rows hold no flagged 16-pixel segment inside the strip multiplies by zeros -- skipped,
    // and the pipeline restarts at the next row that does.  The block's rows are classified ONCE, here (lane i of wave 0 looks at tile row
    // ty_beg + 64 j + i, the ballots live in SGPRs): a per-step flag load would expose a scalar-memory round trip in front of every step.
    unsigned long long rowmask[4] = {~0ull, ~0ull, ~0ull, ~0ull};     // host: tiles_per_chunk <= 256
    if (P.seg_flags) {
        const int32_t* fl = P.seg_flags + (int64_t)n * P.nseg;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ty = ty_beg + 64 * j + lane;
            int any = 0;
            if (ty < ty_end) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int y = 2 * ty + k;
                    if (y < P.H) {
                        const int a = (y * P.W + px0) >> 4, b = (y * P.W + min(px0 + 31, P.W - 1)) >> 4;
                        for (int sg = a; sg <= b; ++sg) any |= fl[sg];
                    }
                }
            }
            rowmask[j] = __ballot(any != 0);
        }
    }
    auto flagged = [&](int ty) {
        const int r = ty - ty_beg;
        const unsigned long long m = (r >> 6) == 0 ? rowmask[0] : (r >> 6) == 1 ? rowmask[1] : (r >> 6) == 2 ? rowmask[2] : rowmask[3];
        return ((m >> (r & 63)) & 1ull) != 0;
    };

    bool primed = false, any_step = false;
    for (int ty = ty_beg; ty < ty_end; ++ty) {
        if (!flagged(ty)) { primed = false; continue; }
        if (!primed) prime(ty);
        primed = any_step = true;
        const int yn = 2 * ty + 2;                                    // first gradient row of the next tile row
        // 8 tile pairs x 16 frequencies.  Behind every MFMA one micro-slot of side work (an MFMA holds the pipe for 64 cycles after a 4-cycle
        // issue): slots 0..5 of a pair read the raw patch of the next pair, slots 6..10 transform it, the 34 row transfers of the next step are
        // spread over pairs 0..6, and pair 7 waits for them, meets the other waves and prepares the next step's first pair.  No branch in the
        // body: after the block's last tile row the same work runs once more on rows nobody reads (transfers of rows outside the image are zeros).
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            f32x2w (&ca)[8] = opa[p & 1];
            f32x2w (&cb)[8] = opb[p & 1];
            f32x2w (&na)[8] = opa[(p + 1) & 1];
            f32x2w (&nb)[8] = opb[(p + 1) & 1];
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa(ca, f), fb(cb, f), acc[f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (p < 7) {
                    if (f < 4) read_x(p + 1, f);
                    else if (f < 6) read_d(p + 1, f - 4);
                    else if (f < 10) xform_x(nb, f - 6);
                    else if (f == 10) xform_d(na);
                    else {                                            // slots 11..15 of pairs 0..6: 35 places for the 34 transfers
                        const int j = p * 5 + (f - 11);
                        if (j < 17) xfer(0, yn + 1, (yn + 2) % GXR, j);
                        else if (j < 34) xfer(1, yn, yn % GDR, j - 17);
                    }
                } else {
                    if (f == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); set_rows(ty + 1); }
                    else if (f < 5) read_x(0, f - 1);
                    else if (f < 7) read_d(0, f - 5);
                    else if (f < 11) xform_x(nb, f - 7);
                    else if (f == 11) xform_d(na);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    if (!any_step) return;                                            // nothing flagged in this block's rows: its partial sums are zero
    // ---- output transform A^T (s s^T (.) M) A and the atomic adds.  C/D layout: col = lane & 31 (input channel), row = (r & 3) + 8 (r >> 2) + 4 h.
    //      The accumulators leave the AGPRs through LDS (ds_write takes an AGPR as its data operand), four rows r at a time, and every lane reads
    //      its own 16 frequencies of a row back into VGPRs.  Doing the arithmetic on the accumulators directly makes the register allocator copy all
    //      16 tuples (256 registers) to VGPRs at the loop's exit and spill the main loop's operands to make room (106 spills, reloads in front of
    //      the MFMAs); with the detour the kernel needs ~110 VGPRs and none.  The ring buffers are free: the last step waited for its transfers.
    const int c = cib + icw * 32 + l32;
    float* dwn = dw + (P.nw > 1 ? (int64_t)n * P.wbs : 0) + (int64_t)c * P.wsc;
    float* stage = lds + wave * 4096 + lane;                          // [16 f][4 r][64 lanes] per wave
#pragma unroll
    for (int rc = 0; rc < 4; ++rc) {
#pragma unroll
        for (int f = 0; f < 16; ++f)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) stage[(f * 4 + rr) * 64] = acc[f][rc * 4 + rr];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = rc * 4 + rr;
            const int m = cob + ocw * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float cc[3][4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float m0 = stage[((0 + b) * 4 + rr) * 64], m1 = 0.5f * stage[((4 + b) * 4 + rr) * 64], m2 = 0.5f * stage[((8 + b) * 4 + rr) * 64],
                            m3 = stage[((12 + b) * 4 + rr) * 64];
                cc[0][b] = m0 + m1 + m2; cc[1][b] = m1 - m2; cc[2][b] = m1 + m2 + m3;
            }
            // no branch around the atomics: rows / columns past the tensor (only in a layer whose channel counts are not multiples of 64) add 0 to dw[0]
            const bool ok = m < P.Mo && c < P.Ci;
            float* q = ok ? dwn + (int64_t)m * P.wsm : dw;
            const float sc = ok ? 1.f : 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float e1 = 0.5f * cc[ky][1], e2 = 0.5f * cc[ky][2];
                atomicAdd(q + (ok ? P.widx[ky * 3 + 0] : 0), sc * (cc[ky][0] + e1 + e2));
                atomicAdd(q + (ok ? P.widx[ky * 3 + 1] : 0), sc * (e1 - e2));
                atomicAdd(q + (ok ? P.widx[ky * 3 + 2] : 0), sc * (e1 + e2 + cc[ky][3]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// =================================================================================================
// Round 6: Winograd F(4x4, 3x3) for the >= 256^2 layers -- forward / dgrad of the super-resolution convolutions, 40 % of the Winograd time.
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A          g: 3x3 taps, d: 6x6 patch, Y: 4x4 outputs; interpolation points 0, +-1, +-2, inf (Lavin & Gray)
//
// 36 multiplications per 4x4 output tile and channel pair instead of 64 with F(2x2, 3x3) and 144 direct: 1.78x fewer MFMAs than wino_conv_kernel,
// still v_mfma_f32_32x32x2_f32 with fp32 operands.  The price is arithmetic in the transforms (coefficients up to 8 instead of +-1: the result differs
// from the direct sum by ~1e-6 of the tensor's range instead of ~2e-7 -- measured per layer in tests/test_hip_conv_gpu.py and stated in DESIGN.md; cuDNN's
// fp32 Winograd for the reference is this algorithm) and 2.25x as much transformed data per input pixel.
//
// One block = 4 x 8 tiles (16 x 32 output pixels) x 64 output channels x 36 frequencies = 73 728 accumulators = 288 per lane (one wave per SIMD).
// A wave owns 32 output channels x the 32 tiles x 18 frequencies (rows a = 3 fh .. 3 fh + 2 of the 6 x 6 frequency grid).  Per slab of 4 input channels:
//   * U (transformed weights [f 36][h 2][oc 64][j 2], channel k = 2 j + h, written in exactly this image by wino4_weight_kernel) and the slab's raw
//     4 x 18 x 34 input window (LDS rows of 36 floats) go global -> LDS directly (LDS-DMA, one resp. two slabs ahead), as in wino_conv_kernel;
//   * a thread transforms HALF of one 6 x 6 patch (three rows of B^T d, then (.) B: 72 fused multiply-adds) and writes 18 values of V [f][h][tile 32][j 2];
//   * every wave runs 36 MFMAs (18 frequencies x 2 k-steps), operands read as one 8-byte LDS fragment per (frequency, operand).
// The output transform A^T M A needs all six rows a of a tile: each wave applies its three rows, the two waves of an output-channel half exchange
// half of their partial 4 x 4 tiles through LDS, and each finishes (noise / bias / activation epilogue, 16-byte stores) two of the four output rows.
// =================================================================================================
constexpr int W4KC = 4;                      // input channels per slab
constexpr int W4U = 36 * 2 * 64 * 2;         // floats of one U slab image in LDS (36 KB)
constexpr int W4V = 36 * 2 * 32 * 2;         // floats of one V slab (18 KB)
constexpr int W4RC = 36;                     // floats per LDS row of the raw window (34 used)
constexpr int W4RAW = 41 * 64;               // floats of one raw window [4][18][36] = 2592, rounded up to whole wave transfers
constexpr int W4LDS = 2 * W4U + 2 * W4V + 2 * W4RAW;     // 32 896 floats = 128.5 KB

// U[n][slab][f][h][ocp][j] = (G g G^T)[f] of channel c = slab*4 + 2j + h, output channel oc (zero rows up to ocp); the transform in double, rounded once
__global__ void __launch_bounds__(256) wino4_weight_kernel(WinoParams P, const float* __restrict__ w, float* __restrict__ U) {
    const int64_t total = (int64_t)P.nw * (P.Ci / W4KC) * 2 * P.ocp;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        const int oc = (int)(g % P.ocp);
        int64_t r = g / P.ocp;
        const int h = (int)(r & 1); r >>= 1;
        const int slab = (int)(r % (P.Ci / W4KC));
        const int n = (int)(r / (P.Ci / W4KC));
        float u[2][36];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double gk[3][3];
#pragma unroll
            for (int t = 0; t < 9; ++t) gk[t / 3][t % 3] = 0.0;
            if (oc < P.Mo) {
                const float* wp = w + (int64_t)n * P.wbs + (int64_t)oc * P.wsm + (int64_t)(slab * W4KC + 2 * j + h) * P.wsc;
#pragma unroll
                for (int t = 0; t < 9; ++t) gk[t / 3][t % 3] = (double)wp[P.widx[t]];       // widx[(dy+1)*3 + (dx+1)]
            }
            double t6[6][3];                                                             // G g
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const double g0 = gk[0][s], g1 = gk[1][s], g2 = gk[2][s];
                t6[0][s] = g0 * 0.25;
                t6[1][s] = -(g0 + g1 + g2) * (1.0 / 6.0);
                t6[2][s] = -(g0 - g1 + g2) * (1.0 / 6.0);
                t6[3][s] = g0 * (1.0 / 24.0) + g1 * (1.0 / 12.0) + g2 * (1.0 / 6.0);
                t6[4][s] = g0 * (1.0 / 24.0) - g1 * (1.0 / 12.0) + g2 * (1.0 / 6.0);
                t6[5][s] = g2;
            }
#pragma unroll
            for (int a = 0; a < 6; ++a) {                                                // (G g) G^T
                const double g0 = t6[a][0], g1 = t6[a][1], g2 = t6[a][2];
                u[j][a * 6 + 0] = (float)(g0 * 0.25);
                u[j][a * 6 + 1] = (float)(-(g0 + g1 + g2) * (1.0 / 6.0));
                u[j][a * 6 + 2] = (float)(-(g0 - g1 + g2) * (1.0 / 6.0));
                u[j][a * 6 + 3] = (float)(g0 * (1.0 / 24.0) + g1 * (1.0 / 12.0) + g2 * (1.0 / 6.0));
                u[j][a * 6 + 4] = (float)(g0 * (1.0 / 24.0) - g1 * (1.0 / 12.0) + g2 * (1.0 / 6.0));
                u[j][a * 6 + 5] = (float)g2;
            }
        }
        float2* dst = reinterpret_cast<float2*>(U + (int64_t)n * P.u_bs) + ((int64_t)(slab * 36) * 2 + h) * P.ocp + oc;
#pragma unroll
        for (int f = 0; f < 36; ++f) dst[(int64_t)f * 2 * P.ocp] = make_float2(u[0][f], u[1][f]);
    }
}

// 1-D input transform B^T (6 -> 6), rows selected at compile time: t = B^T d
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
template <int HALF>
__device__ __forceinline__ void w4_bt_rows(const float (&d)[6], float (&t)[3]) {
    if (HALF == 0) {
        t[0] = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
        const float a = fmaf(-4.f, d[2], d[4]), b = fmaf(-4.f, d[1], d[3]);
        t[1] = a + b; t[2] = a - b;
    } else {
        const float c = d[4] - d[2], e = d[3] - d[1];
        t[0] = fmaf(2.f, e, c); t[1] = fmaf(-2.f, e, c);
        t[2] = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
    }
}
__device__ __forceinline__ void w4_bt_full(const float (&d)[6], float (&t)[6]) {
    t[0] = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
    const float a = fmaf(-4.f, d[2], d[4]), b = fmaf(-4.f, d[1], d[3]);
    t[1] = a + b; t[2] = a - b;
    const float c = d[4] - d[2], e = d[3] - d[1];
    t[3] = fmaf(2.f, e, c); t[4] = fmaf(-2.f, e, c);
    t[5] = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
}

__global__ void __launch_bounds__(256, 1) wino4_conv_kernel(WinoParams P, const float* __restrict__ in, const float* __restrict__ U,
                                                            float* __restrict__ out, WinoEpilogue ep) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // Us[2][W4U] | Vs[2][W4V] | Rs[2][W4RAW]
    float* Us = lds;
    float* Vs = lds + 2 * W4U;
    float* Rs = lds + 2 * W4U + 2 * W4V;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31;
    const int ocw = wave & 1, fh = wave >> 1;                         // MFMA role: 32 output channels, frequency rows a = 3 fh .. 3 fh + 2
    const int n = blockIdx.z, oc0 = blockIdx.y * WOC;
    const int bxi = blockIdx.x % P.bx, byi = blockIdx.x / P.bx;       // (P.bx: 32-pixel block columns)
    const int oy0 = byi * 16, ox0 = bxi * 32;
    const int64_t HW = (int64_t)P.H * P.W;
    float* ob = out + (int64_t)n * P.out_bs;

    // ---- needed-output map (forward) / zero-segment map of the gradient operand (dgrad): nothing flagged in the block's output rows resp.
    //      receptive field -> the result is exactly zero: write it and leave (wino_conv_kernel's rule on a 16 x 32 block)
    if (P.out_flags || P.seg_flags) {
        const int32_t* fl = (P.out_flags ? P.out_flags : P.seg_flags) + (int64_t)n * P.nseg;
        const int halo = P.out_flags ? 0 : 1;
        const int ylo = max(oy0 - halo, 0), yhi = min(oy0 + 15 + halo, P.H - 1);
        const int xlo = max(ox0 - halo, 0), xhi = min(ox0 + 31 + halo, P.W - 1);
        int any = 0;
        const int rows = yhi - ylo + 1;
        for (int e = tid; e < rows * 4; e += 256) {                  // <= 4 segments per row (34 pixels span at most 4 when rows start on a segment boundary or not)
            const int iy = ylo + (e >> 2);
            const int sg = ((iy * P.W + xlo) >> 4) + (e & 3);
            if (sg <= ((iy * P.W + xhi) >> 4)) any |= fl[sg];
        }
        if (!__syncthreads_or(any)) {
            for (int e = tid; e < WOC * 512; e += 256) {
                const int m = oc0 + (e >> 9), yy = oy0 + ((e >> 5) & 15), xx = ox0 + (e & 31);
                if (m < P.Mo) ob[(int64_t)m * HW + (int64_t)yy * P.W + xx] = 0.f;
            }
            return;
        }
    }

    const __amdgpu_buffer_rsrc_t rsI = make_rsrc(in + (int64_t)n * P.in_bs, P.in_bs * 4);
    const int chs4 = __builtin_amdgcn_readfirstlane((int)HW * 4);
    const int nslab = P.Ci / W4KC;
    // raw window image [4 channels][18 rows][36 columns] (columns 34, 35 unused): lane l of wave transfer T writes slot T * 64 + l
    unsigned voffR[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const int slot = (i * 4 + wave) * 64 + lane;
        const int ch = slot / (18 * W4RC), rem = slot - ch * (18 * W4RC), r = rem / W4RC, c = rem - r * W4RC;
        const int iy = oy0 - 1 + r, ix = ox0 - 1 + c;
        voffR[i] = (slot < W4KC * 18 * W4RC && c < 34 && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W) ? (unsigned)(ch * chs4 + (iy * P.W + ix) * 4) : BUF_OOB;
    }
    // U slab image: 72 rows (f, h) of 64 oc x 2 floats = 512 B; wave transfer T (16 B per lane) carries rows 2 T, 2 T + 1
    unsigned uvoff[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int T = i * 4 + wave, row = 2 * T + (lane >> 5);
        uvoff[i] = (unsigned)((((int64_t)row * P.ocp + oc0) * 2) * 4 + (lane & 31) * 16);
    }
    const char* Ubase = reinterpret_cast<const char*>(U + (int64_t)n * P.u_bs);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    auto copy_u = [&](int buf, int s, int i) {
        s = min(s, nslab - 1);
        const char* src = Ubase + (int64_t)s * 72 * P.ocp * 8;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * W4U + (i * 4 + wave) * 256) * 4));
        SPI_LDS_DMA_GLOBAL_X4(dst, uvoff[i], src);
    };
    auto copy_raw = [&](int rbuf, int s, int i) {
        s = min(s, nslab - 1);
        const int soff = __builtin_amdgcn_readfirstlane(s * W4KC * chs4);
        if (i < 10 || wave == 0) {                                   // 41 wave transfers: the last round is wave 0's alone
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((2 * W4U + 2 * W4V + rbuf * W4RAW + (i * 4 + wave) * 64) * 4));
            SPI_LDS_DMA_BUFFER_X1(dst, voffR[i], rsI, soff);
        }
    };

    // ---- transform role: tile = lane & 31 (ty = tile >> 3, tx = tile & 7), channel k = (wave & 1) + 2 (lane >> 5)  [h_ = wave & 1, j_ = lane >> 5],
    //      rows a = 3 half .. 3 half + 2 of the frequency grid, half = wave >> 1
    const int t_h = wave & 1, t_j = lane >> 5, t_half = wave >> 1;
    const int t_k = t_h + 2 * t_j;
    const int ty = l32 >> 3, tx = l32 & 7;
    const int roff = (t_k * 18 + 4 * ty) * W4RC + 4 * tx;            // the patch's first float in the raw window
    auto transform = [&](const float* Rb, float* Vb) {
        float T[3][6];
        // rows of the patch: half 0 needs d rows 0..4, half 1 rows 1..5; column c of (B^T d) from the column c of d
        float d[6][6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if ((t_half == 0 && r == 5) || (t_half == 1 && r == 0)) {          // (wave-uniform)
#pragma unroll
                for (int c = 0; c < 6; ++c) d[r][c] = 0.f;
                continue;
            }
            const float4 lo = *reinterpret_cast<const float4*>(Rb + roff + r * W4RC);
            const float2 hi = *reinterpret_cast<const float2*>(Rb + roff + r * W4RC + 4);
            d[r][0] = lo.x; d[r][1] = lo.y; d[r][2] = lo.z; d[r][3] = lo.w; d[r][4] = hi.x; d[r][5] = hi.y;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const float col[6] = {d[0][c], d[1][c], d[2][c], d[3][c], d[4][c], d[5][c]};
            float t3[3];
            if (t_half == 0) w4_bt_rows<0>(col, t3); else w4_bt_rows<1>(col, t3);
            T[0][c] = t3[0]; T[1][c] = t3[1]; T[2][c] = t3[2];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v[6];
            w4_bt_full(T[a], v);
#pragma unroll
            for (int b = 0; b < 6; ++b) Vb[((((3 * t_half + a) * 6 + b) * 2 + t_h) * 32 + l32) * 2 + t_j] = v[b];
        }
    };

    f32x16 acc[18];
#pragma unroll
    for (int f = 0; f < 18; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // prologue: U[0], raw[0], raw[1] -> LDS; V[0] from raw[0]
#pragma unroll
    for (int i = 0; i < 11; ++i) { if (i < 9) copy_u(0, 0, i); copy_raw(0, 0, i); copy_raw(1, 1, i); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    transform(Rs, Vs);
    __syncthreads();

    const int aoff = ((fh * 18 * 2 + h) * 64 + ocw * 32 + l32) * 2, boff = ((fh * 18 * 2 + h) * 32 + l32) * 2;
    // Hand-interleaved main loop (wino_conv_kernel's scheme): behind EVERY MFMA one micro-slot of side work -- an fp32 MFMA occupies the pipe for 64
    // cycles after a 4-cycle issue, a handful of instructions per slot run in its shadow.  36 slots per slab:
    //   * the operand fragments of the next group of three frequencies (first three slots of a group),
    //   * this iteration's 20 transfers (U[s+1]: 9, raw[s+2]: 11) on slots 0..19, so that the barrier at the end of the slab finds them landed,
    //   * raw[s+1] -> V[s+1]: five patch rows on slots 0..4, the three packed column-pair transforms on slots 6, 8, 10, then per frequency row a the
    //     row transform and its six stores on slots 18 + 6 a .. 23 + 6 a.
    // (A first version without the slots -- transfers, transform, then 36 MFMAs -- ran at F(2x2)'s speed; the steps to 1.3x: transfers early,
    //  two accumulator tiles pinned to VGPRs, packed column transform.  Ablation, same box: MFMAs + fragments alone 237 us, + transform 307, + transfers 315
    //  on 128 -> 128 at 512^2: build with -DW4_NO_DMA / -DW4_NO_XFORM to repeat it.)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 d2[5][3], T2[3][3];                                        // five patch rows / three transformed rows as column PAIRS (v_pk_fma_f32 / v_pk_add_f32)
    float v[6];
    const f32x2 k4 = {4.f, 4.f}, km4 = {-4.f, -4.f}, km5 = {-5.f, -5.f}, k2 = {2.f, 2.f}, km2 = {-2.f, -2.f};
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        const float* Ub = Us + buf * W4U + aoff;
        const float* Vb = Vs + buf * W4V + boff;
        const float* Rn = Rs + (buf ^ 1) * W4RAW;
        float* Vw = Vs + (buf ^ 1) * W4V;
        // MFMA order: frequencies in groups of three, k-step inside the group outermost -- slot m = 6 g + 3 j + i is frequency 3 g + i, k-step j: the two
        // MFMAs of an accumulator are three slots (192 cycles) apart instead of back to back (a dependent MFMA waits for its predecessor's last pass)
        float2 af[2][3], bf[2][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { af[0][i] = *reinterpret_cast<const float2*>(Ub + i * 256); bf[0][i] = *reinterpret_cast<const float2*>(Vb + i * 128); }
#pragma unroll
        for (int m = 0; m < 36; ++m) {
            const int g = m / 6, j = (m % 6) / 3, i3 = m % 3, f = 3 * g + i3;
            const float av = j ? af[g & 1][i3].y : af[g & 1][i3].x, bv = j ? bf[g & 1][i3].y : bf[g & 1][i3].x;
            // 288 accumulators do not fit the 256 AGPRs: left to itself the compiler parks two of the 18 tiles in VGPRs and swaps them through an AGPR
            // tile around their MFMAs (96 v_accvgpr moves per slab).  Frequencies 16 and 17 are pinned to VGPRs instead: the MFMA takes VGPR C / D.
            if (f < 16) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[f], 0, 0, 0);
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[f]) : "v"(av), "v"(bv));
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < 6 && (m % 6) < 3) {                                          // the next group's fragments, one frequency per slot
                af[(g + 1) & 1][i3] = *reinterpret_cast<const float2*>(Ub + (3 * (g + 1) + i3) * 256);
                bf[(g + 1) & 1][i3] = *reinterpret_cast<const float2*>(Vb + (3 * (g + 1) + i3) * 128);
            }
#ifndef W4_NO_DMA
            // all 20 transfers in the first 20 slots: the barrier at the end of the slab waits for them, 16 MFMAs (~1000 cycles) later
            if (m < 20) { if (m & 1) { if ((m >> 1) < 9) copy_u(buf ^ 1, s + 1, m >> 1); } else copy_raw(buf, s + 2, m >> 1); }
            if (m == 19) copy_raw(buf, s + 2, 10);
#endif
#ifndef W4_NO_XFORM
            if (m < 5) {                                                            // patch rows (half 0: rows 0..4, half 1: rows 1..5)
                const int r = m + t_half;
                const float4 lo = *reinterpret_cast<const float4*>(Rn + roff + r * W4RC);
                const float2 hi = *reinterpret_cast<const float2*>(Rn + roff + r * W4RC + 4);
                d2[m][0] = f32x2{lo.x, lo.y}; d2[m][1] = f32x2{lo.z, lo.w}; d2[m][2] = f32x2{hi.x, hi.y};
            }
            if (m == 6 || m == 8 || m == 10) {                                       // columns 2 p, 2 p + 1 of B^T d (three rows of it), packed
                const int p = (m - 6) >> 1;
                if (t_half == 0) {                                                  // d2[0..4] = patch rows 0..4: rows 0, 1, 2 of B^T
                    T2[0][p] = __builtin_elementwise_fma(k4, d2[0][p], __builtin_elementwise_fma(km5, d2[2][p], d2[4][p]));
                    const f32x2 a = __builtin_elementwise_fma(km4, d2[2][p], d2[4][p]), b = __builtin_elementwise_fma(km4, d2[1][p], d2[3][p]);
                    T2[1][p] = a + b; T2[2][p] = a - b;
                } else {                                                            // d2[0..4] = patch rows 1..5: rows 3, 4, 5 of B^T
                    const f32x2 c = d2[3][p] - d2[1][p], e = d2[2][p] - d2[0][p];
                    T2[0][p] = __builtin_elementwise_fma(k2, e, c); T2[1][p] = __builtin_elementwise_fma(km2, e, c);
                    T2[2][p] = __builtin_elementwise_fma(k4, d2[0][p], __builtin_elementwise_fma(km5, d2[2][p], d2[4][p]));
                }
            }
            if (m >= 18) {
                const int a = (m - 18) / 6, ph = (m - 18) % 6;
                if (ph == 0) { const float Ta[6] = {T2[a][0].x, T2[a][0].y, T2[a][1].x, T2[a][1].y, T2[a][2].x, T2[a][2].y}; w4_bt_full(Ta, v); }
                if (ph >= 2 && ph < 5) {
                    const int b0 = 2 * (ph - 2);
                    Vw[((((3 * t_half + a) * 6 + b0) * 2 + t_h) * 32 + l32) * 2 + t_j] = v[b0];
                    Vw[((((3 * t_half + a) * 6 + b0 + 1) * 2 + t_h) * 32 + l32) * 2 + t_j] = v[b0 + 1];
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this iteration's transfers have landed
        __syncthreads();
    }

    // ---- output transform A^T M A + epilogue.  C/D layout: col = lane & 31 (tile), row = (r & 3) + 8 (r >> 2) + 4 h.
    //   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].  Per accumulator row r the wave has M[a][b], a = 3 fh + (0, 1, 2), b = 0..5:
    //   R[a][j] = sum_b M[a][b] A[b][j], then its share of Y[i][j] = sum_a A^T[i][a] R[a][j].  The wave keeps output rows i = 2 fh, 2 fh + 1 and
    //   hands rows 2 (1 - fh), 2 (1 - fh) + 1 to its partner (same ocw, other fh) through LDS: [8 values][4 rows r][64 lanes] per wave and chunk.
    //   (The main loop's buffers are free: its last iteration ended with a barrier.)
    float* xch = lds + wave * 2048 + lane;                            // this wave's outgoing chunk
    const float* xin = lds + (wave ^ 2) * 2048 + lane;                // the partner's
    const int oy = oy0 + 4 * ty + 2 * fh, ox = ox0 + 4 * tx;
    const int64_t pix = (int64_t)oy * P.W + ox;
    float nz[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) nz[i][j] = 0.f;
    if (ep.noise) {
        const float ng = ep.noise_gain ? ep.noise_gain[0] : 1.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* np = ep.noise + pix + (int64_t)i * P.W;          // (scalar loads: a noise map that is a view need not be 16-byte aligned)
            nz[i][0] = np[0] * ng; nz[i][1] = np[1] * ng; nz[i][2] = np[2] * ng; nz[i][3] = np[3] * ng;
        }
    }
    const bool has_epi = ep.act != 0 || ep.bias || ep.noise;
#pragma unroll
    for (int rc = 0; rc < 4; ++rc) {
        float keep[4][2][4];                                          // [r in chunk][output row of mine][column]
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = rc * 4 + rr;
            float R[3][4];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float m0 = acc[a * 6 + 0][r], m1 = acc[a * 6 + 1][r], m2 = acc[a * 6 + 2][r], m3 = acc[a * 6 + 3][r], m4 = acc[a * 6 + 4][r], m5 = acc[a * 6 + 5][r];
                const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                R[a][0] = m0 + s1 + s2; R[a][1] = fmaf(2.f, d2, d1); R[a][2] = fmaf(4.f, s2, s1); R[a][3] = fmaf(8.f, d2, d1) + m5;
            }
            float Y[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (fh == 0) {                                        // a = 0, 1, 2: columns [1 0 0 0], [1 1 1 1], [1 -1 1 -1] of A^T
                    const float s = R[1][j] + R[2][j], dd = R[1][j] - R[2][j];
                    Y[0][j] = R[0][j] + s; Y[1][j] = dd; Y[2][j] = s; Y[3][j] = dd;
                } else {                                              // a = 3, 4, 5: columns [1 2 4 8], [1 -2 4 -8], [0 0 0 1]
                    const float s = R[0][j] + R[1][j], dd = R[0][j] - R[1][j];
                    Y[0][j] = s; Y[1][j] = 2.f * dd; Y[2][j] = 4.f * s; Y[3][j] = fmaf(8.f, dd, R[2][j]);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    keep[rr][i][j] = fh == 0 ? Y[i][j] : Y[2 + i][j];
                    xch[((i * 4 + j) * 4 + rr) * 64] = fh == 0 ? Y[2 + i][j] : Y[i][j];
                }
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = rc * 4 + rr;
            const int m = oc0 + ocw * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float bv = (ep.bias && m < P.Mo) ? ep.bias[m] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = keep[rr][i][j] + xin[((i * 4 + j) * 4 + rr) * 64];
                    if (has_epi) v = conv_act_gain_clamp(ep.act, ep.alpha, ep.gain, ep.clamp, v + nz[i][j] + bv);
                    y[j] = v;
                }
                if (m < P.Mo) *reinterpret_cast<float4*>(ob + (int64_t)m * HW + pix + (int64_t)i * P.W) = make_float4(y[0], y[1], y[2], y[3]);
            }
        }
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------------
int64_t spi_wino_workspace_bytes(const WinoParams& P) { return (int64_t)P.nw * P.u_bs_of() * 4; }

int spi_wino_launch(WinoParams P, const float* in, const float* w, float* out, const WinoEpilogue& ep, void* workspace, hipStream_t st, bool u_ready) {
    float* U = static_cast<float*>(workspace);
    P.u_bs = P.nw > 1 ? P.u_bs_of() : 0;
    if (P.f4) {
        if (!u_ready) {
            const int64_t total = (int64_t)P.nw * (P.Ci / W4KC) * 2 * P.ocp;
            const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
            WinoParams Pw = P; Pw.u_bs = P.u_bs_of();
            hipLaunchKernelGGL(wino4_weight_kernel, dim3(grid), dim3(256), 0, st, Pw, w, U);
        }
        constexpr size_t lds4 = (size_t)W4LDS * sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
        if (e != hipSuccess) { spi_set_error("winograd F(4x4,3x3) conv: cannot reserve %zu bytes of LDS: %s", lds4, hipGetErrorString(e)); return SPI_ERR_LAUNCH; }
        P.bx = P.W / 32; P.by = P.H / 16;
        dim3 grid4((unsigned)(P.bx * P.by), (unsigned)(P.ocp / WOC), (unsigned)P.N);
        hipLaunchKernelGGL(wino4_conv_kernel, grid4, dim3(256), lds4, st, P, in, U, out, ep);
        return SPI_OK;
    }
    if (!u_ready) {           // (u_ready: the workspace still holds the transform of these weights from an earlier call -- frozen weights, spi_conv_desc.workspace_ready)
        const int64_t total = (int64_t)P.nw * (P.Ci / WKC) * 2 * P.ocp;
        const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
        WinoParams Pw = P; Pw.u_bs = P.u_bs_of();
        hipLaunchKernelGGL(wino_weight_kernel, dim3(grid), dim3(256), 0, st, Pw, w, U);
    }
    // The dynamic-LDS limit is a per-DEVICE function attribute: set it on every launch (a host-side table update, legal during stream
    // capture) instead of caching "done" in a process-wide flag, which was wrong for a process that touches a second GPU and racy
    // between threads.
    constexpr size_t lds_bytes = (4 * WSLAB + 2 * WRAW) * sizeof(float);
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) { spi_set_error("winograd conv: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e)); return SPI_ERR_LAUNCH; }
    }
    const int ksplit = P.ksplit > 1 ? P.ksplit : 1;
    dim3 grid((unsigned)(P.bx * P.by), (unsigned)(P.ocp / WOC), (unsigned)(P.N * ksplit));
    hipLaunchKernelGGL(wino_conv_kernel, grid, dim3(256), lds_bytes, st, P, in, U, out, ep);
    return SPI_OK;
}

// Weight-gradient launch.  Returns SPI_OK; the caller has zeroed dw.
int spi_wino_wgrad_launch(const WinoParams& Wp, const float* x, const float* dy, float* dw, hipStream_t st) {
    WinoWgradParams P;
    P.N = Wp.N; P.nw = Wp.nw; P.Mo = Wp.Mo; P.Ci = Wp.Ci; P.H = Wp.H; P.W = Wp.W;
    P.strips = (Wp.W + 31) / 32; P.nci = (Wp.Ci + 63) / 64;
    P.in_bs = Wp.in_bs; P.out_bs = Wp.out_bs; P.wbs = Wp.wbs; P.wsm = Wp.wsm; P.wsc = Wp.wsc;
    for (int t = 0; t < 9; ++t) P.widx[t] = Wp.widx[t];
    P.seg_flags = Wp.seg_flags; P.nseg = Wp.nseg;
    const int tiles_y = (Wp.H + 1) / 2;
    const int cc = ((Wp.Mo + 63) / 64) * P.nci;
    // one block per CU (85 KB of LDS, 512 registers per lane): split the tile rows so that one round of blocks fills the 256 CUs
    const int64_t base = (int64_t)cc * P.strips * Wp.N;
    int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(256 / std::max<int64_t>(base, 1), tiles_y / 4));
    // (masked gradient: cutting the rows finer -- 16 tile rows per block, so that the dispatcher balances the flagged part of the image -- was
    //  measured slower: every extra block pays the 4-row prologue and the 144-atomic epilogue; tools/bench_wgrad_masked.py)
    P.tiles_per_chunk = std::min((tiles_y + chunks - 1) / chunks, 256);          // (the kernel's row classification holds 4 x 64 rows)
    chunks = (tiles_y + P.tiles_per_chunk - 1) / P.tiles_per_chunk;
    constexpr size_t lds_bytes = GLDS * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { spi_set_error("winograd wgrad: cannot reserve %zu bytes of LDS: %s", lds_bytes, hipGetErrorString(e)); return SPI_ERR_LAUNCH; }
    dim3 grid((unsigned)(P.strips * chunks), (unsigned)cc, (unsigned)Wp.N);
    hipLaunchKernelGGL(wino_wgrad_kernel, grid, dim3(256), lds_bytes, st, P, x, dy, dw);
    return SPI_OK;
}
