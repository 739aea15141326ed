// filtered_lrelu in ONE pass (the reference's filtered_lrelu.cu:119-1105 fuses up-FIR, activation and down-FIR in shared memory; until round 3 this
// library ran them as two launches through an upsampled buffer in HBM):
//
//     y = downFIR( clamp( lrelu( upFIR(x + b) * up^2 ) * gain ) )            filtered_lrelu.py:58-118, filtered_lrelu.cpp:20-214
//
// One block produces a TO x TO tile of one output plane.  It loads the input window the tile depends on into LDS (bias added inside the image,
// zeros outside), evaluates the upsampling filter for the (TO - 1) down + fd "mid" samples under the tile's down-filter footprint -- only the taps
// that meet a real sample, like upfirdn2d_kernel --, applies gain / leaky ReLU / clamp and keeps the result in LDS, and sums the down filter from
// there.  The upsampled tensor never exists in memory.  The reference's bit-packed SIGN tensor (2 bits per mid sample: 1 = negative, 2 = clamped;
// [NC, sH, sW / 4] bytes) is written (mode 1: every block stores the bytes of the mid rectangle it owns, zeros where it has no sample) or read
// (mode 2, the gradient pass: slope / 0 / 1 by the sign at (x + sx, y + sy), no clamp), exactly as spi_filtered_lrelu_act does on a
// materialised tensor -- so the differentiable op keeps nothing but the filters and the signs.
// Generic in up, down >= 1, 2-D filters of up to 256 taps, paddings, flip; no global device state (the reference's constant-memory filter
// buffer makes its kernel non-reentrant across streams, filtered_lrelu.cu:81-82).
#include "common.hpp"

namespace {

constexpr int FL_MAX_TAPS = 256;

struct FlParams {
    int C, inH, inW, fuH, fuW, fdH, fdW, up, down, px0, py0, flip, midH, midW, outH, outW;
    float gain, slope, clamp;
    int mode, sH, sW, sx, sy;           // sign tensor: 0 none, 1 write, 2 read
    int TO, tmid, tin;                  // output tile edge; mid / input window edges in LDS
    int tiles_x, tiles_y;
};

__device__ __forceinline__ int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

__global__ void __launch_bounds__(256) flrelu_fused_kernel(const float* __restrict__ x, const float* __restrict__ fu, const float* __restrict__ fd,
                                                          const float* __restrict__ b, uint8_t* __restrict__ signs, float* __restrict__ y, FlParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // fu taps | fd taps | input window [tin][tin] | mid [tmid][tmid] | sign codes (bytes)
    float* sfu = lds;
    float* sfd = sfu + p.fuH * p.fuW;
    float* xin = sfd + p.fdH * p.fdW;
    float* mid = xin + p.tin * p.tin;
    uint8_t* sgn = reinterpret_cast<uint8_t*>(mid + p.tmid * p.tmid);
    const int tid = threadIdx.x;
    const int64_t nc = blockIdx.y;
    const int tx_ = blockIdx.x % p.tiles_x, ty_ = blockIdx.x / p.tiles_x;
    const int ox0 = tx_ * p.TO, oy0 = ty_ * p.TO;
    const int mx0 = ox0 * p.down, my0 = oy0 * p.down;                  // first mid sample of the tile
    // filters, flipped like upfirdn2d_kernel; the up filter carries the up^2 gain
    for (int i = tid; i < p.fuH * p.fuW; i += 256) {
        const int ty = i / p.fuW, tx = i % p.fuW;
        sfu[i] = (p.flip ? fu[ty * p.fuW + tx] : fu[(p.fuH - 1 - ty) * p.fuW + (p.fuW - 1 - tx)]) * (float)(p.up * p.up);
    }
    for (int i = tid; i < p.fdH * p.fdW; i += 256) {
        const int ty = i / p.fdW, tx = i % p.fdW;
        sfd[i] = p.flip ? fd[ty * p.fdW + tx] : fd[(p.fdH - 1 - ty) * p.fdW + (p.fdW - 1 - tx)];
    }
    // input window: sample iy feeds mid row my through tap ty when my - py0 + ty == iy * up
    const int ix0 = floor_div(mx0 - p.px0 + p.up - 1, p.up), iy0 = floor_div(my0 - p.py0 + p.up - 1, p.up);      // ceil((m0 - pad) / up)
    const float* xp = x + nc * (int64_t)p.inH * p.inW;
    const float bv = b ? b[nc % p.C] : 0.f;
    for (int i = tid; i < p.tin * p.tin; i += 256) {
        const int iy = iy0 + i / p.tin, ix = ix0 + i % p.tin;
        xin[i] = (iy >= 0 && iy < p.inH && ix >= 0 && ix < p.inW) ? xp[(int64_t)iy * p.inW + ix] + bv : 0.f;
    }
    __syncthreads();
    // mid samples of the tile: up-FIR, gain, leaky ReLU, clamp (or the saved signs)
    const uint8_t* srd = (p.mode == 2) ? signs + nc * (int64_t)p.sH * (p.sW >> 2) : nullptr;
    for (int i = tid; i < p.tmid * p.tmid; i += 256) {
        const int ly = i / p.tmid, lx = i - ly * p.tmid;
        const int my = my0 + ly, mx = mx0 + lx;
        float v = 0.f;
        uint8_t sg = 0;
        if (my < p.midH && mx < p.midW) {
            const int by = my - p.py0, bx = mx - p.px0;
            int ty0 = (-by) % p.up; if (ty0 < 0) ty0 += p.up;
            int tx0 = (-bx) % p.up; if (tx0 < 0) tx0 += p.up;
            float acc = 0.f;
            for (int ty = ty0; ty < p.fuH; ty += p.up) {
                const int ry = (by + ty) / p.up - iy0;                 // by + ty is a multiple of up here; negative ones fall outside the window
                if (by + ty < 0 || ry < 0 || ry >= p.tin) continue;
                for (int tx = tx0; tx < p.fuW; tx += p.up) {
                    const int rx = (bx + tx) / p.up - ix0;
                    if (bx + tx < 0 || rx < 0 || rx >= p.tin) continue;
                    acc = fmaf(sfu[ty * p.fuW + tx], xin[ry * p.tin + rx], acc);
                }
            }
            v = acc * p.gain;
            if (p.mode == 2) {
                const unsigned ux = (unsigned)(mx + p.sx), uy = (unsigned)(my + p.sy);
                if (ux < (unsigned)p.sW && uy < (unsigned)p.sH) {
                    const unsigned sb = srd[(int64_t)uy * (p.sW >> 2) + (ux >> 2)] >> ((ux & 3) << 1);
                    if (sb & 1) v *= p.slope;
                    if (sb & 2) v = 0.f;
                }
            } else {
                if (v < 0.f) { v *= p.slope; sg = 1; }
                if (fabsf(v) > p.clamp) { v = fminf(fmaxf(v, -p.clamp), p.clamp); sg = 2; }
            }
        }
        mid[i] = v;
        sgn[i] = sg;
    }
    __syncthreads();
    // the sign bytes this block owns: mid columns [mx0, next tile) x rows [my0, next tile); the last tile of a row / column owns the rest
    if (p.mode == 1) {
        uint8_t* swr = signs + nc * (int64_t)p.sH * (p.sW >> 2);
        const int ex = (tx_ == p.tiles_x - 1) ? p.sW : min(mx0 + p.TO * p.down, p.sW);
        const int ey = (ty_ == p.tiles_y - 1) ? p.sH : min(my0 + p.TO * p.down, p.sH);
        const int bx0 = mx0 >> 2, nbx = ((ex + 3) >> 2) - bx0, nby = ey - my0;     // host: TO * down % 4 == 0, so tiles start on byte boundaries
        for (int i = tid; i < nbx * nby; i += 256) {
            const int ry = i / nbx, rb = i - ry * nbx;
            unsigned bits = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lx = (rb << 2) + j;
                if (ry < p.tmid && lx < p.tmid) bits |= (unsigned)sgn[ry * p.tmid + lx] << (j << 1);
            }
            if (nbx > 0 && nby > 0) swr[(int64_t)(my0 + ry) * (p.sW >> 2) + bx0 + rb] = (uint8_t)bits;
        }
    }
    // outputs: down-FIR over the tile's mid samples
    float* yp = y + nc * (int64_t)p.outH * p.outW;
    for (int i = tid; i < p.TO * p.TO; i += 256) {
        const int ly = i / p.TO, lx = i - ly * p.TO;
        const int oy = oy0 + ly, ox = ox0 + lx;
        if (oy >= p.outH || ox >= p.outW) continue;
        const float* m = mid + (ly * p.down) * p.tmid + lx * p.down;
        float acc = 0.f;
        for (int ty = 0; ty < p.fdH; ++ty)
            for (int tx = 0; tx < p.fdW; ++tx) acc = fmaf(sfd[ty * p.fdW + tx], m[ty * p.tmid + tx], acc);
        yp[(int64_t)oy * p.outW + ox] = acc;
    }
}

}  // namespace

extern "C" int spi_filtered_lrelu_fused(const float* x, const float* fu, const float* fd, const float* b, uint8_t* signs, float* y, int N, int C,
                                        int inH, int inW, int fuH, int fuW, int fdH, int fdW, int up, int down, int px0, int px1, int py0, int py1,
                                        float gain, float slope, float clamp, int flip, int mode, int sH, int sW, int sx, int sy, int outH,
                                        int outW, spi_stream_t stream) {
    SPI_REQUIRE(x && fu && fd && y && N > 0 && C > 0, "spi_filtered_lrelu_fused: null tensor");
    SPI_REQUIRE(up >= 1 && down >= 1 && fuH * fuW <= FL_MAX_TAPS && fdH * fdW <= FL_MAX_TAPS && fuH > 0 && fuW > 0 && fdH > 0 && fdW > 0,
                "spi_filtered_lrelu_fused: bad factors / filter too large");
    SPI_REQUIRE(mode >= 0 && mode <= 2, "spi_filtered_lrelu_fused: mode must be 0 (no signs), 1 (write signs) or 2 (read signs)");
    SPI_REQUIRE(mode == 0 || (signs && sH > 0 && sW > 0 && (sW & 3) == 0), "spi_filtered_lrelu_fused: sign tensor [NC, sH, sW/4] needs sW %% 4 == 0");
    const int midH = inH * up + py0 + py1 - fuH + 1, midW = inW * up + px0 + px1 - fuW + 1;
    const int eh = (midH - fdH + down) / down, ew = (midW - fdW + down) / down;
    SPI_REQUIRE(midH > 0 && midW > 0 && outH == eh && outW == ew, "spi_filtered_lrelu_fused: output size must be %dx%d", eh, ew);
    SPI_REQUIRE(mode != 1 || (sH >= midH && sW >= midW), "spi_filtered_lrelu_fused: the written sign tensor must cover the %dx%d upsampled samples", midH, midW);
    SPI_REQUIRE((int64_t)N * C <= 65535, "spi_filtered_lrelu_fused: N * C must be <= 65535 planes per call");
    FlParams p{C, inH, inW, fuH, fuW, fdH, fdW, up, down, px0, py0, flip, midH, midW, outH, outW, gain, slope, clamp < 0.f ? INFINITY : clamp,
               mode, sH, sW, sx, sy, 0, 0, 0, 0, 0};
    // largest tile whose windows fit 48 KB of LDS (no per-device attribute needed); TO * down must be a multiple of 4 (sign bytes)
    size_t lds = 0;
    for (int TO : {32, 16, 8, 4}) {
        const int tmid = (TO - 1) * down + std::max(fdH, fdW);
        const int tin = (tmid + std::max(fuH, fuW) - 2) / up + 2;
        lds = (size_t)(fuH * fuW + fdH * fdW + tin * tin + tmid * tmid) * 4 + (size_t)tmid * tmid;
        p.TO = TO; p.tmid = tmid; p.tin = tin;
        if (lds <= 48 * 1024 && (TO * down) % 4 == 0) break;
    }
    SPI_REQUIRE(lds <= 48 * 1024 && (p.TO * down) % 4 == 0, "spi_filtered_lrelu_fused: filters / factors too large for one pass (%zu bytes of LDS)", lds);
    p.tiles_x = (outW + p.TO - 1) / p.TO; p.tiles_y = (outH + p.TO - 1) / p.TO;
    hipLaunchKernelGGL(flrelu_fused_kernel, dim3((unsigned)(p.tiles_x * p.tiles_y), (unsigned)(N * C)), dim3(256), lds, as_stream(stream), x, fu, fd, b,
                       signs, y, p);
    SPI_LAUNCH_CHECK("spi_filtered_lrelu_fused");
    return SPI_OK;
}
