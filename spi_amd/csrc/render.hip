// Volumetric renderer kernels for gfx950 (wave64): ray sampler, tri-plane gather + OSG decoder
// (forward / backward), hierarchical importance resampling, depth merge-sort, alpha-composite ray
// march (forward / backward).  Reference semantics: eg3d/training/volumetric_rendering/*.py and
// eg3d/training/triplane.py:112-135 (cited per kernel).  All fp32.
#include "common.hpp"

// ------------------------------------------------------------------------------------------------
// RaySampler.forward (ray_sampler.py:24-63): one thread per ray.
// ------------------------------------------------------------------------------------------------
__global__ void ray_sampler_kernel(const float* __restrict__ c2w, const float* __restrict__ K, int N, int res,
                                   float* __restrict__ ro, float* __restrict__ rd) {
    const int M = res * res;
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)N * M) return;
    const int n = (int)(g / M), m = (int)(g % M);
    const int i = m / res, j = m % res;
    const float* A = c2w + n * 16;
    const float* Kn = K + n * 9;
    const float fx = Kn[0], sk = Kn[1], cx = Kn[2], fy = Kn[4], cy = Kn[5];
    const float inv = 1.f / (float)res, half = 0.5f / (float)res;
    const float u = (float)j * inv + half, v = (float)i * inv + half;
    const float x = (u - cx + cy * sk / fy - sk * v / fy) / fx;
    const float y = (v - cy) / fy;
    float w[3], o[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o[r] = A[r * 4 + 3];
        w[r] = A[r * 4 + 0] * x + A[r * 4 + 1] * y + A[r * 4 + 2] + A[r * 4 + 3];
        w[r] -= o[r];
    }
    const float nrm = fmaxf(sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), 1e-12f);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        ro[g * 3 + r] = o[r];
        rd[g * 3 + r] = w[r] / nrm;
    }
}

__global__ void coarse_depths_kernel(const float* __restrict__ xi, int64_t total, int S, float start, float end,
                                     float* __restrict__ out) {
    // torch.linspace(start, end, S) is symmetric: step*k from the start for the first half,
    // end - step*(S-1-k) for the second half (ATen RangeFactories); reproduce that rounding.
    const float step = (end - start) / (float)(S - 1);
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(g % S);
        const float base = (k < S / 2) ? start + step * (float)k : end - step * (float)(S - 1 - k);
        out[g] = base + xi[g] * step;
    }
}

// ------------------------------------------------------------------------------------------------
// NCHW <-> NHWC plane relayout through a 32x33 LDS tile (channels-last makes every bilinear corner
// one 128-byte line when C = 32).
// ------------------------------------------------------------------------------------------------
template <bool TO_NHWC>
__global__ void relayout_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int64_t HW) {
    __shared__ float tile[32][33];
    const int64_t np = blockIdx.z;
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 32 x 8
    if (TO_NHWC) {
        for (int r = ty; r < 32; r += 8) {                          // r: channel, tx: pixel
            const int c = c0 + r; const int64_t p = p0 + tx;
            tile[r][tx] = (c < C && p < HW) ? src[(np * C + c) * HW + p] : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {                          // r: pixel, tx: channel
            const int c = c0 + tx; const int64_t p = p0 + r;
            if (c < C && p < HW) dst[(np * HW + p) * C + c] = tile[tx][r];
        }
    } else {
        for (int r = ty; r < 32; r += 8) {                          // r: pixel, tx: channel
            const int c = c0 + tx; const int64_t p = p0 + r;
            tile[r][tx] = (c < C && p < HW) ? src[(np * HW + p) * C + c] : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {                          // r: channel, tx: pixel
            const int c = c0 + r; const int64_t p = p0 + tx;
            if (c < C && p < HW) dst[(np * C + c) * HW + p] = tile[tx][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Tri-plane gather + OSG decoder.
//   256 threads own a tile of 256 points.  Phase A: 8 lanes per point, each lane one float4 of the
//   32 channels, so every corner fetch is a whole 128-B line (coalesced per 8-lane group);
//   the interpolated, plane-averaged feature goes to LDS (row stride 36 floats: conflict-free b128).
//   Phase B: one point per lane, decoder weights are wave-uniform -> scalar loads + v_fmac with an
//   SGPR operand.  Outputs return through LDS so global stores are full lines again.
// ------------------------------------------------------------------------------------------------
static int g_spi_debug = 0;   // spi_debug_set: profiling experiments only
constexpr int DT = 256;        // points per tile == threads per block
constexpr int FWD_MFMA_GRID = 512;      // persistent blocks of decode_fwd_mfma_kernel: two per CU
constexpr int FS = 36;         // LDS row stride (floats)
constexpr int DEC_IN = 32, DEC_HID = 64, DEC_OUT = 33;

struct DecodeArgs {
    const float* planes; const float* coords; const float* ray_o; const float* ray_d; const float* depths;
    int N; int64_t P; int S; int H; int W; float scale;          // scale = 2 / box_warp
    int out_S; int out_off;                                      // output row mapping (0 = plain)
};

__device__ __forceinline__ void point_xyz(const DecodeArgs& a, int64_t g, int& n, float& x, float& y, float& z) {
    n = (int)(g / a.P);
    const int64_t p = g - (int64_t)n * a.P;
    if (a.ray_o == nullptr) {
        const float* c = a.coords + g * 3;
        x = c[0]; y = c[1]; z = c[2];
    } else {
        const int64_t ray = (int64_t)n * (a.P / a.S) + p / a.S;
        const float t = a.depths[g];
        const float* o = a.ray_o + ray * 3; const float* d = a.ray_d + ray * 3;
        x = o[0] + t * d[0]; y = o[1] + t * d[1]; z = o[2] + t * d[2];
    }
    x *= a.scale; y *= a.scale; z *= a.scale;
}

// Index arithmetic of a 256-point tile WITHOUT per-point 64-bit divisions (there is no integer divide instruction: `g / a.P`, `p / a.S` and
// out_row's `g / a.S` cost ~25 VALU + SALU instructions each, 25 of them per point and tile in the forward kernel = 11 % of its instructions).
// The tile's first point is divided once (block-uniform); a point s in 0..255 of the tile is then  ray = ray0 + q,  sample k = t - q S  with
// t = k0 + s < S + 256 and q = t / S taken as (t * m) >> 20, m = 2^20 / S + 1 (exact while t * S < 2^20: S <= 256 here), and the
// sample index n advances when ray0's offset inside its image wraps -- at most once per tile when an image holds >= 512 rays (P >= 512 in
// the explicit-coordinate mode).  Anything else (tiny launches) takes the general path.
struct TileIndex {
    bool fast;
    int64_t ray0;        // ray (ray mode) of the tile's first point
    int k0, m;           // its sample index; the multiplier for / S
    int n0; int64_t r0, rpi;     // its image, the ray's (point's) offset inside the image, rays (points) per image
};
__device__ __forceinline__ TileIndex tile_index(const DecodeArgs& a, int64_t base) {
    TileIndex t;
    const bool ray_mode = a.ray_o != nullptr;
    t.rpi = ray_mode ? a.P / a.S : a.P;
    t.fast = t.rpi >= 512 && (!ray_mode || (a.S >= 1 && a.S <= 256 && a.P % a.S == 0));
    if (!t.fast) return t;
    const int S = ray_mode ? a.S : 1;
    t.ray0 = base / S;                                   // block-uniform: once per tile
    t.k0 = (int)(base - t.ray0 * S);
    t.m = (int)((1u << 20) / (unsigned)S) + 1;
    t.n0 = (int)(t.ray0 / t.rpi);
    t.r0 = t.ray0 - (int64_t)t.n0 * t.rpi;
    return t;
}
// point s (0..255) of the tile -> image n, ray (ray mode: global ray index; coordinate mode: global point index), sample k
__device__ __forceinline__ void tile_point(const DecodeArgs& a, const TileIndex& t, int64_t base, int s, int& n, int64_t& ray, int& k) {
    if (t.fast) {
        const int S = a.ray_o != nullptr ? a.S : 1;
        const int tt = t.k0 + s;
        const int q = a.ray_o != nullptr ? (int)(((unsigned)tt * (unsigned)t.m) >> 20) : tt;
        k = tt - q * S;
        ray = t.ray0 + q;
        n = t.n0 + ((t.r0 + q >= t.rpi) ? 1 : 0);
    } else {
        const int64_t g = base + s;
        n = (int)(g / a.P);
        const int64_t p = g - (int64_t)n * a.P;
        if (a.ray_o != nullptr) { ray = (int64_t)n * (a.P / a.S) + p / a.S; k = (int)(p % a.S); } else { ray = g; k = 0; }
    }
}
__device__ __forceinline__ void point_xyz_at(const DecodeArgs& a, int64_t g, int n_, int64_t ray, float& x, float& y, float& z) {
    if (a.ray_o == nullptr) {
        const float* c = a.coords + g * 3;
        x = c[0]; y = c[1]; z = c[2];
    } else {
        const float t = a.depths[g];
        const float* o = a.ray_o + ray * 3; const float* d = a.ray_d + ray * 3;
        x = o[0] + t * d[0]; y = o[1] + t * d[1]; z = o[2] + t * d[2];
    }
    x *= a.scale; y *= a.scale; z *= a.scale;
    (void)n_;
}
__device__ __forceinline__ int64_t out_row_at(const DecodeArgs& a, int64_t g, int64_t ray, int k) {
    return a.out_S == 0 ? g : ray * a.out_S + a.out_off + k;
}

__device__ __forceinline__ int64_t out_row(const DecodeArgs& a, int64_t g) {
    if (a.out_S == 0) return g;
    const int64_t ray = g / a.S;
    return ray * a.out_S + a.out_off + (g - ray * a.S);
}

__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }   // C/D row of accumulator register r in half h
constexpr float THIRD = 1.f / 3.f;        // the mean over the three planes (renderer.py:62-64)
struct Corner { int x0, y0; float wx0, wx1, wy0, wy1; };

__device__ __forceinline__ Corner make_corner(float gx, float gy, int W, int H) {
    // grid_sample(align_corners=False): ix = ((gx + 1) * W - 1) / 2  (ATen grid_sampler_unnormalize)
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    const float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    Corner c;
    c.x0 = (int)fx; c.y0 = (int)fy;
    c.wx1 = ix - fx; c.wx0 = (fx + 1.f) - ix;
    c.wy1 = iy - fy; c.wy0 = (fy + 1.f) - iy;
    return c;
}

__device__ __forceinline__ void plane_uv(int pl, float x, float y, float z, float& gx, float& gy) {
    // renderer.py:23-53 with the inverse plane axes: planes are sampled at (x,y), (x,z), (z,x)
    gx = (pl == 2) ? z : x;
    gy = (pl == 0) ? y : ((pl == 1) ? z : x);
}

// Bilinear corner offsets (bytes) into the channels-last planes tensor for the 16-byte piece `sub` of a texel row; corners
// outside the plane get the out-of-range offset, for which a raw buffer load returns zeros -- no predication, no
// exec-masked regions, all 12 loads of a point in flight together.
struct CornerOff { unsigned o00, o01, o10, o11; float w00, w01, w10, w11; };
__device__ __forceinline__ CornerOff corner_offsets(const Corner& c, int W, int H, unsigned plane_byte_base, int sub) {
    const bool x0ok = (c.x0 >= 0) & (c.x0 < W), x1ok = (c.x0 + 1 >= 0) & (c.x0 + 1 < W);
    const bool y0ok = (c.y0 >= 0) & (c.y0 < H), y1ok = (c.y0 + 1 >= 0) & (c.y0 + 1 < H);
    const unsigned o = plane_byte_base + (unsigned)(((c.y0 * W + c.x0) * DEC_IN + sub * 4) * 4);       // wraps harmlessly when unused
    CornerOff r;
    r.o00 = (x0ok & y0ok) ? o : BUF_OOB;
    r.o01 = (x1ok & y0ok) ? o + DEC_IN * 4 : BUF_OOB;
    r.o10 = (x0ok & y1ok) ? o + (unsigned)(W * DEC_IN * 4) : BUF_OOB;
    r.o11 = (x1ok & y1ok) ? o + (unsigned)((W + 1) * DEC_IN * 4) : BUF_OOB;
    r.w00 = c.wx0 * c.wy0; r.w01 = c.wx1 * c.wy0; r.w10 = c.wx0 * c.wy1; r.w11 = c.wx1 * c.wy1;
    return r;
}
__device__ __forceinline__ void corner_accumulate(float4& acc, __amdgpu_buffer_rsrc_t rs, const CornerOff& k) {
    f32x4_t v00, v01, v10, v11;
    buf_load4_f32x4(rs, k.o00, k.o01, k.o10, k.o11, v00, v01, v10, v11);
    acc.x += v00.x * k.w00 + v01.x * k.w01 + v10.x * k.w10 + v11.x * k.w11;
    acc.y += v00.y * k.w00 + v01.y * k.w01 + v10.y * k.w10 + v11.y * k.w11;
    acc.z += v00.z * k.w00 + v01.z * k.w01 + v10.z * k.w10 + v11.z * k.w11;
    acc.w += v00.w * k.w00 + v01.w * k.w01 + v10.w * k.w10 + v11.w * k.w11;
}

// One tri-plane sample: 12 corner rows requested together, combined with the bilinear weights, mean over the planes.
__device__ __forceinline__ float4 gather_point(__amdgpu_buffer_rsrc_t rs, int n, float x, float y, float z, int W, int H, unsigned plane_bytes, int sub) {
    unsigned o[12]; float w[12];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        float gx, gy;
        plane_uv(pl, x, y, z, gx, gy);
        const CornerOff k = corner_offsets(make_corner(gx, gy, W, H), W, H, (unsigned)(n * 3 + pl) * plane_bytes, sub);
        o[4 * pl] = k.o00; o[4 * pl + 1] = k.o01; o[4 * pl + 2] = k.o10; o[4 * pl + 3] = k.o11;
        w[4 * pl] = k.w00; w[4 * pl + 1] = k.w01; w[4 * pl + 2] = k.w10; w[4 * pl + 3] = k.w11;
    }
    f32x4_t v[12];
    buf_load12_f32x4(rs, o, v);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {                       // same summation order as before: per plane, then across planes
        acc.x += v[4 * pl].x * w[4 * pl] + v[4 * pl + 1].x * w[4 * pl + 1] + v[4 * pl + 2].x * w[4 * pl + 2] + v[4 * pl + 3].x * w[4 * pl + 3];
        acc.y += v[4 * pl].y * w[4 * pl] + v[4 * pl + 1].y * w[4 * pl + 1] + v[4 * pl + 2].y * w[4 * pl + 2] + v[4 * pl + 3].y * w[4 * pl + 3];
        acc.z += v[4 * pl].z * w[4 * pl] + v[4 * pl + 1].z * w[4 * pl + 1] + v[4 * pl + 2].z * w[4 * pl + 2] + v[4 * pl + 3].z * w[4 * pl + 3];
        acc.w += v[4 * pl].w * w[4 * pl] + v[4 * pl + 1].w * w[4 * pl + 1] + v[4 * pl + 2].w * w[4 * pl + 2] + v[4 * pl + 3].w * w[4 * pl + 3];
    }
    // (x * (1/3), not x / 3: an IEEE division is ~10 VALU instructions per channel; the two differ by at most one ulp of the plane mean)
    acc.x *= THIRD; acc.y *= THIRD; acc.z *= THIRD; acc.w *= THIRD;
    return acc;
}

// Phase A shared by forward and backward: feat[s][0..31] = mean over planes of the bilinear samples.
// Round 4: the kernel is VALU-issue bound, and the gather's ADDRESS ARITHMETIC was almost half of its vector instructions: the 8 lanes that
// fetch one point's twelve corner rows each recomputed the point's position, three plane projections, twelve corner offsets and twelve
// bilinear weights (~250 instructions per pass, 8 passes per tile) -- the same numbers in 8 lanes.  Now a PRE-PASS with one point per lane
// (64 points per wave-instruction instead of 8) writes the point's 12 offsets + 12 weights into columns 0..23 of the point's OWN feature row
// (the row is dead until the point has been gathered; no extra LDS), and a gather pass reads its point's record back with six broadcast
// ds_read_b128, adds its 16-byte piece offset and issues the twelve loads: ~50 instructions per pass.  The record of pass p + 1 is read before
// pass p waits for its gathers, so the LDS latency hides behind them.  Same offsets, weights and summation order: bit-identical features.
// (The eight lanes of a point sit in one wave, whose LDS operations execute in order: their record reads precede their feature writes.)
// the point's gather record: 12 corner offsets (piece 0 of each texel row; out-of-plane corners / invalid points -> out of range = zeros from the
// buffer load) + 12 bilinear weights -> columns 0..23 of its feature row
__device__ __forceinline__ void write_gather_record(float* frow, int n, float x, float y, float z, int W, int H, unsigned plane_bytes, bool valid) {
    uint4* rec = reinterpret_cast<uint4*>(frow);                     // (row stride 144 B: 16-byte aligned)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        float gx, gy;
        plane_uv(pl, x, y, z, gx, gy);
        const CornerOff c = corner_offsets(make_corner(gx, gy, W, H), W, H, (unsigned)(n * 3 + pl) * plane_bytes, 0);
        rec[pl] = valid ? make_uint4(c.o00, c.o01, c.o10, c.o11) : make_uint4(BUF_OOB, BUF_OOB, BUF_OOB, BUF_OOB);
        reinterpret_cast<float4*>(rec)[3 + pl] = make_float4(c.w00, c.w01, c.w10, c.w11);
    }
}

// the eight gather passes of a 256-point tile over records written by write_gather_record (block-wide barrier in between): 8 lanes per point,
// one 16-byte piece of the 32 channels each; feat[s][0..31] = plane mean of the bilinear samples.  `zero_from`: points s >= zero_from get zeros.
__device__ __forceinline__ void gather_from_records(float* feat, __amdgpu_buffer_rsrc_t rs, int zero_from) {
    const int sub = threadIdx.x & 7, grp = threadIdx.x >> 3;
    const unsigned sub16 = (unsigned)sub * 16u;
    uint4 ro[3]; float4 rw[3];
    auto read_record = [&](int s) {
        const uint4* rec = reinterpret_cast<const uint4*>(feat + s * FS);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { ro[pl] = rec[pl]; rw[pl] = reinterpret_cast<const float4*>(rec)[3 + pl]; }
    };
    read_record(grp);
#pragma unroll
    for (int pass = 0; pass < DT / 32; ++pass) {
        const int s = pass * 32 + grp;
        unsigned o[12]; float w[12];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            o[4 * pl] = ro[pl].x + sub16; o[4 * pl + 1] = ro[pl].y + sub16; o[4 * pl + 2] = ro[pl].z + sub16; o[4 * pl + 3] = ro[pl].w + sub16;      // (out of range stays out of range)
            w[4 * pl] = rw[pl].x; w[4 * pl + 1] = rw[pl].y; w[4 * pl + 2] = rw[pl].z; w[4 * pl + 3] = rw[pl].w;
        }
        f32x4_t v[12];
        buf_load12_f32x4_nowait(rs, o, v);
        if (pass + 1 < DT / 32) read_record(s + 32);             // the next pass's record, while this pass's gathers are in flight (its row is still a record)
        buf_wait_gathers(v);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {                       // same summation order as gather_point: per plane, then across planes
            acc.x += v[4 * pl].x * w[4 * pl] + v[4 * pl + 1].x * w[4 * pl + 1] + v[4 * pl + 2].x * w[4 * pl + 2] + v[4 * pl + 3].x * w[4 * pl + 3];
            acc.y += v[4 * pl].y * w[4 * pl] + v[4 * pl + 1].y * w[4 * pl + 1] + v[4 * pl + 2].y * w[4 * pl + 2] + v[4 * pl + 3].y * w[4 * pl + 3];
            acc.z += v[4 * pl].z * w[4 * pl] + v[4 * pl + 1].z * w[4 * pl + 1] + v[4 * pl + 2].z * w[4 * pl + 2] + v[4 * pl + 3].z * w[4 * pl + 3];
            acc.w += v[4 * pl].w * w[4 * pl] + v[4 * pl + 1].w * w[4 * pl + 1] + v[4 * pl + 2].w * w[4 * pl + 2] + v[4 * pl + 3].w * w[4 * pl + 3];
        }
        acc.x *= THIRD; acc.y *= THIRD; acc.z *= THIRD; acc.w *= THIRD;
        if (s >= zero_from) acc = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(feat + s * FS + sub * 4) = acc;
    }
}

__device__ __forceinline__ void gather_tile(const DecodeArgs& a, int64_t base, int64_t total, float* feat) {
    const int t = threadIdx.x;
    const unsigned plane_bytes = (unsigned)(a.H * a.W * DEC_IN * 4);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.planes, (int64_t)a.N * 3 * plane_bytes);          // host: < 2 GiB
    const int last = (int)min((int64_t)DT - 1, total - 1 - base);      // the tail tile clamps to its last real point
    {
        const TileIndex ti = tile_index(a, base);
        int n; int64_t ray; int k; float x, y, z;
        const int sc = min(t, last);
        tile_point(a, ti, base, sc, n, ray, k);
        point_xyz_at(a, base + sc, n, ray, x, y, z);
        write_gather_record(feat + t * FS, n, x, y, z, a.W, a.H, plane_bytes, true);
    }
    __syncthreads();
    gather_from_records(feat, rs, last + 1);
}

// ------------------------------------------------------------------------------------------------
// sample_from_planes as its own operator (renderer.py:55-65): per-PLANE bilinear features [N,3,P,32] at explicit coordinates --
// what a decoder other than the OSG MLP consumes (ImportanceRenderer accepts any decoder callable, renderer.py:88,142-148).
// 8 lanes per (point, plane), one float4 of the 32 channels each: every corner fetch is a whole 128-B line; out-of-plane corners
// take the buffer's out-of-range offset (hardware zeros == padding_mode 'zeros').  Off the SPI hot path, which uses the fused kernels above.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sample_planes_fwd_kernel(const float* __restrict__ planes, const float* __restrict__ coords, int N, int64_t P,
                                                                int H, int W, float scale, float* __restrict__ out) {
    const int sub = threadIdx.x & 7;
    const int64_t e = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);              // (n, pl, p) flattened, p fastest
    if (e >= (int64_t)N * 3 * P) return;
    const int64_t p = e % P;
    const int pl = (int)((e / P) % 3), n = (int)(e / (3 * P));
    const float* c = coords + ((int64_t)n * P + p) * 3;
    float gx, gy;
    plane_uv(pl, c[0] * scale, c[1] * scale, c[2] * scale, gx, gy);
    const unsigned plane_bytes = (unsigned)(H * W * DEC_IN * 4);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(planes, (int64_t)N * 3 * plane_bytes);
    const CornerOff k = corner_offsets(make_corner(gx, gy, W, H), W, H, (unsigned)(n * 3 + pl) * plane_bytes, sub);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    corner_accumulate(acc, rs, k);
    *reinterpret_cast<float4*>(out + e * DEC_IN + sub * 4) = acc;
}

__global__ void __launch_bounds__(256) sample_planes_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ coords, int N, int64_t P,
                                                                int H, int W, float scale, float* __restrict__ d_planes) {
    const int sub = threadIdx.x & 7;
    const int64_t e = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (e >= (int64_t)N * 3 * P) return;
    const int64_t p = e % P;
    const int pl = (int)((e / P) % 3), n = (int)(e / (3 * P));
    const float* c = coords + ((int64_t)n * P + p) * 3;
    float gx, gy;
    plane_uv(pl, c[0] * scale, c[1] * scale, c[2] * scale, gx, gy);
    const Corner k = make_corner(gx, gy, W, H);
    const float4 d4 = *reinterpret_cast<const float4*>(d_out + e * DEC_IN + sub * 4);
    float* pb = d_planes + ((int64_t)(n * 3 + pl) * H * W) * DEC_IN + sub * 4;
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
            const int xx = k.x0 + cx, yy = k.y0 + cy;
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            const float w = (cx ? k.wx1 : k.wx0) * (cy ? k.wy1 : k.wy0);
            float* q = pb + ((int64_t)yy * W + xx) * DEC_IN;
            atomicAdd(q + 0, d4.x * w); atomicAdd(q + 1, d4.y * w); atomicAdd(q + 2, d4.z * w); atomicAdd(q + 3, d4.w * w);
        }
}

// Decoder MLP.  Both layers walk ROWS of 64 contiguous weights (w1t = W1^T [32][64], w2 [33][64]) in a
// runtime loop: the row is wave-uniform -> 4 x s_load_dwordx16 + 64 v_fmac with an SGPR operand;
// the per-point vector that is indexed by the loop variable lives in LDS, the 64 accumulators in
// registers.  (A fully unrolled version makes the compiler hoist all 4160 weights into SGPRs and
// spill them through v_writelane/v_readlane.)
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Packed fp32 (v_pk_fma_f32: two FMAs per lane and cycle -- the 157 TFLOP/s vector peak assumes it): hidden units are processed in
// pairs, the weight pair comes from SGPRs, the per-point input is broadcast into both halves.
__device__ __forceinline__ void layer1_forward(const float* __restrict__ w1t, const float* __restrict__ b1,
                                               const float* frow, float (&h)[DEC_HID]) {
    f32x2_t h2[DEC_HID / 2];
    const f32x2_t* b2v = reinterpret_cast<const f32x2_t*>(b1);
#pragma unroll
    for (int j = 0; j < DEC_HID / 2; ++j) h2[j] = b2v[j];
#pragma unroll 2
    for (int i = 0; i < DEC_IN; ++i) {
        const float fi = frow[i];
        const f32x2_t f2 = {fi, fi};
        const f32x2_t* wr = reinterpret_cast<const f32x2_t*>(w1t + i * DEC_HID);
#pragma unroll
        for (int j = 0; j < DEC_HID / 2; ++j) h2[j] = __builtin_elementwise_fma(wr[j], f2, h2[j]);
    }
#pragma unroll
    for (int j = 0; j < DEC_HID / 2; ++j) { h[2 * j] = softplus_fast(h2[j].x); h[2 * j + 1] = softplus_fast(h2[j].y); }
}

__global__ void __launch_bounds__(DT) decode_fwd_kernel(DecodeArgs a, const float* __restrict__ w1t, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ rgb, float* __restrict__ sigma) {
    __shared__ __attribute__((aligned(16))) float feat[DT * FS];
    const int64_t total = (int64_t)a.N * a.P;
    const int64_t base = (int64_t)blockIdx.x * DT;
    gather_tile(a, base, total, feat);
    __syncthreads();
    const int t = threadIdx.x;
    const TileIndex ti_out = tile_index(a, base);
    float* frow = feat + t * FS;
    float h[DEC_HID];
    layer1_forward(w1t, b1, frow, h);
    // layer 2, one output per iteration; the rgb row overwrites this thread's own feature row.  rgb == NULL (depth-only rendering):
    // only the density row of W2 is evaluated and nothing but sigma is written.
    const int nout = rgb ? DEC_OUT : 1;
#pragma unroll 2
    for (int o = 0; o < nout; ++o) {
        const f32x2_t* wr = reinterpret_cast<const f32x2_t*>(w2 + o * DEC_HID);
        f32x2_t acc2 = {b2[o], 0.f};
#pragma unroll
        for (int j = 0; j < DEC_HID / 2; ++j) acc2 = __builtin_elementwise_fma(wr[j], f32x2_t{h[2 * j], h[2 * j + 1]}, acc2);
        const float acc = acc2.x + acc2.y;
        if (o == 0) {
            if (base + t < total) {
                int nn; int64_t ray; int k;
                tile_point(a, ti_out, base, t, nn, ray, k);
                sigma[out_row_at(a, base + t, ray, k)] = acc;
            }
        }
        else frow[o - 1] = sigmoid_fast(acc) * 1.002f - 0.001f;
    }
    if (!rgb) return;
    __syncthreads();
    // the tile's rgb block is contiguous in global memory: 256 points x 32 floats
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = it * DT + t;                 // float4 index within the tile
        const int s = e >> 3, q = e & 7;
        if (base + s < total) {
            int nn; int64_t ray; int k;
            tile_point(a, ti_out, base, s, nn, ray, k);
            __builtin_nontemporal_store(*reinterpret_cast<const f32x4_t*>(feat + s * FS + q * 4),
                                        reinterpret_cast<f32x4_t*>(rgb + out_row_at(a, base + s, ray, k) * DEC_IN + q * 4));    // 400 MB per image, read back once
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Decoder forward on the matrix cores (round 6): fp32 operands cut into three bf16 pieces, six piece products on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//   decode_fwd_kernel above is vector-issue bound: ~1040 packed FMAs + ~460 other vector instructions per 32 points and wave,
//   0.39 of the vector peak and 0.33 of the L2 gather rate at the same time (profiles/r05*, roofline_gather).  An fp32 MFMA does
//   not help on gfx950 (v_mfma_f32_32x32x2_f32 shares the FP32 lanes with the VALU, 64 cycles for 4096 FLOP: the 96 MFMAs of a
//   32-point tile are as long as the packed FMAs).  The bf16 pipe is 16x faster per instruction: a truncating three-way split
//   x = x0 + x1 + x2 is EXACT for an fp32 number (3 x 8 mantissa bits), and the six products x0 y0, x0 y1, x1 y0, x0 y2, x1 y1,
//   x2 y0 carry everything above 2^-24 |x y| -- the arithmetic is fp32's, not bf16's (conv.hip's split_bf16, NS = 3; same
//   scheme as its `bf16x6` mode).  Per 32 points and wave: 48 MFMAs (32 cycles each) + ~270 vector instructions for the
//   splits instead of ~1040 FMAs.
//   Orientation: D[feature][point] -- the lane owns a point, the accumulator registers its features.  The accumulator layout
//   of layer 1 (register r of half h = hidden row rowmap(r, h)) IS a valid B-operand layout of layer 2 once the reduction axis
//   of W2 is enumerated in the same order (virtual k = 8 registers of a half): no transposition between the layers.
//   The weights are split once per block into REGISTERS (24 fragments x 4 dwords; the kernel is persistent, two blocks per
//   CU): the main loop reads nothing but its own points' 128-byte feature rows from LDS.
//   The density row of W2 (one output) is a 32-term dot product per lane + one cross-half add on the vector ALUs.
// ------------------------------------------------------------------------------------------------
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// eight fp32 -> three packed bf16x8 pieces (bit patterns; truncation: v == p0 + p1 + p2 exactly)
__device__ __forceinline__ void split3_bf16x8(const float (&v)[8], u32x4_t (&p)[3]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = v[2 * i], x1 = v[2 * i + 1];
        p[0][i] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
        const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u), r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
        p[1][i] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
        const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        p[2][i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
}
#define SPI_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, (a)), __builtin_bit_cast(bf16x8_t, (b)), (c), 0, 0, 0)
// the six significant piece products of one K = 16 step, small terms first (conv.hip's mfma_split<3>)
__device__ __forceinline__ f32x16_t mfma_split6(const u32x4_t (&a)[3], const u32x4_t (&b)[3], f32x16_t acc) {
    acc = SPI_MFMA_BF16(a[2], b[0], acc);
    acc = SPI_MFMA_BF16(a[0], b[2], acc);
    acc = SPI_MFMA_BF16(a[1], b[1], acc);
    acc = SPI_MFMA_BF16(a[1], b[0], acc);
    acc = SPI_MFMA_BF16(a[0], b[1], acc);
    acc = SPI_MFMA_BF16(a[0], b[0], acc);
    return acc;
}

__global__ void __launch_bounds__(DT, 2) decode_fwd_mfma_kernel(DecodeArgs a, const float* __restrict__ w1t, const float* __restrict__ b1,
                                                                const float* __restrict__ w2, const float* __restrict__ b2,
                                                                float* __restrict__ rgb, float* __restrict__ sigma, int64_t tiles) {
    __shared__ __attribute__((aligned(16))) float feat[DT * FS];          // rows [point][36]: features 0..31 | out row 32
    __shared__ __attribute__((aligned(16))) float tab[2][80];             // per half h: b1[32 mt + rowmap(r, h)] (32) | W2[0][same] (32) | b2[1 + rowmap(r, h)] (16)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane & 31, hh = lane >> 5;
    const int64_t total = (int64_t)a.N * a.P;
    if (t < 160) {
        const int h = t / 80, e = t - h * 80;
        float v;
        if (e < 32) v = b1[(e >> 4) * 32 + rowmap(e & 15, h)];
        else if (e < 64) v = w2[((e - 32) >> 4) * 32 + rowmap(e & 15, h)];
        else v = b2[1 + rowmap(e - 64, h)];
        tab[h][e] = v;
    }
    // ---- the block's weight fragments, split once, in registers for the whole kernel
    //  A1[mt][s]: W1[32 mt + q][16 s + 8 hh + e]                          (reduction axis = input channel)
    //  A2[s]    : W2[1 + q][16 s + 8 (e >> 2) + 4 hh + (e & 3)]           (reduction axis = hidden unit, in layer 1's accumulator order)
    u32x4_t A1[2][2][3], A2[4][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = w1t[(16 * s + 8 * hh + e) * DEC_HID + 32 * mt + q];
            split3_bf16x8(v, A1[mt][s]);
        }
    if (rgb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = w2[(1 + q) * DEC_HID + 16 * s + 8 * (e >> 2) + 4 * hh + (e & 3)];
            split3_bf16x8(v, A2[s]);
        }
    }
    const float b2s = b2[0];
    const unsigned plane_bytes = (unsigned)(a.H * a.W * DEC_IN * 4);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.planes, (int64_t)a.N * 3 * plane_bytes);          // host: < 2 GiB
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t base = tile * DT;
        __syncthreads();                                               // the previous tile's rows are no longer in use (first pass: tab is written)
        // ---- phase A: this thread's point -> gather record + output row; then the eight gather passes (gather_tile's two steps)
        const int last = (int)min((int64_t)DT - 1, total - 1 - base);
        {
            const TileIndex ti = tile_index(a, base);
            int n; int64_t ray; int k; float x, y, z;
            const int sc = min(t, last);
            tile_point(a, ti, base, sc, n, ray, k);
            point_xyz_at(a, base + sc, n, ray, x, y, z);
            write_gather_record(feat + t * FS, n, x, y, z, a.W, a.H, plane_bytes, true);
            feat[t * FS + 32] = __int_as_float(t <= last ? (int)out_row_at(a, base + t, ray, k) : -1);       // (host: rows < 2^31)
        }
        __syncthreads();
        gather_from_records(feat, rs, last + 1);
        __syncthreads();
        // ---- phase B: 32 points per pass and wave, everything below touches only this wave's 64 rows
#pragma unroll 1
        for (int nt = 0; nt < 2; ++nt) {
            float* frow = feat + (wave * 64 + nt * 32 + q) * FS;
            const float* tb = tab[hh];
            f32x16_t H1[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(tb + mt * 16 + 4 * g);
                    H1[mt][4 * g] = bv.x; H1[mt][4 * g + 1] = bv.y; H1[mt][4 * g + 2] = bv.z; H1[mt][4 * g + 3] = bv.w;
                }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float4 f0 = *reinterpret_cast<const float4*>(frow + 16 * s + 8 * hh), f1 = *reinterpret_cast<const float4*>(frow + 16 * s + 8 * hh + 4);
                const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                u32x4_t B[3];
                split3_bf16x8(v, B);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) H1[mt] = mfma_split6(A1[mt][s], B, H1[mt]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) H1[mt][r] = softplus_fast(H1[mt][r]);
            // density: the lane's 32 hidden units against the density row of W2, the other 32 from the lane's partner in the other half
            float sg = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 wv = *reinterpret_cast<const float4*>(tb + 32 + mt * 16 + 4 * g);
                    sg = fmaf(wv.x, H1[mt][4 * g], sg); sg = fmaf(wv.y, H1[mt][4 * g + 1], sg);
                    sg = fmaf(wv.z, H1[mt][4 * g + 2], sg); sg = fmaf(wv.w, H1[mt][4 * g + 3], sg);
                }
            sg += __shfl_xor(sg, 32, WAVE);
            const int orow = __float_as_int(frow[32]);
            if (hh == 0 && orow >= 0) sigma[orow] = sg + b2s;
            if (rgb) {
                f32x16_t Y;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(tb + 64 + 4 * g);
                    Y[4 * g] = bv.x; Y[4 * g + 1] = bv.y; Y[4 * g + 2] = bv.z; Y[4 * g + 3] = bv.w;
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = H1[s >> 1][8 * (s & 1) + e];
                    u32x4_t B[3];
                    split3_bf16x8(v, B);
                    Y = mfma_split6(A2[s], B, Y);
                }
                // colour channel rowmap(r, hh) = (r & 3) + 8 (r >> 2) + 4 hh: four 16-byte pieces of the point's own row (its features are dead)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(frow + 8 * g + 4 * hh) =
                        make_float4(sigmoid_fast(Y[4 * g]) * 1.002f - 0.001f, sigmoid_fast(Y[4 * g + 1]) * 1.002f - 0.001f,
                                    sigmoid_fast(Y[4 * g + 2]) * 1.002f - 0.001f, sigmoid_fast(Y[4 * g + 3]) * 1.002f - 0.001f);
            }
        }
        if (rgb) {
            // the wave's 64 colour rows leave as whole 128-byte rows (8 lanes per row); a wave's LDS operations complete in order: no barrier
            asm volatile("" ::: "memory");
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int sp = wave * 64 + it * 8 + (lane >> 3), q4 = lane & 7;
                const int orow = __float_as_int(feat[sp * FS + 32]);
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(feat + sp * FS + q4 * 4);
                if (orow >= 0) __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(rgb + (int64_t)orow * DEC_IN + q4 * 4));
            }
        }
    }
}

__global__ void __launch_bounds__(DT) decode_bwd_kernel(DecodeArgs a, const float* __restrict__ w1t, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        const float* __restrict__ d_rgb, const float* __restrict__ d_sigma,
                                                        float* __restrict__ d_planes, float* __restrict__ dump) {
    __shared__ __attribute__((aligned(16))) float feat[DT * FS];
    __shared__ __attribute__((aligned(16))) float gbuf[DT * FS];
    const int64_t total = (int64_t)a.N * a.P;
    const int64_t base = (int64_t)blockIdx.x * DT;
    const int t = threadIdx.x;
    // stage the tile's d_rgb rows (contiguous 32 KB) while the gathers are in flight
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = it * DT + t;
        const int s = e >> 3, q = e & 7;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (base + s < total) v = *reinterpret_cast<const float4*>(d_rgb + out_row(a, base + s) * DEC_IN + q * 4);
        *reinterpret_cast<float4*>(gbuf + s * FS + q * 4) = v;
    }
    gather_tile(a, base, total, feat);
    __syncthreads();
    const bool valid = base + t < total;
    const bool dumping = (dump != nullptr) && valid;
    float* frow = feat + t * FS;
    float* grow = gbuf + t * FS;
    float h[DEC_HID];
    layer1_forward(w1t, b1, frow, h);
    // dump layout: [193][total] (rows: f 0..31, h 32..95, d_pre1 96..159, d_y 160..192), coalesced per row
    if (dumping) {
        for (int i = 0; i < DEC_IN; ++i) dump[(int64_t)i * total + base + t] = frow[i];
#pragma unroll
        for (int j = 0; j < DEC_HID; ++j) dump[(int64_t)(32 + j) * total + base + t] = h[j];
    }
    // layer 2 forward + backward in one sweep over its rows: y_o -> d_y_o -> dp[j] += W2[o][j] d_y_o
    float dp[DEC_HID];
#pragma unroll
    for (int j = 0; j < DEC_HID; ++j) dp[j] = 0.f;
    const float dsig = valid ? d_sigma[out_row(a, base + t)] : 0.f;
#pragma unroll 2
    for (int o = 0; o < DEC_OUT; ++o) {
        const float* wr = w2 + o * DEC_HID;
        float acc = b2[o];
#pragma unroll
        for (int j = 0; j < DEC_HID; ++j) acc = fmaf(wr[j], h[j], acc);
        float dyo;
        if (o == 0) dyo = dsig;                                  // sigma = y[0]
        else { const float sg = sigmoid_fast(acc); dyo = grow[o - 1] * 1.002f * sg * (1.f - sg); }   // rgb = sigmoid*1.002-0.001
        if (dumping) dump[(int64_t)(160 + o) * total + base + t] = dyo;
#pragma unroll
        for (int j = 0; j < DEC_HID; ++j) dp[j] = fmaf(wr[j], dyo, dp[j]);
    }
    // through the softplus: softplus'(pre) = sigmoid(pre) = 1 - exp(-h)
#pragma unroll
    for (int j = 0; j < DEC_HID; ++j) dp[j] *= (h[j] > 20.f) ? 1.f : (1.f - exp_fast(-h[j]));
    if (dumping) {
#pragma unroll
        for (int j = 0; j < DEC_HID; ++j) dump[(int64_t)(96 + j) * total + base + t] = dp[j];
    }
    // d_f[i] = sum_j W1[j][i] d_pre1[j]; the plane mean contributes the 1/3
#pragma unroll 2
    for (int i = 0; i < DEC_IN; ++i) {
        const float* wr = w1t + i * DEC_HID;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < DEC_HID; ++j) acc = fmaf(wr[j], dp[j], acc);
        grow[i] = acc * THIRD;
    }
    __syncthreads();
    // scatter-add into the channels-last plane gradient: 8 lanes per point, one float4 of channels each
    const int sub = t & 7, grp = t >> 3;
    for (int pass = 0; pass < DT / 32; ++pass) {
        const int s = pass * 32 + grp;
        const int64_t g = base + s;
        if (g >= total) continue;
        const float4 d4 = *reinterpret_cast<const float4*>(gbuf + s * FS + sub * 4);
        int n; float x, y3, z;
        point_xyz(a, g, n, x, y3, z);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            float gx, gy;
            plane_uv(pl, x, y3, z, gx, gy);
            const Corner c = make_corner(gx, gy, a.W, a.H);
            float* pb = d_planes + ((int64_t)(n * 3 + pl) * a.H * a.W) * DEC_IN + sub * 4;
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) {
#pragma unroll
                for (int cx = 0; cx < 2; ++cx) {
                    const int xx = c.x0 + cx, yy = c.y0 + cy;
                    if (xx < 0 || xx >= a.W || yy < 0 || yy >= a.H) continue;
                    const float w = (cx ? c.wx1 : c.wx0) * (cy ? c.wy1 : c.wy0);
                    float* p = pb + ((int64_t)yy * a.W + xx) * DEC_IN;
                    atomicAdd(p + 0, d4.x * w); atomicAdd(p + 1, d4.y * w);
                    atomicAdd(p + 2, d4.z * w); atomicAdd(p + 3, d4.w * w);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of gather + decoder for the render path: tiled for scatter locality, decoder on the matrix
// cores, weight gradients fused in.
//   A tile is an 8x8 patch of neighbouring rays x 4 consecutive SORTED sample positions: those 256
//   points land on a few dozen texels per plane, so their plane-gradient contributions are first
//   summed in a 12x12-texel LDS window and only the window's non-zero texels go to HBM with global
//   atomics.  Corners outside the window fall back to direct global atomics, so the result never
//   depends on the window placement.  Sample (ray, k) is read through the sort permutation.
//
//   Phase B (the 32-64-33 MLP, forward recompute + backward) runs on v_mfma_f32_32x32x2_f32 with the
//   POINTS along the MFMA N axis ("orientation 1": accumulator register r of lane (q,h) holds
//   D[feature rowmap(r,h)][point q]).  In that layout the output of one layer is directly the B
//   operand of the next (the K index is walked in the accumulator's own (register, half) order and the
//   weight fragments are pre-permuted to match), so activations never leave registers.
//   The weight gradients contract over POINTS, which needs activations with the feature in the lane
//   position ("orientation 2").  Instead of transposing through LDS (no room next to the 72 KB of
//   feature / gradient rows at 2 blocks per CU) the needed activations are simply computed a second
//   time in orientation 2 -- H2 = F W1^T, Y2 = H1 W2^T and dH2 = dY1 W2 take their A operand
//   straight from orientation-1 registers or LDS rows -- and then dW1 += dpre2^T F, dW2 += dY2^T H2 are
//   MFMAs whose operands are all in registers.  292 MFMAs per 32 points (130 with a frozen decoder)
//   instead of 8320 VALU FMAs per point + a 772 B/point activation dump + a separate reduction kernel.
//   The kernel is persistent (grid <= 512): per-wave weight-gradient accumulators live in registers
//   across tiles and are written once to a scratch row that decoder_partial_reduce_kernel sums.
// ------------------------------------------------------------------------------------------------
constexpr int WIN = 16;                       // window edge in texels; WIN*WIN*32 int32 accumulators <= DT*FS floats of LDS

// weight fragments (built by decoder_frag_kernel for every call), round 6: bf16 PIECES.  A fragment = one A operand of
// v_mfma_f32_32x32x16_bf16 = 16 bytes per lane (8 consecutive virtual-k values of row q, k-group h), three pieces per operand
// (split3_bf16x8: the truncating three-way split is exact for fp32), 64 lanes x uint4 = 256 dwords each:
//   A1 [mt 2][s 2][piece 3]   W1[32 mt + q][16 s + 8 h + e]                                   (H1 = W1 F:          M = hidden, K = channel)
//   A3 [mt 2][s 2][piece 3]   W2[1 + rowmap(8 s + e, h)][32 mt + q]                            (dH1 = W2^T dY:      M = hidden, K = colour, in dY's register order)
//   A4 [s 4][piece 3]         W1[32 (s >> 1) + rowmap(8 (s & 1) + e, h)][q]                     (dF = W1^T dpre1:    M = channel, K = hidden, in dpre1's register order)
// followed by the density row of W2 in the accumulator order of a half: wsig[h][16 mt + r] = W2[0][32 mt + rowmap(r, h)].
constexpr int FRAG_A1 = 0, FRAG_A3 = 12, FRAG_A4 = 24, FRAG_N = 36;       // fragment indices
constexpr int FRAG_WSIG = FRAG_N * 256;                                    // dword offset of wsig [2][32]
constexpr int FRAG_TOTAL = FRAG_WSIG + 64;                                 // 9280 dwords
constexpr int PART_ROW = 4352;                      // scratch row per wave: dW1 2048 | dW2 2112 | db1 64 | db2 33 | pad
constexpr int PART_DW2 = 2048, PART_DB1 = 4160, PART_DB2 = 4224;
constexpr int BWD_MAX_GRID = 512;

__global__ void decoder_frag_kernel(const float* __restrict__ w1t, const float* __restrict__ w2, float* __restrict__ frag) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 12 * 64) {
        const int grp = idx >> 6, lane = idx & 63, q = lane & 31, h = lane >> 5;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (grp < 4) { const int mt = grp >> 1, s2 = grp & 1; v[e] = w1t[(16 * s2 + 8 * h + e) * DEC_HID + 32 * mt + q]; }             // W1[j][i] = w1t[i][j]
            else if (grp < 8) { const int mt = (grp - 4) >> 1, s2 = grp & 1; v[e] = w2[(1 + rowmap(8 * s2 + e, h)) * DEC_HID + 32 * mt + q]; }
            else { const int s4 = grp - 8; v[e] = w1t[q * DEC_HID + 32 * (s4 >> 1) + rowmap(8 * (s4 & 1) + e, h)]; }
        }
        u32x4_t pc[3];
        split3_bf16x8(v, pc);
        u32x4_t* out = reinterpret_cast<u32x4_t*>(frag) + (grp * 3) * 64 + lane;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c * 64] = pc[c];
    } else if (idx < 12 * 64 + 64) {
        const int e = idx - 12 * 64, h = e >> 5, mr = e & 31;
        frag[FRAG_WSIG + e] = w2[(mr >> 4) * 32 + rowmap(mr & 15, h)];
    }
}

// sum the per-wave scratch rows -> dw1 [64,32], db1 [64], dw2 [33,64], db2 [33] (pre-zeroed): blockIdx.y owns a slice of rows
__global__ void decoder_partial_reduce_kernel(const float* __restrict__ part, int rows, float* __restrict__ dw1, float* __restrict__ db1,
                                              float* __restrict__ dw2, float* __restrict__ db2) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= PART_DB2 + 33) return;
    const int per = (rows + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(r0 + per, rows);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += part[(int64_t)r * PART_ROW + e]; s1 += part[(int64_t)(r + 1) * PART_ROW + e];
        s2 += part[(int64_t)(r + 2) * PART_ROW + e]; s3 += part[(int64_t)(r + 3) * PART_ROW + e];
    }
    for (; r < r1; ++r) s0 += part[(int64_t)r * PART_ROW + e];
    const float v = (s0 + s1) + (s2 + s3);
    if (e < PART_DW2) atomicAdd(dw1 + e, v);
    else if (e < PART_DB1) atomicAdd(dw2 + e - PART_DW2, v);
    else if (e < PART_DB1 + 64) atomicAdd(db1 + e - PART_DB1, v);
    else if (e >= PART_DB2) atomicAdd(db2 + e - PART_DB2, v);
}

// ------------------------------------------------------------------------------------------------
// Depth-binned point order for the tiled decoder backward.
//   A tile of decode_bwd_tiled_kernel scatters its 256 points into one 16 x 16-texel LDS window per plane.  Taking the tile
//   as "8 x 8 rays x 4 consecutive SORTED SAMPLE INDICES" makes the XY footprint compact but not the depth extent: once the
//   importance samples cluster, the k-th sorted sample of neighbouring rays sits at different depths and 18 % of the
//   points left the XZ / ZX windows (straight to global atomics).  Here the 64 x S points of a ray patch are ordered by
//   DEPTH instead: NBIN uniform depth bins over the patch's own range, bin-major, ray-minor, sample order inside a ray (each
//   ray's samples are already sorted, so a bin is one contiguous run per ray -> rank = prefix over (bin, ray) counts +
//   offset inside the run: no sort, deterministic).  256 consecutive entries of that order are one tile: a slab of the
//   view frustum a few texels thick.  Rays whose gradient is exactly zero (ray_active == 0) are dropped here, so the
//   masked pseudo-view branches only pay for the rays that carry a gradient.
//   order[patch][pos] = ray-in-patch << 8 | sorted sample index (S <= 256), count[patch] = live points.
// ------------------------------------------------------------------------------------------------
constexpr int NBIN = 64;
constexpr int MAXBINS_S = 256;          // samples per ray the binning handles (the order entries keep the sample index in 8 bits)

__device__ __forceinline__ int patch_ray(int patch, int rl, int ray_w, int patch2d) {
    if (patch2d) {
        const int pw = ray_w >> 3;
        const int py = patch / pw, px = patch - py * pw;
        return ((py << 3) + (rl >> 3)) * ray_w + (px << 3) + (rl & 7);
    }
    return patch * 64 + rl;
}

__global__ void __launch_bounds__(256) bin_points_kernel(const float* __restrict__ depths, const int32_t* __restrict__ ray_active, int M, int S,
                                                         int ray_w, int patch2d, int patches, uint16_t* __restrict__ order,
                                                         int32_t* __restrict__ count) {
    __shared__ int cnt[NBIN * 64];                // live points of (bin, ray); then its exclusive prefix
    __shared__ int first[NBIN * 64];              // smallest sample index of that run
    __shared__ float red[2][4];
    __shared__ int wsum_s[4];
    __shared__ int8_t live[64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = blockIdx.x / patches, patch = blockIdx.x - n * patches;
    bool mine = false;
    if (t < 64) {
        const int m = patch_ray(patch, t, ray_w, patch2d);
        mine = (m < M) && (!ray_active || ray_active[(int64_t)n * M + m] != 0);
        live[t] = mine;
    }
    for (int i = t; i < NBIN * 64; i += 256) { cnt[i] = 0; first[i] = 0x7fffffff; }
    if (!__syncthreads_or(mine)) {                // a patch of dead rays (most of a masked pseudo-view): nothing to order
        if (t == 0) count[blockIdx.x] = 0;
        return;
    }
    // The patch's 64 x S depths live in REGISTERS for all three passes (range, counts, ranks): wave w owns rays 16w .. 16w+15, lane l the samples
    // l, l + 64, ... of each -- every load a coalesced 256-B row piece, all of them in flight together, the ray index wave-uniform.  (Until
    // round 3 each pass re-read the depths through a flat p -> (p / S, p % S) loop: three times 48 dependent round trips, 95 us per image.)
    constexpr int RW = 16, KC = MAXBINS_S / 64;
    float dv[RW][KC];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int m = min(patch_ray(patch, wave * RW + r, ray_w, patch2d), M - 1);
        const float* row = depths + ((int64_t)n * M + m) * S;
#pragma unroll
        for (int c = 0; c < KC; ++c) dv[r][c] = (c * 64 < S) ? row[min(c * 64 + lane, S - 1)] : 0.f;
    }
    float lo = INFINITY, hi = -INFINITY;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        if (!live[wave * RW + r]) continue;       // wave-uniform
#pragma unroll
        for (int c = 0; c < KC; ++c)
            if (c * 64 + lane < S) { lo = fminf(lo, dv[r][c]); hi = fmaxf(hi, dv[r][c]); }
    }
    lo = wave_min(lo); hi = wave_max(hi);
    if (lane == 0) { red[0][wave] = lo; red[1][wave] = hi; }
    __syncthreads();
    lo = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    hi = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    const float inv = hi > lo ? (float)NBIN / (hi - lo) : 0.f;
    auto bin_of = [&](float d) { return min(max((int)((d - lo) * inv), 0), NBIN - 1); };
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int rl = wave * RW + r;
        if (!live[rl]) continue;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int k = c * 64 + lane;
            if (k < S) {
                const int b = bin_of(dv[r][c]);
                atomicAdd(&cnt[b * 64 + rl], 1);
                atomicMin(&first[b * 64 + rl], k);
            }
        }
    }
    __syncthreads();
    // exclusive prefix over the NBIN * 64 counts in (bin, ray) order: 16 consecutive entries per thread
    constexpr int PER = NBIN * 64 / 256;
    int loc[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { loc[i] = sum; sum += cnt[t * PER + i]; }
    const int incl = wave_scan_add(sum);
    if (lane == 63) wsum_s[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += wsum_s[w];
#pragma unroll
    for (int i = 0; i < PER; ++i) cnt[t * PER + i] = base + loc[i];
    if (t == 255) count[blockIdx.x] = base + sum;
    __syncthreads();
    uint16_t* out = order + (int64_t)blockIdx.x * 64 * S;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int rl = wave * RW + r;
        if (!live[rl]) continue;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int k = c * 64 + lane;
            if (k < S) {
                const int e = bin_of(dv[r][c]) * 64 + rl;
                out[cnt[e] + k - first[e]] = (uint16_t)((rl << 8) | k);
            }
        }
    }
}

struct TiledArgs {
    const float* planes; const float* ray_o; const float* ray_d; const float* depths; const int32_t* perm;
    const int32_t* ray_active;                // optional per-ray flags from spi_raymarch_bwd: 0 = the ray's gradient rows are all zero (and unwritten)
    int N; int M; int S; int H; int W; float scale;
    int ray_w; int patch2d;                   // ray grid width; 1 = 8x8 patches over the (M/ray_w) x ray_w grid, 0 = 64 consecutive rays
    int patches; int kchunks; int tiles;
    const uint16_t* order; const int32_t* count;   // depth-binned point order of every ray patch (bin_points_kernel)
    float* dfeat;                             // split scatter (plane_scatter_kernel): d_feat rows [patch][pos][32] in the patches' depth-binned order
    int dbg;                                  // tools/bench_render.py only: 1 = skip scatter, 4 = skip the MLP, 8 = no flush, 16 = skip points that leave the window, 32 = no LDS atomics, 64 = skip the plane gather
};

#define SPI_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <bool WGRAD, bool RGB>
__global__ void __launch_bounds__(DT, 2) decode_bwd_tiled_kernel(TiledArgs a, const float* __restrict__ frag_g, const float* __restrict__ b1_g,
                                                                 const float* __restrict__ b2_g, const float* __restrict__ d_rgb,
                                                                 const float* __restrict__ d_rgb_scale, const float* __restrict__ colors,
                                                                 const float* __restrict__ d_sigma, float* __restrict__ part) {
    // colors: the forward's colour rows [R*S, 32] (= sigmoid(y) * 1.002 - 0.001), required with d_rgb.  The sigmoid the colour layer's
    // derivative needs is read back from them instead of recomputing the layer's pre-activations (32 MFMAs per 32 points) for one
    // 128-byte row per point.
    // d_rgb_scale == NULL: d_rgb is the materialised per-sample gradient [R*S, 32].  Otherwise the gradient of sample row i of
    // ray r is d_rgb[r][:] * d_rgb_scale[i] (what the ray marcher's backward produces: a per-ray vector times a per-sample
    // scalar); the 2 MB per-ray array stays in L2 and the 128 B per point of gradient traffic disappears.
    //
    // LDS map (dwords):
    //   feat  [DT][FS]   rows [point][36]: features 0..31 | d_sigma 32 | ray 33 | colour-gradient scale 34; phase B overwrites 0..31 with d_feat
    //   gbuf  [FRAG_TOTAL]  the bf16 weight fragments (36 x 256 dwords) + the density row of W2 in accumulator order (64)
    //   s_row [DT], s_acc [8]
    // (Until round 5 the weight gradients' operands were transposed through four 4.25 KB LDS tiles behind 25 KB of fp32 fragments; the
    //  transpositions now run on the matrix cores -- see `transpose` below -- and the room went to the bf16 pieces of the fragments.)
    constexpr int LDS_GBUF = DT * FS, LDS_ROW = LDS_GBUF + FRAG_TOTAL, LDS_ACC = LDS_ROW + DT;
    __shared__ __attribute__((aligned(16))) float lds[LDS_ACC + 8];
    static_assert((LDS_ACC + 8) * 4 <= 80 * 1024, "two blocks per CU");
    float* const feat = lds;
    float* const gbuf = lds + LDS_GBUF;
    int* const s_row = reinterpret_cast<int*>(lds + LDS_ROW);  // row index into the [R*S] sample arrays (-1 = padding point)
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, q = lane & 31, hh = lane >> 5;
    // Weight-gradient accumulators: the four 32x32 tiles of a wave (dW1 rows 0-31 / 32-63, dW2 colour rows x hidden 0-31 / 32-63) stay
    // in 64 registers for the whole kernel and go to the wave's scratch row once, at the end.
    float s_sig[2] = {0.f, 0.f}, s_b1[2] = {0.f, 0.f}, s_b2 = 0.f, s_d = 0.f;
    float* const pr = part + ((int64_t)blockIdx.x * 4 + wave) * PART_ROW;     // this wave's scratch row
    // (Round 2 kept them in that row and pulled them through L2 around their own MFMAs: 64 loads + 64 stores per lane and 32 points.
    //  2048 rows x 17 KB do not fit the 4 MB of L2 an XCD has: 2 GB per image went out to memory and came back -- the "20x write
    //  amplification" PMC showed was mostly this, not the scatter.  With the scatter in its own kernel the decoder phases leave room for
    //  the 64 accumulator registers; the row is written once, at the end.)
    f32x16_t accW1[2], accW2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accW1[i][r] = 0.f; accW2[i][r] = 0.f; }
    // The weight fragments go to LDS ONCE (round 2 re-copied the 25 KB for every tile: the scatter phase used to overwrite them).
    if (!(a.dbg & 4))
        for (int i = t; i < FRAG_TOTAL / 4; i += DT) reinterpret_cast<float4*>(gbuf)[i] = reinterpret_cast<const float4*>(frag_g)[i];
    // A tile's point data comes out of a chain of dependent loads: count -> order -> permutation / depth / ray -> gradient rows.  A wave issues
    // in order, so a chain resolved at the top of a tile stalls it for three memory latencies (and a load inside a divergent `if` makes hipcc
    // wait for everything outstanding).  The chain is spread over the tile loop instead -- the id two tiles ahead, permutation / depth / ray
    // one tile ahead -- with unconditional loads on clamped indices; the gradient rows of the current tile are requested at its top and
    // written to LDS after the gather phase, which hides them.
    const int tpp = a.patches * a.kchunks;
    auto tile_pidx = [&](int tl, int& kc_) -> int { const int tc = min(tl, a.tiles - 1); const int n_ = tc / tpp; const int rem_ = tc - n_ * tpp; const int patch_ = rem_ / a.kchunks; kc_ = rem_ - patch_ * a.kchunks; return n_ * a.patches + patch_; };
    auto load_live = [&](int tl) -> int { int kc_; return a.count[tile_pidx(tl, kc_)]; };
    auto load_id = [&](int tl, int live) -> int {
        int kc_; const int pidx_ = tile_pidx(tl, kc_);
        return (int)a.order[(int64_t)pidx_ * (64 * a.S) + max(min(kc_ * DT + t, live - 1), 0)];
    };
    struct RawPoint { int prow; float dpt, o0, o1, o2, d0, d1, d2; };
    auto load_raw = [&](int tl, int id, RawPoint& r) {
        int kc_; const int pidx_ = tile_pidx(tl, kc_);
        const int n_ = pidx_ / a.patches, patch_ = pidx_ - n_ * a.patches;
        const int k_ = min(id & 255, a.S - 1);
        const int64_t ray = (int64_t)n_ * a.M + min(patch_ray(patch_, (id >> 8) & 63, a.ray_w, a.patch2d), a.M - 1);
        const int64_t si = ray * a.S + k_;
        r.prow = a.perm ? a.perm[si] : k_;                    // (block-uniform branch)
        r.dpt = a.depths[si];
        const float* o = a.ray_o + ray * 3; const float* d = a.ray_d + ray * 3;
        r.o0 = o[0]; r.o1 = o[1]; r.o2 = o[2]; r.d0 = d[0]; r.d1 = d[1]; r.d2 = d[2];
    };
    const int G = (int)gridDim.x;
    int live_cur = load_live(blockIdx.x), live_n = load_live(blockIdx.x + G), live_nn = load_live(blockIdx.x + 2 * G);
    int id_cur = load_id(blockIdx.x, live_cur), id_n = load_id(blockIdx.x + G, live_n);
    RawPoint raw_n;
    load_raw(blockIdx.x, id_cur, raw_n);
    for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
    __syncthreads();                                           // LDS of the previous tile is no longer in use
    const int n = tile / tpp;
    const int rem = tile - n * tpp;
    const int patch = rem / a.kchunks, kc = rem - patch * a.kchunks;
    const int pidx = n * a.patches + patch;
    // ---- rotate the prefetch pipeline (before any early exit)
    const RawPoint raw = raw_n;
    const int id = id_cur, live_pts = live_cur;
    id_cur = id_n; live_cur = live_n; live_n = live_nn;
    live_nn = load_live(tile + 3 * G);
    id_n = load_id(tile + 2 * G, live_n);
    load_raw(tile + G, id_cur, raw_n);
    if (kc * DT >= live_pts) continue;                         // past the patch's last live point (block-uniform)
    // ---- point owned by this thread: entry kc * 256 + t of the patch's depth-binned order
    const int pos = kc * DT + t;
    const bool valid = pos < live_pts;
    const int rl = (id >> 8) & 63, k = min(id & 255, a.S - 1);
    const int m = min(patch_ray(patch, rl, a.ray_w, a.patch2d), a.M - 1);
    const int64_t ray = (int64_t)n * a.M + m;
    const int row = valid ? (int)(ray * a.S + raw.prow) : -1;
    const float x = valid ? (raw.o0 + raw.dpt * raw.d0) * a.scale : 0.f, y = valid ? (raw.o1 + raw.dpt * raw.d1) * a.scale : 0.f,
                z = valid ? (raw.o2 + raw.dpt * raw.d2) * a.scale : 0.f;
    s_row[t] = row;
    // ---- phase A: gather features.  This thread's point -> its gather record (round 4, see gather_tile: the corner arithmetic once per point instead
    //      of once per lane of the 8-lane gather group); padding points get out-of-range offsets (zeros), no branch around the loads
    const unsigned plane_bytes = (unsigned)(a.H * a.W * DEC_IN * 4);
    if (!(a.dbg & 64)) write_gather_record(feat + t * FS, n, x, y, z, a.W, a.H, plane_bytes, valid);
    // the point's gradient scalars: requested now (clamped row), stored to its LDS row after the gather phase
    const int64_t rowc = ray * a.S + raw.prow;
    const float g_sigma = d_sigma[rowc];
    const float g_scale = (RGB && d_rgb_scale) ? d_rgb_scale[rowc] : 0.f;
    __syncthreads();
    if (!(a.dbg & 64)) gather_from_records(feat, make_rsrc(a.planes, (int64_t)a.N * 3 * plane_bytes), DT);       // (host: < 2 GiB)
    else for (int pass = 0; pass < DT / 32; ++pass) *reinterpret_cast<float4*>(feat + (pass * 32 + (t >> 3)) * FS + (t & 7) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);      // (tools/bench_render.py only)
    feat[t * FS + 32] = valid ? g_sigma : 0.f;                 // columns 32..34 of the point's LDS row: d_sigma | its ray | its colour-gradient scale
    if (RGB && d_rgb_scale) {
        feat[t * FS + 33] = __int_as_float(valid ? n * a.M + m : 0);
        feat[t * FS + 34] = valid ? g_scale : 0.f;
    }
    __syncthreads();
    // ---- phase B: decoder forward + backward on the matrix cores, 32 points (one MFMA N tile) at a time per wave.
    //      Everything below touches only this wave's 64 rows of feat / gbuf.
    if (!(a.dbg & 4)) {
#pragma unroll 1
    for (int nt = 0; nt < 2; ++nt) {
        const int pbase = wave * 64 + nt * 32;
        // The bias loads do not depend on the tile: without this opaque zero LICM hoists them out of the tile loop and keeps
        // them in registers for the whole kernel.
        int opq = 0;
        asm volatile("" : "+s"(opq));
        int q_ = q, hh_ = hh;                                   // opaque per iteration: keeps ~70 lane_-derived address terms from being
        asm volatile("" : "+v"(q_), "+v"(hh_));                 // hoisted out of the tile loop and parked in (spilled) registers
        const int lane_ = q_ + 32 * hh_;
        const u32x4_t* frag = reinterpret_cast<const u32x4_t*>(gbuf) + lane_;      // fragment f: frag[f * 64], one ds_read_b128 per lane
        const float* wsig = gbuf + FRAG_WSIG + 32 * hh_;           // W2[0][32 mt + rowmap(r, hh)] at [16 mt + r]
        const float* b1 = b1_g + opq;
        const float* b2 = b2_g + opq;
        float* frow = feat + (pbase + q_) * FS;                 // the lane_'s own point (orientation 1: lane_ <-> point)
        const float dsg = frow[32];                             // d_sigma of the lane_'s point
        const float dsq = hh_ == 0 ? dsg : 0.f;                 //   ... counted once (half 0) in the bias sum
        // d_rgb of the lane_'s point straight from HBM (its row is 128 contiguous bytes; each half takes 4 x 16 B), requested
        // now and consumed after the forward layer
        const int myrow = s_row[pbase + q_];
        // (unconditional loads on a clamped row, zeroed afterwards: a guarded load `ok ? *p : 0` becomes an exec-masked region
        // that ends in s_waitcnt vmcnt(0) -- 4 + 16 serialised HBM round trips per 32 points before this was changed)
        float4 dr[4];
        if (RGB) {
            const int64_t grow = d_rgb_scale ? (int64_t)__float_as_int(frow[33]) : (int64_t)max(myrow, 0);
            const float gsc = d_rgb_scale ? frow[34] : 1.f;
            const float4* drp = reinterpret_cast<const float4*>(d_rgb + grow * DEC_IN + 4 * hh_);
            const bool keep = myrow >= 0;              // select, not multiply: the clamped row may be an unwritten (inactive-ray) row
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v = drp[2 * g];
                dr[g] = make_float4(keep ? v.x * gsc : 0.f, keep ? v.y * gsc : 0.f, keep ? v.z * gsc : 0.f, keep ? v.w * gsc : 0.f);
            }
        }
        // Round 6: the three products of the data path (H1 = W1 F, dH1 = W2^T dY, dF = W1^T dpre1) run on v_mfma_f32_32x32x16_bf16 with every
        // fp32 operand cut into three bf16 pieces and the six significant piece products accumulated in fp32 (decode_fwd_mfma_kernel's scheme:
        // the arithmetic stays fp32's, 72 MFMAs of 32 cycles instead of 98 of 64); the activations are split in registers, in the
        // accumulator's own (register, half) order = the virtual K order the fragments were built in.
        // H1[j][p] = softplus(W1 F + b1)
        f32x16_t H1[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) H1[mt][r] = b1[mt * 32 + rowmap(r, hh_)];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const float4 f0 = *reinterpret_cast<const float4*>(frow + 16 * s2 + 8 * hh_), f1 = *reinterpret_cast<const float4*>(frow + 16 * s2 + 8 * hh_ + 4);
            const float v[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
            u32x4_t B[3];
            split3_bf16x8(v, B);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const u32x4_t A[3] = {frag[(FRAG_A1 + (mt * 2 + s2) * 3) * 64], frag[(FRAG_A1 + (mt * 2 + s2) * 3 + 1) * 64], frag[(FRAG_A1 + (mt * 2 + s2) * 3 + 2) * 64]};
                H1[mt] = mfma_split6(A, B, H1[mt]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) H1[mt][r] = softplus_fast(H1[mt][r]);
        // dY1[o][p] (orientation 1: lane <-> point) = d_rgb * d(sigmoid * 1.002 - 0.001), the sigmoid read back from the saved colour rows:
        // (c + 0.001) / 1.002.  RGB == false: only the density gradient is non-zero (SPI's depth branch) -> the whole colour layer drops out.
        f32x16_t Y1;
        u32x4_t BY[2][3];                                      // dY1 as B operand of dH1 / A operand of its transposition: pieces of registers 8 s .. 8 s + 7
        if (RGB) {
            const float4* crp = reinterpret_cast<const float4*>(colors + (int64_t)max(myrow, 0) * DEC_IN + 4 * hh_);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 c4 = crp[2 * g];
                Y1[4 * g] = c4.x; Y1[4 * g + 1] = c4.y; Y1[4 * g + 2] = c4.z; Y1[4 * g + 3] = c4.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sg = (Y1[r] + 0.001f) * (1.f / 1.002f);
                const float4 d4 = dr[r >> 2];
                const float dv1 = (r & 3) == 0 ? d4.x : ((r & 3) == 1 ? d4.y : ((r & 3) == 2 ? d4.z : d4.w));      // channel rowmap(r,hh_) = (r&3) + 8(r>>2) + 4hh
                Y1[r] = dv1 * 1.002f * sg * (1.f - sg);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = Y1[8 * s2 + e];
                split3_bf16x8(v, BY[s2]);
            }
        }
        // ---- weight gradients: they contract over the tile's 32 POINTS, i.e. both MFMA operands need the FEATURE in the lane position
        // ("orientation 2"), while the activations sit in accumulator layout (lane <-> point).  Round 2 recomputed them in orientation 2 (66 extra
        // MFMAs per 32 points), rounds 3-5 sent every 32 x 32 tile through a 4.25 KB LDS tile.  Round 6: the transposition is a matrix product with
        // the identity.  An accumulator-layout tile X[feature][point] of a lane IS a valid A operand with M = point and K = feature (A and B operands
        // have the same register form), and D = A I has the lane on N = feature and the registers on M = point: out[r] = X[q][point rowmap(r, hh)].
        // With X as bf16 pieces (exact split, products with 1.0 exact, three fp32 additions that reproduce x) the transposition is exact, costs
        // 6 MFMAs of 32 cycles per 32 x 32 tile and no LDS at all -- and the pieces are the ones the data path needs anyway.
        u32x4_t IDN[2];                                        // identity in virtual-K order: B[k = (hh, e) of step s][n = q] = (rowmap(8 s + e, hh) == q)
        if (WGRAD) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    IDN[s2][i] = (rowmap(8 * s2 + 2 * i, hh_) == q_ ? 0x3f80u : 0u) | (rowmap(8 * s2 + 2 * i + 1, hh_) == q_ ? 0x3f800000u : 0u);
        }
        auto transpose = [&](const u32x4_t (&x0)[3], const u32x4_t (&x1)[3], f32x16_t& out) {
            // in : pieces of the registers 0..7 (x0) and 8..15 (x1) of a tile the lane (p = q_, hh_) holds as v[r] = X[rowmap(r, hh_)][p]
            // out: lane (j = q_, hh_) holds out[r] = X[j][point rowmap(r, hh_)]
#pragma unroll
            for (int r = 0; r < 16; ++r) out[r] = 0.f;
#pragma unroll
            for (int c = 2; c >= 0; --c) {                     // small pieces first
                out = SPI_MFMA_BF16(x0[c], IDN[0], out);
                out = SPI_MFMA_BF16(x1[c], IDN[1], out);
            }
        };
        if (WGRAD) {
            f32x16_t Y2;                                           // orientation 2: dY[1 + q_][point rowmap(r, hh_)]
            if (RGB) {
                transpose(BY[0], BY[1], Y2);
#pragma unroll
                for (int r = 0; r < 16; ++r) s_b2 += Y2[r];
            }
            s_d += dsq;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                u32x4_t BH[2][3];                              // H1 tile i in pieces: registers 0..7 | 8..15
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = H1[i][8 * s2 + e];
                    split3_bf16x8(v, BH[s2]);
                }
                f32x16_t HT;
                transpose(BH[0], BH[1], HT);                       // HT[r] = H1[i*32 + q_][point rowmap(r, hh_)]
                float ls = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (RGB) accW2[i] = SPI_MFMA(Y2[r], HT[r], accW2[i]);       // dW2[o][j]: o = 1 + rowmap(r',h), j = i*32 + q_
                    ls = fmaf(feat[(pbase + rowmap(r, hh_)) * FS + 32], HT[r], ls);       // sigma row of dW2
                }
                s_sig[i] += ls;
            }
        }
        // dH1[j][p] = W2^T dY1 (colour rows on the matrix cores, the density row as a rank-1 term on the vector ALUs) -> dpre1 = dH1 * softplus'(pre1)
        f32x16_t dH1[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 wv = *reinterpret_cast<const float4*>(wsig + mt * 16 + 4 * g);
                dH1[mt][4 * g] = wv.x * dsg; dH1[mt][4 * g + 1] = wv.y * dsg; dH1[mt][4 * g + 2] = wv.z * dsg; dH1[mt][4 * g + 3] = wv.w * dsg;
            }
        if (RGB) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const u32x4_t A[3] = {frag[(FRAG_A3 + (mt * 2 + s2) * 3) * 64], frag[(FRAG_A3 + (mt * 2 + s2) * 3 + 1) * 64], frag[(FRAG_A3 + (mt * 2 + s2) * 3 + 2) * 64]};
                    dH1[mt] = mfma_split6(A, BY[s2], dH1[mt]);
                }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dH1[mt][r] *= (H1[mt][r] > 20.f) ? 1.f : (1.f - exp_fast(-H1[mt][r]));
        // dF[i][p] = W1^T dpre1, and (WGRAD) dW1[j][c] += sum_p dpre1[j][p] F[c][p]; the B operand of the latter is read from the feature rows (still F)
        f32x16_t dF;
#pragma unroll
        for (int r = 0; r < 16; ++r) dF[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4_t BD[2][3];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = dH1[i][8 * s2 + e];
                split3_bf16x8(v, BD[s2]);
                const int s4 = 2 * i + s2;
                const u32x4_t A[3] = {frag[(FRAG_A4 + s4 * 3) * 64], frag[(FRAG_A4 + s4 * 3 + 1) * 64], frag[(FRAG_A4 + s4 * 3 + 2) * 64]};
                dF = mfma_split6(A, BD[s2], dF);
            }
            if (WGRAD) {
                f32x16_t DT_;
                transpose(BD[0], BD[1], DT_);                      // DT_[r] = dpre1[i*32 + q_][point rowmap(r, hh_)]
                float lb = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    accW1[i] = SPI_MFMA(DT_[r], feat[(pbase + rowmap(r, hh_)) * FS + q_], accW1[i]);       // dW1[j][c]: j = i*32 + rowmap(r',h), c = q_
                    lb += DT_[r];
                }
                s_b1[i] += lb;
            }
        }
        // this tile's feature rows are dead now (the weight gradients are done with them): d_feat takes their place
#pragma unroll
        for (int r = 0; r < 16; ++r) frow[rowmap(r, hh_)] = dF[r] * THIRD;           // the plane mean contributes the 1/3
    }
    }
    // ---- hand-over to plane_scatter_kernel: this tile's 256 d_feat rows (contiguous: 32 KB; rows of padding points are zero).  The
    // scatter into the plane gradients wants windows that SURVIVE from one depth slab of a ray patch to the next (64 KB of fp64 per plane)
    // -- there is no room for them beside the rows, the fragments and the transposition tiles, so it is a kernel of its own.
    if (a.dbg & 1) continue;
    __syncthreads();                                       // phase B done in every wave: feat rows = d_feat
    {
        float* drow = a.dfeat + ((int64_t)pidx * (64 * a.S) + (int64_t)kc * DT) * DEC_IN;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = it * DT + t;                     // float4 index within the tile
            const int sp = e >> 3, q4 = e & 7;
            *reinterpret_cast<float4*>(drow + e * 4) = *reinterpret_cast<const float4*>(feat + sp * FS + q4 * 4);
        }
    }
    }   // tile loop
    if (WGRAD) {
        // the wave's four dW tiles + bias / sigma-row sums -> its scratch row; decoder_partial_reduce_kernel sums the rows
        for (int e = lane; e < PART_ROW; e += 64) pr[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pr[(i * 32 + rowmap(r, hh)) * DEC_IN + q] = accW1[i][r];
                if (RGB) pr[PART_DW2 + (1 + rowmap(r, hh)) * DEC_HID + i * 32 + q] = accW2[i][r];
            }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float sg = s_sig[i] + __shfl_xor(s_sig[i], 32, WAVE);
            const float sb = s_b1[i] + __shfl_xor(s_b1[i], 32, WAVE);
            if (hh == 0) { pr[PART_DW2 + i * 32 + q] = sg; pr[PART_DB1 + i * 32 + q] = sb; }     // sigma row of dW2, db1
        }
        const float sb2 = s_b2 + __shfl_xor(s_b2, 32, WAVE);
        if (hh == 0) pr[PART_DB2 + 1 + q] = sb2;
        const float sd = wave_sum(s_d);
        if (lane == 0) pr[PART_DB2] = sd;
    }
}

// ------------------------------------------------------------------------------------------------
// Plane-gradient scatter of the tiled decoder backward (round 3: its own kernel).
//   One block per (ray patch, plane) walks the patch's tiles -- 256 points each, thin slabs of the view frustum in depth order -- and
//   adds every point's d_feat row to its four bilinear corners in that plane.
//   Round 2 did this at the end of decode_bwd_tiled_kernel with ONE 16 x 16-texel LDS window per plane and TILE: zeroed, filled with
//   32-bit fixed-point LDS atomics (ds_add_f32 costs ~190 cycles per wave-instruction on gfx950), converted and flushed to HBM after
//   every tile -- 12 288 tiles x 3 windows x ~120 non-zero texel rows of 128-byte global atomics per image, ~20x the compulsory write
//   traffic (profiles/r02t_pmc_render_summary.txt), 0.87 ms of a 2.5 ms backward.
//   Two measurements changed the design (tools/ubench/lds_atomic.hip, round 3): (1) ds_add_f64 runs at 9.4 cycles per conflict-free
//   wave-instruction -- 20x faster than ds_add_f32, 1.5x the 32-bit integer add -- so the sums can simply be DOUBLES: no per-tile
//   scale, no zeroing, no conversion pass, and more exact than fp32; (2) a patch's footprint in a plane moves by only a few texels
//   from one slab to the next, so the window can live as long as the block: 16 x 16 texels x 32 channels of fp64 = 64 KB, addressed
//   modulo 16 in both axes (it follows the footprint without copying); a texel row goes to HBM once, when the footprint has moved
//   past it, not once per tile.  Three persistent windows do not fit beside the decoder's rows / fragments in one CU's LDS, hence the
//   split: decode_bwd_tiled_kernel leaves the tile's 256 d_feat rows in HBM (403 MB per image, read back once per plane).
//   Points whose corners leave the window go straight to HBM atomics (rare), so the result never depends on window placement.
//   LDS: 64 KB + 4 KB of point data -> two blocks of 512 threads per CU.
// ------------------------------------------------------------------------------------------------
constexpr int SCT = 512;                                     // threads per block: 16 half-waves, one 128-byte texel row each

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + barrier: hipcc drains EVERY outstanding
// memory operation before it (s_waitcnt vmcnt(0)), i.e. also the rows this kernel requests a whole tile ahead.  Here only the LDS
// counter is drained; loads stay in flight across the barrier and are awaited where their registers are first used.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// value of lane `j` of the caller's own half-wave (j: compile-time constant 0..31): two v_readlane + a select, no LDS round trip
__device__ __forceinline__ int half_bcast(int v, int j, bool upper) {
    const int lo = __builtin_amdgcn_readlane(v, j), hi = __builtin_amdgcn_readlane(v, j + 32);
    return upper ? hi : lo;
}
__device__ __forceinline__ float half_bcast(float v, int j, bool upper) { return __int_as_float(half_bcast(__float_as_int(v), j, upper)); }

// The footprint of every decoder tile in every plane -- the four corner rays of its ray patch at the depths of the tile's first and last
// point, one texel of slack -- as {x0, x1, y0, y1} (texels).  Round 5: plane_scatter_kernel evaluated these eight points per tile in EVERY one
// of its waves (~200 vector instructions of block-uniform arithmetic per tile and wave, a fifth of the kernel); one thread per (tile, plane)
// of this kernel does it once, and the scatter blocks read the four numbers through scalar loads.
__global__ void __launch_bounds__(256) tile_bbox_kernel(TiledArgs a, int4* __restrict__ bbox) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= a.tiles * 3) return;
    const int tile = e / 3, pl = e - tile * 3;
    const int pidx = tile / a.kchunks, kc = tile - pidx * a.kchunks;
    const int n = pidx / a.patches, patch = pidx - n * a.patches;
    const int live = a.count[pidx];
    if (kc * DT >= live) { bbox[e] = make_int4(0, 0, 0, 0); return; }
    const uint16_t* porder = a.order + (int64_t)pidx * (64 * a.S);
    auto depth_of = [&](int id) -> float {
        const int64_t ray = (int64_t)n * a.M + min(patch_ray(patch, (id >> 8) & 63, a.ray_w, a.patch2d), a.M - 1);
        return a.depths[ray * a.S + min(id & 255, a.S - 1)];
    };
    const float dfirst = depth_of(porder[kc * DT]), dlast = depth_of(porder[min(kc * DT + DT - 1, live - 1)]);
    int bx0 = 0x7fffffff, bx1 = -0x7fffffff, by0 = 0x7fffffff, by1 = -0x7fffffff;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int rl = (r & 1 ? 7 : 0) + (r & 2 ? 56 : 0);                // rays (0,0), (0,7), (7,0), (7,7) of the 8 x 8 patch
        const int64_t ray = (int64_t)n * a.M + min(patch_ray(patch, rl, a.ray_w, a.patch2d), a.M - 1);
        const float* o = a.ray_o + ray * 3; const float* d = a.ray_d + ray * 3;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float dpt = q ? dlast : dfirst;
            float gx, gy;
            plane_uv(pl, (o[0] + dpt * d[0]) * a.scale, (o[1] + dpt * d[1]) * a.scale, (o[2] + dpt * d[2]) * a.scale, gx, gy);
            const Corner cc = make_corner(gx, gy, a.W, a.H);
            bx0 = min(bx0, cc.x0); bx1 = max(bx1, cc.x0 + 1); by0 = min(by0, cc.y0); by1 = max(by1, cc.y0 + 1);
        }
    }
    bbox[e] = make_int4(bx0 - 1, bx1 + 1, by0 - 1, by1 + 1);            // (a depth bin is not infinitely thin: one texel of slack)
}

__global__ void __launch_bounds__(SCT, 4) plane_scatter_kernel(TiledArgs a, const int4* __restrict__ bbox, float* __restrict__ d_planes) {
    // Structure (what the first versions of this kernel got wrong, measured -- profiles/r03*):
    //  * a tile is only 256 points x 4 row-atomics, so anything paid per tile dominates.  A block-wide reduction for the window centre,
    //    point data handed from the computing thread to the scattering lanes through LDS arrays and four barriers cost 17 k cycles per
    //    tile for 5 k cycles of atomics.  Now the window position comes from block-UNIFORM data -- the patch's four corner rays at the
    //    depth of the tile's first and last point (tile_bbox_kernel) -- and the window only moves when that footprint no longer fits;
    //    only such a move needs barriers (the rows that leave are flushed by their owners between two of them).
    //  * every half-wave prepares the corners of its own 16 points in lanes 0..15 and hands them to its 32 lanes through a 24-byte LDS record.
    //  * Round 5: the kernel was VALU-issue bound on PER-TILE work (~1000 vector instructions per 256-point tile and wave, the atomics' own
    //    loop a third of it; profiles/r05a_bench_render.txt: 0.38 ms of vector work beside 0.26 ms of LDS atomics).  Gone: the footprint
    //    (-> tile_bbox_kernel + scalar loads), the row loads' address arithmetic (raw buffer loads: wave-uniform tile base in the scalar
    //    offset, the row in the immediate; rows past the patch's last point read as zeros), the copies of the row registers (two
    //    buffers whose roles alternate), the cell-offset arithmetic of the loop (records carry cell << 8: one v_and_or / SDWA-or per corner).
    //    Measured and dropped on the way: 512-point tiles with one point per thread (the slab is twice as thick, its XZ / ZX footprint no
    //    longer fits the 16 x 16 window: 1.79 vs 1.63 ms per image) and 256-thread blocks with 32 points per half-wave (half the per-tile
    //    work, but at 2 waves per SIMD the LDS round trips of the loop are exposed: 1.83 ms).
    __shared__ __attribute__((aligned(16))) double pwin[WIN * WIN * DEC_IN];        // the persistent window: cell ((y & 15) << 4 | (x & 15)), 32 channels each
    __shared__ __attribute__((aligned(16))) float tabw[SCT / 32][16][4];          // per half-wave, per point: w00 w01 w10 w11
    __shared__ __attribute__((aligned(8))) unsigned tabc[SCT / 32][16][2];        // byte offsets of the cells: (c00 << 8) | (c01 << 24), (c10 << 8) | (c11 << 24)
    const int t = threadIdx.x;
    const int hw = t >> 5, ch = t & 31;
    const bool upper = (t & 32) != 0;
    const int pidx = blockIdx.x, pl = blockIdx.y;
    const int n = pidx / a.patches, patch = pidx - n * a.patches;
    const int live_pts = a.count[pidx];
    if (live_pts <= 0) return;
    // blockIdx.z splits the patch's tiles into consecutive ranges (own window each): 256 patches x 3 planes = 768 blocks would fill the
    // chip's 512 slots 1.5 times (the second half-round runs on half the chip); 1536 blocks are exactly three rounds.
    const int ntiles = (live_pts + DT - 1) / DT;
    const int kc_begin = (int)((int64_t)ntiles * blockIdx.z / gridDim.z), kc_end = (int)((int64_t)ntiles * (blockIdx.z + 1) / gridDim.z);
    if (kc_begin >= kc_end) return;
    float* const gplane = d_planes + (int64_t)(n * 3 + pl) * ((int64_t)a.H * a.W * DEC_IN);
    for (int i = t; i < WIN * WIN * DEC_IN / 2; i += SCT) reinterpret_cast<double2*>(pwin)[i] = make_double2(0.0, 0.0);
    __syncthreads();
    int ox = 0, oy = 0;                                      // window origin (texels), valid once `placed`
    bool placed = false;
    // one texel row of the window -> HBM (non-zero entries only), then cleared
    auto flush_cell = [&](int cell, int xx, int yy) {
        double* pc = pwin + cell * DEC_IN + ch;
        const double v = *pc;
        if (v != 0.0) {
            if (!(a.dbg & 8)) atomicAdd(gplane + ((int64_t)yy * a.W + xx) * DEC_IN + ch, (float)v);
            *pc = 0.0;
        }
    };
    // Every prefetch is UNCONDITIONAL (clamped indices instead of guards): a load inside a divergent `if` makes hipcc lose count of what is
    // outstanding, and it then waits with s_waitcnt vmcnt(0) -- for the loads it has just issued for the next tile as well.  The chain
    // order -> depth / ray is spread over three iterations (id two tiles ahead, depth / ray one tile ahead, arithmetic on arrival): a
    // wave issues in order, and a dependent chain resolved inside one iteration would stall the tile for two memory latencies.
    const uint16_t* porder = a.order + (int64_t)pidx * (64 * a.S);
    auto load_id = [&](int kc, int p) -> int { return (int)porder[min(min(kc, ntiles - 1) * DT + p, live_pts - 1)]; };
    auto ray_of = [&](int id) -> int64_t { return (int64_t)n * a.M + min(patch_ray(patch, (id >> 8) & 63, a.ray_w, a.patch2d), a.M - 1); };
    struct RawPoint { float dpt, o0, o1, o2, d0, d1, d2; };
    auto load_raw = [&](int id, RawPoint& r) {
        const int64_t ray = ray_of(id);
        r.dpt = a.depths[ray * a.S + min(id & 255, a.S - 1)];
        const float* o = a.ray_o + ray * 3; const float* d = a.ray_d + ray * 3;
        r.o0 = o[0]; r.o1 = o[1]; r.o2 = o[2]; r.d0 = d[0]; r.d1 = d[1]; r.d2 = d[2];
    };
    const int4* const pbox = bbox + ((int64_t)pidx * a.kchunks) * 3 + pl;       // + 3 kc: this plane's footprint of tile kc (block-uniform -> scalar loads)
    const int myp = hw * 16 + (t & 15);                      // the point whose corner this lane prepares (lanes 16..31 of a half-wave mirror 0..15)
    // this half-wave's 16 rows of a tile, one channel per lane, requested one tile ahead (a tile is several us of work): the patch's rows are one buffer of live_pts * 128 bytes -- rows of decoder tiles that were never written
    // (past the last live point) read as zeros
    const __amdgpu_buffer_rsrc_t rrs = make_rsrc(a.dfeat + (int64_t)pidx * (64 * a.S) * DEC_IN, (int64_t)live_pts * (DEC_IN * 4));
    const int voff = (hw * 16) * (DEC_IN * 4) + ch * 4;      // per lane: its half-wave's first row, its channel
    auto load_rows = [&](int kc, float (&d)[16]) __attribute__((always_inline)) {
        const int soff = min(kc, ntiles - 1) * (DT * DEC_IN * 4);         // (uniform)
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, voff + j * (DEC_IN * 4), soff, 0));
    };
    float dva[16], dvb[16];
    load_rows(kc_begin, dva);
    int id_cur = load_id(kc_begin, myp), id_n = load_id(kc_begin + 1, myp);
    RawPoint raw_n;
    load_raw(id_cur, raw_n);
    int4 box_n = pbox[3 * kc_begin];
    const unsigned chb = (unsigned)ch * 8u;                  // this lane's channel inside a cell (256 B per cell)
    char* const wbase = reinterpret_cast<char*>(pwin);
    // one tile: `dv` holds its rows, `dvnext` receives those of tile kc + 1
    auto do_tile = [&](int kc, const float (&dv)[16], float (&dvnext)[16]) __attribute__((always_inline)) {
        const RawPoint raw = raw_n;
        const int4 box = box_n;
        const bool valid = kc * DT + myp < live_pts;
        id_cur = id_n;
        id_n = load_id(kc + 2, myp);
        load_rows(kc + 1, dvnext);
        load_raw(id_cur, raw_n);                             // (id_cur was requested during the previous iteration)
        box_n = pbox[3 * min(kc + 1, ntiles - 1)];
        // ---- the tile's footprint in this plane (block-uniform): does the window hold it?
        const int bx0 = box.x, bx1 = box.y, by0 = box.z, by1 = box.w;
        const bool fits = placed && bx0 >= ox && bx1 < ox + WIN && by0 >= oy && by1 < oy + WIN;
        if (!fits) {
            const int wx0 = (bx0 + bx1 + 1) / 2 - WIN / 2, wy0 = (by0 + by1 + 1) / 2 - WIN / 2;      // centred on the footprint
            if (placed) {
                // ---- move the window: texel rows that fall out of [wx0, wx0 + 16) x [wy0, wy0 + 16) go to HBM.  The only barriers of the loop.
                lds_barrier();                               // every wave's atomics of the earlier tiles have landed
                for (int cell = hw; cell < WIN * WIN; cell += SCT / 32) {
                    const int xo = ox + (((cell & (WIN - 1)) - ox) & (WIN - 1)), yo = oy + (((cell >> 4) - oy) & (WIN - 1));    // the texel this cell holds
                    if (xo < wx0 || xo >= wx0 + WIN || yo < wy0 || yo >= wy0 + WIN) flush_cell(cell, xo, yo);
                }
                lds_barrier();                               // the freed cells are zero before anybody adds to them
            }
            ox = wx0; oy = wy0; placed = true;
        }
        // ---- this lane's point: corner, bilinear fractions, window cells
        float gx, gy;
        plane_uv(pl, (raw.o0 + raw.dpt * raw.d0) * a.scale, (raw.o1 + raw.dpt * raw.d1) * a.scale, (raw.o2 + raw.dpt * raw.d2) * a.scale, gx, gy);
        const Corner c = make_corner(gx, gy, a.W, a.H);
        // points whose four corners all lie outside the plane image contribute nothing to this plane (padding_mode zeros)
        const bool vin = valid && c.x0 + 1 >= 0 && c.x0 < a.W && c.y0 + 1 >= 0 && c.y0 < a.H;
        const int lxo = c.x0 - ox, lyo = c.y0 - oy;
        // fast path: all four corners inside the image AND inside the window -> no per-corner tests in the loop
        const bool fast = vin && lxo >= 0 && lxo + 1 < WIN && lyo >= 0 && lyo + 1 < WIN &&
                          c.x0 >= 0 && c.x0 + 1 < a.W && c.y0 >= 0 && c.y0 + 1 < a.H;
        const int my_base = ((c.y0 & (WIN - 1)) << 4) | (c.x0 & (WIN - 1));          // the cell of corner (x0, y0)
        const int my_xy = ((c.x0 + 0x4000) & 0xffff) | ((c.y0 + 0x4000) << 16);
        const float my_wx = c.wx1, my_wy = c.wy1;
        {
            const float fx1 = my_wx, fy1 = my_wy;
            const float fx0 = 1.f - fx1, fy0 = 1.f - fy1;      // == (floor+1) - x up to 1 ulp; the forward uses the same pair through make_corner
            const int c01 = (my_base & ~(WIN - 1)) | ((my_base + 1) & (WIN - 1));            // x + 1, wrapped inside the row
            const int c10 = (my_base + WIN) & (WIN * WIN - 1), c11 = (c01 + WIN) & (WIN * WIN - 1);     // y + 1, wrapped
            if ((t & 16) == 0) {                               // lanes 0..15 of the half-wave own its 16 points (lanes 16..31 mirror them)
                // points that are skipped (padding, outside the plane) or take the slow path below get ZERO weights on cell 0: the main loop has
                // no branch (a branch on a value that has just been read from LDS costs the LDS latency per point -- measured, first version)
                *reinterpret_cast<float4*>(&tabw[hw][t & 15][0]) = fast ? make_float4(fx0 * fy0, fx1 * fy0, fx0 * fy1, fx1 * fy1) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<uint2*>(&tabc[hw][t & 15][0]) = fast ? make_uint2(((unsigned)my_base << 8) | ((unsigned)c01 << 24), ((unsigned)c10 << 8) | ((unsigned)c11 << 24))
                                                                       : make_uint2(0u, 0u);
            }
            asm volatile("" ::: "memory");
        }
        const unsigned long long slow_all = __ballot(vin && !fast && (t & 16) == 0);     // bits 0..15: lower half-wave's points, 32..47: upper's
        // ---- a half-wave (32 lanes = the 32 channels of one texel row) per point: four ds_add_f64 of 32 consecutive doubles, straight-line code
        // The records are read four points at a time, the NEXT four before the atomics of the current four are issued: LDS operations of a
        // wave complete in order, so a record read issued behind 16 atomics waits for all of them (and for the other 15 waves' atomics queued
        // in between) -- with one read per point the loop ran at the LDS round-trip latency, not at the atomics' rate.
        if (!(a.dbg & 32)) {
            uint2 pcb[2][4]; float4 wb[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { pcb[0][i] = *reinterpret_cast<const uint2*>(&tabc[hw][i][0]); wb[0][i] = *reinterpret_cast<const float4*>(&tabw[hw][i][0]); }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < 3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { pcb[(g + 1) & 1][i] = *reinterpret_cast<const uint2*>(&tabc[hw][4 * g + 4 + i][0]); wb[(g + 1) & 1][i] = *reinterpret_cast<const float4*>(&tabw[hw][4 * g + 4 + i][0]); }
                }
                asm volatile("" ::: "memory");                 // (keeps the reads above the atomics)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint2 pc = pcb[g & 1][i];
                    const float4 w = wb[g & 1][i];
                    const float dvj = dv[4 * g + i];
                    // the product in fp32 (what grid_sample's backward forms too), the SUM in fp64
                    atomicAdd(reinterpret_cast<double*>(wbase + ((pc.x & 0xff00u) | chb)), (double)(dvj * w.x));
                    atomicAdd(reinterpret_cast<double*>(wbase + ((pc.x >> 16) | chb)), (double)(dvj * w.y));
                    atomicAdd(reinterpret_cast<double*>(wbase + ((pc.y & 0xff00u) | chb)), (double)(dvj * w.z));
                    atomicAdd(reinterpret_cast<double*>(wbase + ((pc.y >> 16) | chb)), (double)(dvj * w.w));
                }
                asm volatile("" ::: "memory");
            }
        }
        // ---- rare: points with a corner outside the window or outside the plane image, corner by corner (wave-uniform test first)
        if (slow_all != 0ull && !(a.dbg & 16)) {
            const unsigned mine = (unsigned)(upper ? (slow_all >> 32) : slow_all) & 0xffffu;
            const unsigned either = (unsigned)((slow_all | (slow_all >> 32)) & 0xffffu);      // wave-uniform: point j of either half-wave is slow
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (!((either >> j) & 1u)) continue;           // scalar branch: most waves have one or two such points, not sixteen
                const float fx1 = half_bcast(my_wx, j, upper), fy1 = half_bcast(my_wy, j, upper);
                const int pk = half_bcast(my_xy, j, upper);
                if (!((mine >> j) & 1u)) continue;
                const float fx0 = 1.f - fx1, fy0 = 1.f - fy1;
                const float dvj = dv[j];
                const int x0 = (pk & 0xffff) - 0x4000, y0 = (pk >> 16) - 0x4000;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int cx = qq & 1, cy = qq >> 1;
                    const int xx = x0 + cx, yy = y0 + cy;
                    if (xx < 0 || xx >= a.W || yy < 0 || yy >= a.H) continue;
                    const float wq = (cx ? fx1 : fx0) * (cy ? fy1 : fy0);
                    if (xx >= ox && xx < ox + WIN && yy >= oy && yy < oy + WIN) {
                        if (!(a.dbg & 32)) atomicAdd(pwin + (((yy & (WIN - 1)) << 4) | (xx & (WIN - 1))) * DEC_IN + ch, (double)(dvj * wq));
                    } else if (!(a.dbg & 128)) {
                        atomicAdd(gplane + ((int64_t)yy * a.W + xx) * DEC_IN + ch, dvj * wq);                    // rare: straight to HBM
                    }
                }
            }
        }
    };
    for (int kc = kc_begin; kc < kc_end; kc += 2) {
        do_tile(kc, dva, dvb);
        if (kc + 1 < kc_end) do_tile(kc + 1, dvb, dva);
    }
    // ---- the block's tiles are done: everything still resident goes to HBM
    __syncthreads();
    if (placed)
        for (int cell = hw; cell < WIN * WIN; cell += SCT / 32)
            flush_cell(cell, ox + (((cell & (WIN - 1)) - ox) & (WIN - 1)), oy + (((cell >> 4) - oy) & (WIN - 1)));
}

// ------------------------------------------------------------------------------------------------
// Decoder weight gradients from the activation dump of the backward kernels
//   dump rows (each `cols` long, contiguous): f 0..31 | h 32..95 | d_pre1 96..159 | d_y 160..192 (160 = sigma)
//   dW1[j][i] = sum_p d_pre1[j][p] f[i][p]      (64 x 32)      db1[j] = sum_p d_pre1[j][p]
//   dW2[o][j] = sum_p d_y[o][p]   h[j][p]       (33 x 64)      db2[o] = sum_p d_y[o][p]
//   A streaming reduction over points: HBM-bound (772 B/point).  Four 32x32 MFMA tiles, one per wave
//   (dW1 rows 0-31 / 32-63, dW2 rgb rows x h cols 0-31 / 32-63); the sigma row of dW2 and the two bias
//   sums ride along on the VALU while the tiles are staged.  Blocks own point chunks and add their
//   partial results to the (pre-zeroed) outputs with global atomics.
// ------------------------------------------------------------------------------------------------
constexpr int WG_BK = 16;                  // points per slab
constexpr int WG_LD = 192 + 4;             // LDS row: [dpre 64 | dy_rgb 32 | f 32 | h 64] + pad

__global__ void __launch_bounds__(256) decoder_wgrad_kernel(const float* __restrict__ dump, int64_t cols, int64_t chunk,
                                                            float* __restrict__ dw1, float* __restrict__ db1,
                                                            float* __restrict__ dw2, float* __restrict__ db2) {
    __shared__ __attribute__((aligned(16))) float sm[2][WG_BK * WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t pbeg = (int64_t)blockIdx.x * chunk, pend = min(pbeg + chunk, cols);
    if (pbeg >= pend) return;
    // staging map: thread -> (row r = tid / 4 [+64, +128], 4 consecutive points p4 = tid % 4)
    const int r0 = tid >> 2, p4 = (tid & 3) * 4;
    // LDS column of the three rows this thread stages, and their dump rows
    //   LDS cols: 0..63 dpre (dump 96..159), 64..95 dy_rgb (161..192), 96..127 f (0..31), 128..191 h (32..95)
    int lcol[3], drow[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int lc = r0 + 64 * q;
        lcol[q] = lc;
        drow[q] = lc < 64 ? 96 + lc : (lc < 96 ? 161 + (lc - 64) : (lc < 128 ? lc - 96 : 32 + (lc - 128)));
    }
    float4 rv[3]; float4 sg4;                       // sg4: d_y sigma for the thread's 4 points
    float bsum[3] = {0.f, 0.f, 0.f};                // row sums of dpre / dy_rgb rows handled by this thread (q = 0,1)
    float sigacc = 0.f;                             // sum_p dsig[p] * h[j][p] for the h row of q = 2 ... and q = 1/0? (only h rows)
    float sigsum = 0.f;
    auto load = [&](int64_t p) {
        const int64_t pp = p + p4;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float* src = dump + (int64_t)drow[q] * cols + pp;
            if (pp + 3 < pend) rv[q] = *reinterpret_cast<const float4*>(src);
            else { rv[q].x = pp < pend ? src[0] : 0.f; rv[q].y = pp + 1 < pend ? src[1] : 0.f; rv[q].z = pp + 2 < pend ? src[2] : 0.f; rv[q].w = 0.f; }
        }
        const float* ss = dump + (int64_t)160 * cols + pp;
        if (pp + 3 < pend) sg4 = *reinterpret_cast<const float4*>(ss);
        else { sg4.x = pp < pend ? ss[0] : 0.f; sg4.y = pp + 1 < pend ? ss[1] : 0.f; sg4.z = pp + 2 < pend ? ss[2] : 0.f; sg4.w = 0.f; }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float* d = sm[buf] + p4 * WG_LD + lcol[q];
            d[0] = rv[q].x; d[WG_LD] = rv[q].y; d[2 * WG_LD] = rv[q].z; d[3 * WG_LD] = rv[q].w;
            bsum[q] += (rv[q].x + rv[q].y) + (rv[q].z + rv[q].w);
        }
        // rows with LDS col >= 128 are h rows: q = 2 always (r0 + 128 >= 128)
        sigacc += rv[2].x * sg4.x + rv[2].y * sg4.y + rv[2].z * sg4.z + rv[2].w * sg4.w;
        if (r0 == 0) sigsum += (sg4.x + sg4.y) + (sg4.z + sg4.w);
    };
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // wave -> tile: 0: dW1 rows 0-31, 1: dW1 rows 32-63 (A = dpre cols, B = f);  2: dW2 rgb x h 0-31, 3: dW2 rgb x h 32-63
    const int a_off = wave == 0 ? 0 : (wave == 1 ? 32 : 64);
    const int b_off = wave < 2 ? 96 : (wave == 2 ? 128 : 160);
    const int fr = lane & 31, fk = lane >> 5;
    const int nslab = (int)((pend - pbeg + WG_BK - 1) / WG_BK);
    load(pbeg); store(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) load(pbeg + (int64_t)(s + 1) * WG_BK);
        const float* base = sm[buf];
#pragma unroll
        for (int kk = 0; kk < WG_BK / 2; ++kk) {
            const float af = base[(2 * kk + fk) * WG_LD + a_off + fr];
            const float bf = base[(2 * kk + fk) * WG_LD + b_off + fr];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
        }
        if (s + 1 < nslab) store(buf ^ 1);
        __syncthreads();
    }
    // ---- write-out.  acc[r]: row (r&3) + 8*(r>>2) + 4*fk of the tile's A rows, column fr of its B rows
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (wave < 2) atomicAdd(dw1 + (wave * 32 + row) * 32 + fr, acc[r]);                        // dW1[j][i]
        else atomicAdd(dw2 + (1 + row) * 64 + (wave - 2) * 32 + fr, acc[r]);                       // dW2[1 + rgb][j]
    }
    // row sums: the 4 threads sharing a row (tid & 3) combine, then one atomic per row
#pragma unroll
    for (int q = 0; q < 3; ++q) bsum[q] = quad_sum(bsum[q]);
    sigacc = quad_sum(sigacc);
    sigsum = quad_sum(sigsum);
    if ((tid & 3) == 0) {
        atomicAdd(db1 + r0, bsum[0]);                                  // q = 0: dpre row r0
        if (r0 < 32) atomicAdd(db2 + 1 + r0, bsum[1]);                 // q = 1: lcol 64..95 are dy_rgb rows (r0 < 32); 96..127 are f rows (no sum needed)
        atomicAdd(dw2 + r0, sigacc);                                   // q = 2: h row r0 -> dW2[sigma][r0]
        if (r0 == 0) atomicAdd(db2, sigsum);
    }
}

// ------------------------------------------------------------------------------------------------
// min / max reduction (ray_marcher.py:50 clamps composite depth to the range of ALL depths)
// ------------------------------------------------------------------------------------------------
__global__ void minmax_init_kernel(float* out2) { out2[0] = INFINITY; out2[1] = -INFINITY; }

__device__ __forceinline__ void atomic_min_f(float* addr, float v) {     // valid for any sign
    if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f(float* addr, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void minmax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out2) {
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[g];
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    lo = wave_min(lo); hi = wave_max(hi);
    __shared__ float red[2][4];                  // same-address global atomics serialise (~10 ns each): one pair per block
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lo; red[1][threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomic_min_f(out2, fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3])));
        atomic_max_f(out2 + 1, fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3])));
    }
}

// ------------------------------------------------------------------------------------------------
// MipRayMarcher2 (ray_marcher.py:25-57).  Forward: one wave64 per ray, 4 rays per block; backward: up to 8 rays per wave.
//   scalars per interval live one-per-lane (S <= 256 -> up to 4 chunks); transmittance is an
//   exclusive product scan on the DPP path (common.hpp); colour rows are read 8 rows (1 KB) per
//   wave-instruction, 8 lanes x float4 per 128-B row, and reduced across the row groups by DPP adds / xor-shuffles.
// ------------------------------------------------------------------------------------------------
constexpr int MAXS = 256;
constexpr int RM_WAVES = 4;
constexpr int RM_RPW = 8;               // backward: rays per wave -- their eight 128-B gradient rows are ONE 1 KB wave-instruction
constexpr int RM_RING = 8;              // colour row groups (8 rows = 1 KB per wave-instruction) a wave keeps in flight

struct MarchLds { float sig[MAXS]; float dep[MAXS]; float w[MAXS]; float q[MAXS]; int row[MAXS]; float sraw[MAXS]; };   // sraw: the ray's densities in storage order

// First round trip of a ray: its sort permutation, depths and densities (the densities in STORAGE order, coalesced, in the same round
// trip as the permutation, which is applied from LDS afterwards -- gathering them through perm from global memory was a second,
// dependent round trip: final march 0.75 -> 0.80 of the HBM roofline, and the depth-only marches read nothing else).  All loads are
// unconditional on clamped indices (guarded loads compile to one exec-masked region each with a wait in between), addressed as a
// wave-uniform row base + a 32-bit lane offset (the 64-bit per-lane address arithmetic was ~400 of the kernel's VALU instructions).
template <int NCH>
struct MarchRow { int pk[NCH]; float dp[NCH]; float sr[MAXS / 64]; };

template <int NCH>
__device__ __forceinline__ void march_fetch(MarchRow<NCH>& P, const float* __restrict__ densities, const float* __restrict__ depths,
                                            const int32_t* __restrict__ perm, int64_t r, int S, int S_store, int lane) {
    const float* dep_r = depths + r * S;
    const float* den_r = densities + r * S_store;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const unsigned k = (unsigned)min(c * 64 + lane, S - 1);
        P.pk[c] = perm ? (perm + r * S)[k] : (int)k;              // row of sample k inside the ray's S_store rows
        P.dp[c] = dep_r[k];
    }
#pragma unroll
    for (int c = 0; c < MAXS / 64; ++c) P.sr[c] = den_r[(unsigned)min(c * 64 + lane, S_store - 1)];
}

// ... into LDS (chunk c, lane l <-> sample k = c*64 + l; every index written is < MAXS, entries past S hold clamped duplicates) and the
// sorted densities back out of it.
template <int NCH>
__device__ __forceinline__ void march_stage(MarchLds& L, const MarchRow<NCH>& P, int lane, float (&sg)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) { L.row[c * 64 + lane] = P.pk[c]; L.dep[c * 64 + lane] = P.dp[c]; }
#pragma unroll
    for (int c = 0; c < MAXS / 64; ++c) L.sraw[c * 64 + lane] = P.sr[c];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < NCH; ++c) sg[c] = L.sraw[P.pk[c]];
}

// alpha / transmittance for every interval of the ray, one interval per lane and chunk (exclusive product scan on the DPP path).
template <int NCH>
__device__ __forceinline__ void march_scan(const MarchLds& L, int S, int lane, float (&alpha)[NCH], float (&trans)[NCH],
                                           float (&delta)[NCH], float (&smid)[NCH]) {
    float carry = 1.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k = c * 64 + lane;
        float a = 0.f, om = 1.f, dl = 0.f, sm = 0.f;
        if (k < S - 1) {
            dl = L.dep[k + 1] - L.dep[k];
            sm = (L.sig[k] + L.sig[k + 1]) / 2.f;
            a = 1.f - expf(-softplus_f(sm - 1.f) * dl);
            om = 1.f - a + 1e-10f;
        }
        const float incl = wave_scan_mul(om);
        const float excl = dpp_f32<0x138>(1.f, incl);               // wave_shr:1, lane 0 keeps 1
        alpha[c] = a; delta[c] = dl; smid[c] = sm;
        trans[c] = carry * excl;
        carry *= lane_bcast(incl, 63);
        __builtin_amdgcn_sched_barrier(0);          // one chunk's exponentials at a time: interleaved, the three chunks' temporaries cost ~40 registers
    }
}

// (RM_RING row groups in flight: 66 registers, six waves per SIMD -- the depth-only marches, which are all arithmetic, went from 84 to 69 us
// per 65 536 rays with the occupancy; the composite is the same 86 us at 3 to 6 waves per SIMD and 4 to 24 KB in flight per wave)
template <int NCH>
__global__ void __launch_bounds__(64 * RM_WAVES) __attribute__((amdgpu_waves_per_eu(6, 6))) raymarch_fwd_kernel(
        const float* __restrict__ colors, const float* __restrict__ densities, const float* __restrict__ depths,
        const int32_t* __restrict__ perm, const float* __restrict__ clamp2, int64_t R, int S, int S_store, int white_back,
        float* __restrict__ rgb, float* __restrict__ depth_out, float* __restrict__ weights, float* __restrict__ wsum_out) {
    __shared__ MarchLds lds[RM_WAVES];
    constexpr int NIT = NCH * 8, RING = RM_RING < NIT ? RM_RING : NIT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * RM_WAVES + wave));     // wave-uniform: row bases live in SGPRs
    if (r >= R) return;
    MarchLds& L = lds[wave];
    // Round trip 1: march_fetch.  Round trip 2: the colour rows of the ray (8 rows = 1 KB per wave-instruction): the first RING row
    // groups are requested before the scans and exponentials run, the rest stream through the ring while the composite accumulates.
    const int sub = lane & 7, rg = lane >> 3;
    float4 ring[RING];
    const float* col_r = colors + r * S_store * 32;
    auto cload = [&](int it) {
        const int k = min(it * 8 + rg, S - 1);
        // read once, never again: non-temporal (keeps the 400 MB colour stream from evicting what the next kernels reuse)
        const f32x4_t cv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(col_r + (unsigned)(L.row[k] * 32 + sub * 4)));
        return make_float4(cv.x, cv.y, cv.z, cv.w);
    };
    {
        MarchRow<NCH> P;
        march_fetch<NCH>(P, densities, depths, perm, r, S, S_store, lane);
        float sg[NCH];
        march_stage<NCH>(L, P, lane, sg);
        if (rgb != nullptr) {
#pragma unroll
            for (int it = 0; it < RING; ++it) ring[it] = cload(it);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) L.sig[c * 64 + lane] = sg[c];
        __builtin_amdgcn_wave_barrier();
    }
    float alpha[NCH], trans[NCH], delta[NCH], smid[NCH];
    march_scan<NCH>(L, S, lane, alpha, trans, delta, smid);
    float wsum = 0.f, dnum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k = c * 64 + lane;
        const float w = alpha[c] * trans[c];
        if (k < S - 1) {
            wsum += w;
            dnum += w * ((L.dep[k] + L.dep[k + 1]) / 2.f);
            if (weights) (weights + r * (S - 1))[(unsigned)k] = w;
        }
        L.w[k] = (k < S - 1) ? w : 0.f;
    }
    wsum = wave_sum(wsum); dnum = wave_sum(dnum);
    if (lane == 0) {
        if (depth_out) {
            float d = dnum / wsum;
            if (d != d) d = INFINITY;                               // nan_to_num(nan -> inf)
            d = fminf(fmaxf(d, clamp2[0]), clamp2[1]);
            depth_out[r] = d;
        }
        if (wsum_out) wsum_out[r] = wsum;
    }
    if (rgb == nullptr) return;
    __builtin_amdgcn_wave_barrier();
    // sum_k w_k (c_k + c_{k+1})/2  ==  sum_k c_k * (w_{k-1} + w_k)/2
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = it * 8 + rg;
        const float v = (k < S) ? 0.5f * ((k > 0 ? L.w[k - 1] : 0.f) + L.w[k]) : 0.f;
        const float4 c4 = ring[it % RING];
        acc.x = fmaf(v, c4.x, acc.x); acc.y = fmaf(v, c4.y, acc.y);
        acc.z = fmaf(v, c4.z, acc.z); acc.w = fmaf(v, c4.w, acc.w);
        if (it + RING < NIT) {
            // the fence keeps the compiler from hoisting every load to the top again, and pins the accumulation in front of the refill
            // (a sunk FMA keeps its ring slot alive: all 24 row groups ended up in registers of their own)
            asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w) :: "memory");
            ring[it % RING] = cload(it + RING);
        }
    }
    // across the 8 row groups (lanes with the same sub): row_ror:8 pairs lane i with i ^ 8 inside a row of 16; 16 and 32 cross rows
    acc.x += dpp_f32<0x128>(0.f, acc.x); acc.y += dpp_f32<0x128>(0.f, acc.y); acc.z += dpp_f32<0x128>(0.f, acc.z); acc.w += dpp_f32<0x128>(0.f, acc.w);
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
        acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
    }
    if (rg == 0) {
        const float wb = white_back ? 1.f - wsum : 0.f;
        float4 o4 = make_float4((acc.x + wb) * 2.f - 1.f, (acc.y + wb) * 2.f - 1.f, (acc.z + wb) * 2.f - 1.f,
                                (acc.w + wb) * 2.f - 1.f);
        *reinterpret_cast<float4*>(rgb + r * 32 + sub * 4) = o4;
    }
}

// Backward.  One wave owns up to RM_RPW = 8 rays and walks its live ones: a ray whose incoming gradient is exactly zero
// contributes exactly zero everywhere downstream, so it is only flagged (its d_colors / d_color_scale / d_densities rows are NOT
// written and must not be read -- spi_triplane_decode_bwd_sorted takes the same flags).  SPI's masked pseudo-view losses leave
// 65-90 % of the rays of those views in this state; the 128-B gradient rows of a wave's rays are ONE wave-instruction (8 lanes x
// float4 per row), so a dead ray costs an eighth of a load and a few scalar bit operations (round 2: one wave, one dependent flag
// round trip and one store per ray -- 60 us per 49 000 dead rays, and the flag round trip sat at the head of every live ray too).
// The rays of wave g are g, g + W, g + 2W, ... (W = waves of the launch): live regions of an image are contiguous, and the stride
// spreads them evenly over the waves (8 CONSECUTIVE rays per wave: 138 us for 16 384 live of 65 536 rays; strided: 127-133 us;
// all 16 384 live rays of a dense launch: 100 us).  While a ray's colour rows stream in, the first round trip of the wave's NEXT
// live ray is already issued.  The colour rows are consumed as they arrive (q_k = <d_rgb, c_k> needs nothing from the scans)
// through a ring of RM_RING row groups: 32 registers instead of 96 for the 24 KB of a ray, so three waves share a SIMD.
// Where a ray's ~29 000 cycles go (s_memtime stamps per section, dense launch): colour loop 45 % (waiting for memory), staging + issue
// 22 %, suffix sums + stores 21 %, scans 12 %; 92 us per 16 384 rays against the 68 us a ray-shaped non-temporal read of the same 430 MB
// takes (tools/ubench/read_bw).  Measured no better on top of this form: 2 / 3 / 4 waves per SIMD, rings of 4 to 24 row groups, and a
// cross-ray double buffer that keeps the ring full through the arithmetic phases (89.6 us dense, but 138 against 128 us masked).
// gfx9 notes from that work: loads and stores share ONE in-order counter, and a load in flight across a loop's back edge, or issued
// inside a branch, is awaited with vmcnt(0) -- keep the count of what is in flight static; `const __restrict__` loads are hoisted across
// asm("" ::: "memory") fences unless an address operand passes through a volatile asm.
template <int NCH>
__global__ void __launch_bounds__(64 * RM_WAVES) __attribute__((amdgpu_waves_per_eu(NCH <= 3 ? 3 : 2, NCH <= 3 ? 3 : 2))) raymarch_bwd_kernel(
        const float* __restrict__ colors, const float* __restrict__ densities, const float* __restrict__ depths,
        const int32_t* __restrict__ perm, const float* __restrict__ clamp2, const float* __restrict__ d_rgb,
        const float* __restrict__ d_depth, const float* __restrict__ d_weights, int64_t R, int S, int S_store, int white_back, int rpw,
        float* __restrict__ d_colors, float* __restrict__ d_color_scale, float* __restrict__ d_densities, int32_t* __restrict__ ray_active) {
    __shared__ MarchLds lds[RM_WAVES];
    constexpr int NIT = NCH * 8, RING = RM_RING < NIT ? RM_RING : NIT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t gw = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * RM_WAVES + wave));      // wave-uniform: row bases live in SGPRs
    const int64_t W = (int64_t)gridDim.x * RM_WAVES;
    if (gw >= R) return;
    const int sub = lane & 7, rg = lane >> 3;
    uint64_t live;                                  // bit 8*j: ray gw + j*W is to be marched
    float ddep_lane = 0.f;                          // d_depth of ray gw + rg*W (lanes 8*rg .. 8*rg+7)
    {
        const int64_t rr = gw + rg * W;
        const bool valid = rg < rpw && rr < R;
        const int64_t rc = valid ? rr : gw;
        if (d_depth) ddep_lane = d_depth[rc];
        bool nz = true;
        if (ray_active && !d_weights) {
            nz = ddep_lane != 0.f;
            if (d_rgb) {
                const float4 g = *reinterpret_cast<const float4*>(d_rgb + rc * 32 + sub * 4);
                nz = nz || g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f;
            }
        }
        uint64_t m = __ballot(nz && valid);
        m |= m >> 4; m |= m >> 2; m |= m >> 1;
        live = m & 0x0101010101010101ull;
        if (ray_active && sub == 0 && valid) ray_active[rr] = (int)((live >> (8 * rg)) & 1);
    }
    if (live == 0) return;
    MarchLds& L = lds[wave];
    int j = __builtin_ctzll(live) >> 3;
    MarchRow<NCH> P;
    march_fetch<NCH>(P, densities, depths, perm, gw + j * W, S, S_store, lane);
#pragma nounroll
    while (true) {
        const int64_t r = gw + j * W;
        const uint64_t rest = live & (~0xffull << (8 * j));            // live rays after j
        const int jn = rest ? (__builtin_ctzll(rest) >> 3) : -1;
        // Same memory schedule as the forward: round trip 1 = march_fetch (already in flight / arrived), round trip 2 = the colour
        // rows of the ray, the first RING row groups (8 rows = 1 KB per wave-instruction) requested before the scans run.
        float4 ring[RING];
        float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* col_r = colors + r * S_store * 32;
        auto cload = [&](int it) {
            const int k = min(it * 8 + rg, S - 1);
            const f32x4_t cv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(col_r + (unsigned)(L.row[k] * 32 + sub * 4)));   // last use of the colour rows
            return make_float4(cv.x, cv.y, cv.z, cv.w);
        };
        {
            float sg[NCH];
            march_stage<NCH>(L, P, lane, sg);
            if (d_rgb != nullptr) {
                g4 = *reinterpret_cast<const float4*>(d_rgb + r * 32 + sub * 4);
#pragma unroll
                for (int it = 0; it < RING; ++it) ring[it] = cload(it);
            }
            if (jn >= 0) march_fetch<NCH>(P, densities, depths, perm, gw + jn * W, S, S_store, lane);
#pragma unroll
            for (int c = 0; c < NCH; ++c) L.sig[c * 64 + lane] = sg[c];
            __builtin_amdgcn_wave_barrier();
        }
        float alpha[NCH], trans[NCH], delta[NCH], smid[NCH];
        march_scan<NCH>(L, S, lane, alpha, trans, delta, smid);
        float wsum = 0.f, dnum = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k = c * 64 + lane;
            const float w = alpha[c] * trans[c];
            if (k < S - 1) { wsum += w; dnum += w * ((L.dep[k] + L.dep[k + 1]) / 2.f); }
            L.w[k] = (k < S - 1) ? w : 0.f;
        }
        wsum = wave_sum(wsum); dnum = wave_sum(dnum);
        __builtin_amdgcn_wave_barrier();
        // q_k = <d_rgb, c_k>.  The colour-row gradient is d_rgb * (w_{k-1} + w_k): a per-ray vector times a per-sample scalar.
        // d_color_scale != NULL: only that scalar is written (4 B per sample; the decoder backward rebuilds the row from d_rgb),
        // which removes the [R,S,32] gradient tensor -- 403 MB written here and read again there per 128^2 x 192 image.
        if (!d_rgb) {                     // only the depth map is differentiated (SPI's depth branch): no colour traffic at all
#pragma unroll
            for (int c = 0; c < NCH; ++c) L.q[c * 64 + lane] = 0.f;
        } else {
            float* dcol_r = d_colors ? d_colors + r * S_store * 32 : nullptr;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int k = it * 8 + rg;
                const float4 c4 = ring[it % RING];
                const float part = group8_sum(g4.x * c4.x + g4.y * c4.y + g4.z * c4.z + g4.w * c4.w);
                L.q[k] = part;                                           // (the 8 lanes of a row group write the same value)
                if (it + RING < NIT) { asm volatile("" ::: "memory"); ring[it % RING] = cload(it + RING); }      // (the fence keeps the compiler from hoisting every load to the top again)
                if (d_colors && k < S) {
                    const float v = (k > 0 ? L.w[k - 1] : 0.f) + L.w[k];
                    const f32x4_t o = {g4.x * v, g4.y * v, g4.z * v, g4.w * v};
                    __builtin_nontemporal_store(o, reinterpret_cast<f32x4_t*>(dcol_r + (unsigned)(L.row[k] * 32 + sub * 4)));
                }
            }
            if (d_color_scale) {
                float* dcs_r = d_color_scale + r * S_store;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int k = c * 64 + lane;
                    if (k < S) dcs_r[(unsigned)L.row[k]] = (k > 0 ? L.w[k - 1] : 0.f) + L.w[k];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // dL/dw_k
        float depth_scale = 0.f, D = 0.f;
        if (d_depth) {
            D = dnum / wsum;
            const bool ok = (D == D) && (D >= clamp2[0]) && (D <= clamp2[1]) && (fabsf(D) != INFINITY);
            depth_scale = ok ? lane_bcast(ddep_lane, 8 * j) / wsum : 0.f;      // empty rays: the reference yields NaN here; we give 0
            if (!ok) D = 0.f;
        }
        float gsum = 0.f;
        if (white_back) gsum = -2.f * wave_sum(rg == 0 ? (g4.x + g4.y + g4.z + g4.w) : 0.f);
        float gw_[NCH];                   // g_k * w_k
        float g[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k = c * 64 + lane;
            float gk = 0.f;
            if (k < S - 1) {
                gk = L.q[k] + L.q[k + 1] + depth_scale * ((L.dep[k] + L.dep[k + 1]) / 2.f - D) + gsum;
                if (d_weights) gk += (d_weights + r * (S - 1))[(unsigned)k];
            }
            g[c] = gk;
            gw_[c] = gk * alpha[c] * trans[c];
        }
        // suffix (exclusive) sums of g_m w_m over m > k
        float total = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) total += gw_[c];
        total = wave_sum(total);
        __builtin_amdgcn_wave_barrier();                                   // every read of q above precedes its reuse below
        float before = 0.f;               // sum over earlier chunks
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k = c * 64 + lane;
            const float incl = wave_scan_add(gw_[c]) + before;              // sum_{m <= k}
            const float suffix = total - incl;                              // sum_{m > k}
            before = lane_bcast(incl, 63);
            float ds = 0.f;
            if (k < S - 1) {
                const float om = 1.f - alpha[c] + 1e-10f;
                const float da = g[c] * trans[c] - suffix / om;
                const float dsh = da * delta[c] * (1.f - alpha[c]);             // d alpha / d sigma_hat = delta * exp(-sigma_hat delta)
                ds = dsh * sigmoid_f(smid[c] - 1.f) * 0.5f;                     // softplus'(x-1) and the midpoint's 1/2
            }
            L.q[k] = ds;                                                        // reuse q: contribution of interval k
        }
        __builtin_amdgcn_wave_barrier();
        float* dden_r = d_densities + r * S_store;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k = c * 64 + lane;
            if (k < S) dden_r[(unsigned)L.row[k]] = (k < S - 1 ? L.q[k] : 0.f) + (k > 0 ? L.q[k - 1] : 0.f);
        }
        if (jn < 0) break;
        j = jn;
        __builtin_amdgcn_wave_barrier();                                   // the next ray's staging overwrites this ray's LDS rows
    }
}

// ------------------------------------------------------------------------------------------------
// sample_importance / sample_pdf (renderer.py:194-253): one wave per ray.
// ------------------------------------------------------------------------------------------------
// Stable rank order of the two rank sorts below: does `o` (at an earlier position iff `o_first`) come before `v`?  A total order -- NaNs sort last,
// among themselves by position, like torch.sort -- so that the ranks of a row are ALWAYS a permutation.  (With the plain `o < v || (o == v && ...)`
// every NaN got rank 0: the slots at the end of the row were never written, and the permutation the decoder backward indexes with kept whatever the
// buffer held before -- a diverged ray turned into an out-of-range read instead of a NaN in the loss.)
__device__ __forceinline__ int rank_before(float o, float v, bool o_first) {
    const bool on = o != o, vn = v != v;
    if (on || vn) return (!on && vn) || (on && vn && o_first);
    return (o < v) || (o == v && o_first);
}

__global__ void __launch_bounds__(256) importance_kernel(const float* __restrict__ depths, const float* __restrict__ weights,
                                                         const float* __restrict__ u, int64_t R, int S, int Sf,
                                                         float* __restrict__ fine, int sort_out) {
    __shared__ float s_w[4][MAXS + 2];
    __shared__ float s_cdf[4][MAXS];
    __shared__ float s_bin[4][MAXS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= R) return;
    const int L = S - 1;                       // number of coarse weights
    float* w = s_w[wave]; float* cdf = s_cdf[wave]; float* bin = s_bin[wave];
    for (int k = lane; k < L; k += 64) w[k + 1] = weights[r * L + k];
    if (lane == 0) { w[0] = -INFINITY; w[L + 1] = -INFINITY; }
    for (int k = lane; k < L; k += 64) bin[k] = 0.5f * (depths[r * S + k] + depths[r * S + k + 1]);
    __builtin_amdgcn_wave_barrier();
    // smoothed a_i = (max(w_{i-1},w_i) + max(w_i,w_{i+1}))/2 + 0.01 for i in [0,L); pdf uses a_1..a_{L-2}
    const int NP = L - 2;                      // pdf entries
    float tot = 0.f;
    float pv[(MAXS + 63) / 64];
#pragma unroll
    for (int c = 0; c < (MAXS + 63) / 64; ++c) {
        const int j = c * 64 + lane;           // pdf index -> a_{j+1}
        float v = 0.f;
        if (j < NP) {
            const int i = j + 1;
            const float m0 = fmaxf(w[i], w[i + 1]);       // w[] is shifted by one: w[i] = weight_{i-1}
            const float m1 = fmaxf(w[i + 1], w[i + 2]);
            v = 0.5f * (m0 + m1) + 0.01f + 1e-5f;
        }
        pv[c] = v; tot += v;
    }
    tot = wave_sum(tot);
    float carry = 0.f;
#pragma unroll
    for (int c = 0; c < (MAXS + 63) / 64; ++c) {
        const int j = c * 64 + lane;
        const float incl = wave_scan_add(pv[c] / tot) + carry;
        if (j < NP) cdf[j + 1] = incl;
        carry = lane_bcast(incl, 63);
    }
    if (lane == 0) cdf[0] = 0.f;
    __builtin_amdgcn_wave_barrier();
    const int NC = NP + 1;                     // cdf entries
    float* tv = w;                             // the smoothed weights are no longer needed: reuse as the sample buffer
    for (int j = lane; j < Sf; j += 64) {
        const float uu = u[r * Sf + j];
        int lo = 0, hi = NC;                   // searchsorted(right=True): first index with cdf > u
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= uu) lo = mid + 1; else hi = mid; }
        const int below = max(lo - 1, 0), above = min(lo, NP);
        float den = cdf[above] - cdf[below];
        if (den < 1e-5f) den = 1.f;
        const float tval = bin[below] + (uu - cdf[below]) / den * (bin[above] - bin[below]);
        if (sort_out) tv[j] = tval; else fine[r * Sf + j] = tval;
    }
    if (!sort_out) return;
    // Emit the samples in ascending order (stable rank sort).  The reference keeps them in draw order and sorts
    // coarse+fine together afterwards (renderer.py:157-163); the merged result is the same set either way, and
    // sorted fine rows make the final march read two monotone streams through its permutation.
    __builtin_amdgcn_wave_barrier();
    bool nan_here = false;
    for (int j = lane; j < Sf; j += 64) nan_here = nan_here || (tv[j] != tv[j]);
    if (!__any(nan_here)) {                                  // the normal case (wave-uniform): plain comparisons, 2 instead of ~7 operations per pair
        for (int j = lane; j < Sf; j += 64) {
            const float v = tv[j];
            int rank = 0;
            for (int m2 = 0; m2 < Sf; ++m2) { const float o = tv[m2]; rank += (o < v) || (o == v && m2 < j); }
            fine[r * Sf + rank] = v;
        }
        return;
    }
    for (int j = lane; j < Sf; j += 64) {
        const float v = tv[j];
        int rank = 0;
        for (int m2 = 0; m2 < Sf; ++m2) rank += rank_before(tv[m2], v, m2 < j);
        fine[r * Sf + rank] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// unify_samples' sort (renderer.py:157-163) as a stable rank sort: one wave per ray.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) merge_sort_kernel(const float* __restrict__ coarse, const float* __restrict__ fine,
                                                         int64_t R, int Sc, int Sf, float* __restrict__ sorted,
                                                         int32_t* __restrict__ perm) {
    __shared__ float s_d[4][MAXS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= R) return;
    const int S = Sc + Sf;
    float* d = s_d[wave];
    for (int k = lane; k < S; k += 64) d[k] = (k < Sc) ? coarse[r * Sc + k] : fine[r * Sf + (k - Sc)];
    __builtin_amdgcn_wave_barrier();
    // The renderer hands over two ascending runs (stratified coarse samples, importance samples emitted in order): then the stable
    // rank of an element is its index in its own run plus a binary search in the other one (coarse before fine on ties) -- 7 probes
    // instead of S comparisons.  Anything else (NaNs, unsorted runs) takes the generic O(S^2) rank sort below.
    bool asc = true;
    for (int k = lane; k < S; k += 64) asc = asc && (k + 1 >= S || k + 1 == Sc || d[k] <= d[k + 1]);
    if (__all(asc)) {
        for (int k = lane; k < S; k += 64) {
            const float v = d[k];
            const bool is_c = k < Sc;
            int lo = is_c ? Sc : 0, hi = is_c ? S : Sc;                 // search the OTHER run
            while (lo < hi) {                                            // coarse element: #fine < v;  fine element: #coarse <= v
                const int mid = (lo + hi) >> 1;
                const float o = d[mid];
                if (is_c ? (o < v) : (o <= v)) lo = mid + 1; else hi = mid;
            }
            const int rank = is_c ? k + (lo - Sc) : (k - Sc) + lo;
            sorted[r * S + rank] = v;
            perm[r * S + rank] = k;
        }
        return;
    }
    for (int k = lane; k < S; k += 64) {
        const float v = d[k];
        int rank = 0;
        for (int m = 0; m < S; ++m) rank += rank_before(d[m], v, m < k);
        sorted[r * S + rank] = v;
        perm[r * S + rank] = k;
    }
}

// ================================================================================================
// C ABI
// ================================================================================================
// The OSG decoder's four parameter tensors times their FullyConnectedLayer gains (networks_stylegan2.py:114-127 folds `weight_gain` /
// `bias_gain` into every forward) in ONE launch: w1 [64,32] -> transposed [32,64] (forward operands) or as stored (gradients), b1 [64],
// w2 [33,64], b2 [33].  Four `mul`s + a transposing copy per render forward and four per backward were 9 of the ~5 us ATen launches.
__global__ void __launch_bounds__(256) decoder_gains_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, float g_w1, float g_b1, float g_w2, float g_b2,
                                                            float* __restrict__ o_w1, float* __restrict__ o_b1, float* __restrict__ o_w2,
                                                            float* __restrict__ o_b2, int transpose_w1) {
    constexpr int N1 = DEC_HID * DEC_IN, N2 = DEC_HID, N3 = DEC_OUT * DEC_HID, N4 = DEC_OUT;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < N1 + N2 + N3 + N4; e += gridDim.x * 256) {
        if (e < N1) {
            const int j = e / DEC_IN, i = e - j * DEC_IN;                    // w1[j][i]
            o_w1[transpose_w1 ? i * DEC_HID + j : e] = w1[e] * g_w1;
        } else if (e < N1 + N2) o_b1[e - N1] = b1[e - N1] * g_b1;
        else if (e < N1 + N2 + N3) o_w2[e - N1 - N2] = w2[e - N1 - N2] * g_w2;
        else o_b2[e - N1 - N2 - N3] = b2[e - N1 - N2 - N3] * g_b2;
    }
}

// The decoder's four gradient accumulators start at zero.  A caller that carves them from one buffer in the order dW1 | db1 | dW2 | db2
// (the host wrapper does) gets one fill launch instead of four.
static void zero_decoder_grads(float* dw1, float* db1, float* dw2, float* db2, hipStream_t st) {
    if (db1 == dw1 + DEC_HID * DEC_IN && dw2 == db1 + DEC_HID && db2 == dw2 + DEC_OUT * DEC_HID) {
        spi_zero_async(dw1, DEC_HID * DEC_IN + DEC_HID + DEC_OUT * DEC_HID + DEC_OUT, st);
        return;
    }
    spi_zero_async(dw1, DEC_HID * DEC_IN, st); spi_zero_async(db1, DEC_HID, st);
    spi_zero_async(dw2, DEC_OUT * DEC_HID, st); spi_zero_async(db2, DEC_OUT, st);
}

extern "C" {

void spi_debug_set(int flags) { g_spi_debug = flags; }      /* profiling experiments only; 0 in normal operation */

int spi_ray_sampler(const float* cam2world, const float* intrinsics, int N, int res, float* ray_o, float* ray_d,
                    spi_stream_t stream) {
    SPI_REQUIRE(cam2world && intrinsics && ray_o && ray_d && N > 0 && res > 0, "spi_ray_sampler: bad argument");
    const int64_t total = (int64_t)N * res * res;
    hipLaunchKernelGGL(ray_sampler_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, as_stream(stream),
                       cam2world, intrinsics, N, res, ray_o, ray_d);
    SPI_LAUNCH_CHECK("spi_ray_sampler");
    return SPI_OK;
}

int spi_coarse_depths(const float* xi, int64_t n_rays, int S, float ray_start, float ray_end, float* depths,
                      spi_stream_t stream) {
    SPI_REQUIRE(xi && depths && n_rays > 0 && S > 1, "spi_coarse_depths: bad argument");
    const int64_t total = n_rays * S;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(total, 256), 4096);
    hipLaunchKernelGGL(coarse_depths_kernel, dim3(grid), dim3(256), 0, as_stream(stream), xi, total, S, ray_start, ray_end, depths);
    SPI_LAUNCH_CHECK("spi_coarse_depths");
    return SPI_OK;
}

static int relayout(const float* src, float* dst, int NP, int C, int H, int W, bool to_nhwc, spi_stream_t stream) {
    SPI_REQUIRE(src && dst && NP > 0 && C > 0 && H > 0 && W > 0, "spi relayout: bad argument");
    const int64_t HW = (int64_t)H * W;
    dim3 grid((unsigned)ceil_div64(HW, 32), (unsigned)((C + 31) / 32), (unsigned)NP);
    if (to_nhwc) hipLaunchKernelGGL(relayout_kernel<true>, grid, dim3(256), 0, as_stream(stream), src, dst, C, HW);
    else hipLaunchKernelGGL(relayout_kernel<false>, grid, dim3(256), 0, as_stream(stream), src, dst, C, HW);
    SPI_LAUNCH_CHECK("spi relayout");
    return SPI_OK;
}
int spi_nchw_to_nhwc(const float* src, float* dst, int NP, int C, int H, int W, spi_stream_t s) { return relayout(src, dst, NP, C, H, W, true, s); }
int spi_nhwc_to_nchw(const float* src, float* dst, int NP, int C, int H, int W, spi_stream_t s) { return relayout(src, dst, NP, C, H, W, false, s); }

static int fill_decode_args(DecodeArgs& a, const float* planes, const float* coords, const float* ray_o, const float* ray_d,
                            const float* depths, const float* w1, const float* b1, const float* w2, const float* b2,
                            int N, int64_t P, int S, int H, int W, float box_warp, int out_S, int out_off) {
    SPI_REQUIRE(planes && w1 && b1 && w2 && b2, "spi_triplane_decode: null tensor");
    SPI_REQUIRE(out_S == 0 || (ray_o != nullptr && out_off >= 0 && out_off + S <= out_S), "spi_triplane_decode: bad out_S/out_off");
    SPI_REQUIRE(N > 0 && P > 0 && H > 0 && W > 0 && box_warp > 0.f, "spi_triplane_decode: bad size");
    SPI_REQUIRE((int64_t)N * 3 * H * W * DEC_IN * 4 < ((int64_t)1 << 31), "spi_triplane_decode: the planes tensor must be < 2 GiB (one buffer descriptor)");
    if (ray_o == nullptr) SPI_REQUIRE(coords != nullptr, "spi_triplane_decode: need coords or rays");
    else SPI_REQUIRE(ray_d && depths && S > 0 && P % S == 0, "spi_triplane_decode: rays mode needs ray_d, depths, P %% S == 0");
    a.planes = planes; a.coords = coords; a.ray_o = ray_o; a.ray_d = ray_d; a.depths = depths;
    a.N = N; a.P = P; a.S = S > 0 ? S : 1; a.H = H; a.W = W; a.out_S = out_S; a.out_off = out_off;
    a.scale = 2.f / box_warp;
    return SPI_OK;
}

int spi_sample_from_planes_fwd(const float* planes_nhwc, const float* coords, int N, int64_t P, int H, int W, float box_warp, float* out,
                               spi_stream_t stream) {
    SPI_REQUIRE(planes_nhwc && coords && out, "spi_sample_from_planes_fwd: null tensor");
    SPI_REQUIRE(N > 0 && P > 0 && H > 0 && W > 0 && box_warp > 0.f, "spi_sample_from_planes_fwd: bad size");
    SPI_REQUIRE((int64_t)N * 3 * H * W * DEC_IN * 4 < ((int64_t)1 << 31), "spi_sample_from_planes_fwd: the planes tensor must be < 2 GiB (one buffer descriptor)");
    const int64_t total = (int64_t)N * 3 * P;
    hipLaunchKernelGGL(sample_planes_fwd_kernel, dim3((unsigned)ceil_div64(total, 32)), dim3(256), 0, as_stream(stream), planes_nhwc, coords, N, P, H, W,
                       2.f / box_warp, out);
    SPI_LAUNCH_CHECK("spi_sample_from_planes_fwd");
    return SPI_OK;
}

int spi_sample_from_planes_bwd(const float* d_out, const float* coords, int N, int64_t P, int H, int W, float box_warp, float* d_planes_nhwc,
                               spi_stream_t stream) {
    SPI_REQUIRE(d_out && coords && d_planes_nhwc, "spi_sample_from_planes_bwd: null tensor");
    SPI_REQUIRE(N > 0 && P > 0 && H > 0 && W > 0 && box_warp > 0.f, "spi_sample_from_planes_bwd: bad size");
    const int64_t total = (int64_t)N * 3 * P;
    hipLaunchKernelGGL(sample_planes_bwd_kernel, dim3((unsigned)ceil_div64(total, 32)), dim3(256), 0, as_stream(stream), d_out, coords, N, P, H, W,
                       2.f / box_warp, d_planes_nhwc);
    SPI_LAUNCH_CHECK("spi_sample_from_planes_bwd");
    return SPI_OK;
}

int spi_triplane_decode_fwd(const float* planes_nhwc, const float* coords, const float* ray_o, const float* ray_d,
                            const float* depths, const float* w1, const float* b1, const float* w2, const float* b2,
                            int N, int64_t P, int S, int H, int W, float box_warp, int out_S, int out_off, float* rgb,
                            float* sigma, spi_stream_t stream) {
    DecodeArgs a;
    int rc = fill_decode_args(a, planes_nhwc, coords, ray_o, ray_d, depths, w1, b1, w2, b2, N, P, S, H, W, box_warp, out_S, out_off);
    if (rc) return rc;
    SPI_REQUIRE(sigma, "spi_triplane_decode_fwd: null output");          // rgb may be NULL: densities only
    const int64_t total = (int64_t)N * P;
    const int64_t tiles = ceil_div64(total, DT);
    // the split-bf16 matrix-core kernel keeps a point's output row in 32 bits; spi_debug_set(256) selects the vector-ALU kernel (A/B measurements)
    if (!(g_spi_debug & 256) && (out_S == 0 ? total : (total / a.S) * (int64_t)out_S) < ((int64_t)1 << 31))
        hipLaunchKernelGGL(decode_fwd_mfma_kernel, dim3((unsigned)std::min<int64_t>(tiles, FWD_MFMA_GRID)), dim3(DT), 0, as_stream(stream), a, w1, b1, w2, b2, rgb,
                           sigma, tiles);
    else
        hipLaunchKernelGGL(decode_fwd_kernel, dim3((unsigned)tiles), dim3(DT), 0, as_stream(stream), a, w1, b1, w2, b2, rgb, sigma);
    SPI_LAUNCH_CHECK("spi_triplane_decode_fwd");
    return SPI_OK;
}

int spi_triplane_decode_bwd(const float* planes_nhwc, const float* coords, const float* ray_o, const float* ray_d,
                            const float* depths, const float* w1, const float* b1, const float* w2, const float* b2,
                            const float* d_rgb, const float* d_sigma, int N, int64_t P, int S, int H, int W, float box_warp,
                            int out_S, int out_off, float* d_planes_nhwc, float* dump_act, spi_stream_t stream) {
    DecodeArgs a;
    int rc = fill_decode_args(a, planes_nhwc, coords, ray_o, ray_d, depths, w1, b1, w2, b2, N, P, S, H, W, box_warp, out_S, out_off);
    if (rc) return rc;
    SPI_REQUIRE(d_rgb && d_sigma && d_planes_nhwc, "spi_triplane_decode_bwd: null tensor");
    const int64_t total = (int64_t)N * P;
    hipLaunchKernelGGL(decode_bwd_kernel, dim3((unsigned)ceil_div64(total, DT)), dim3(DT), 0, as_stream(stream), a, w1, b1, w2,
                       b2, d_rgb, d_sigma, d_planes_nhwc, dump_act);
    SPI_LAUNCH_CHECK("spi_triplane_decode_bwd");
    return SPI_OK;
}

int spi_triplane_decode_bwd_sorted(const float* planes_nhwc, const float* ray_o, const float* ray_d, const float* depths_sorted,
                                   const int32_t* perm, const float* w1t, const float* b1, const float* w2, const float* b2,
                                   const float* d_rgb, const float* d_rgb_scale, const float* colors, const float* d_sigma, int N, int M, int S, int ray_w,
                                   int H, int W, float box_warp, float* d_planes_nhwc, float* workspace, float* dw1, float* db1,
                                   float* dw2, float* db2, const int32_t* ray_active, spi_stream_t stream) {
    SPI_REQUIRE(planes_nhwc && ray_o && ray_d && depths_sorted && w1t && b1 && w2 && b2 && d_sigma && d_planes_nhwc && workspace,
                "spi_triplane_decode_bwd_sorted: null tensor");
    SPI_REQUIRE(N > 0 && M > 0 && S > 0 && H > 0 && W > 0 && box_warp > 0.f && ray_w > 0, "spi_triplane_decode_bwd_sorted: bad size");
    SPI_REQUIRE((int64_t)N * 3 * H * W * DEC_IN * 4 < ((int64_t)1 << 31), "spi_triplane_decode_bwd_sorted: the planes tensor must be < 2 GiB (one buffer descriptor)");
    const bool wgrad = dw1 != nullptr;
    SPI_REQUIRE(!wgrad || (db1 && dw2 && db2), "spi_triplane_decode_bwd_sorted: the four decoder gradient outputs come together");
    SPI_REQUIRE(d_rgb || !d_rgb_scale, "spi_triplane_decode_bwd_sorted: d_rgb_scale given without the per-ray d_rgb");
    SPI_REQUIRE(!d_rgb || colors, "spi_triplane_decode_bwd_sorted: d_rgb needs the forward's colour rows (colors)");
    TiledArgs a;
    a.planes = planes_nhwc; a.ray_o = ray_o; a.ray_d = ray_d; a.depths = depths_sorted; a.perm = perm; a.ray_active = ray_active;
    a.N = N; a.M = M; a.S = S; a.H = H; a.W = W; a.scale = 2.f / box_warp; a.ray_w = ray_w;
    a.patch2d = (M % ray_w == 0) && (ray_w % 8 == 0) && ((M / ray_w) % 8 == 0);
    a.patches = a.patch2d ? M / 64 : (M + 63) / 64;
    a.kchunks = (S + 3) / 4;                                   // 64 * S points per patch / 256 per tile
    a.dbg = g_spi_debug;
    SPI_REQUIRE(S <= 256, "spi_triplane_decode_bwd_sorted: S <= 256 (sample index packed in 8 bits), got %d", S);
    const int64_t tiles = (int64_t)N * a.patches * a.kchunks;
    SPI_REQUIRE(tiles < (int64_t)1 << 31, "spi_triplane_decode_bwd_sorted: too many tiles");
    a.tiles = (int)tiles;
    const unsigned grid = (unsigned)std::min<int64_t>(tiles, BWD_MAX_GRID);
    hipStream_t st = as_stream(stream);
    float* frag = workspace;
    float* part = workspace + FRAG_TOTAL + 32;                  // (FRAG_TOTAL is a multiple of 64 floats; keep every region 128-byte aligned)
    int32_t* count = reinterpret_cast<int32_t*>(part + (int64_t)grid * 4 * PART_ROW);
    uint16_t* order = reinterpret_cast<uint16_t*>(count + (((int64_t)N * a.patches + 31) & ~(int64_t)31));
    const int64_t order_floats = (((int64_t)N * a.patches * 64 * S + 1) / 2 + 31) & ~(int64_t)31;
    a.dfeat = reinterpret_cast<float*>(order) + order_floats;
    int4* bbox = reinterpret_cast<int4*>(a.dfeat + (int64_t)N * a.patches * 64 * S * DEC_IN);      // footprint of every (tile, plane): tile_bbox_kernel
    a.order = order; a.count = count;
    hipLaunchKernelGGL(bin_points_kernel, dim3((unsigned)(N * a.patches)), dim3(256), 0, st, depths_sorted, ray_active, M, S, ray_w, a.patch2d,
                       a.patches, order, count);
    hipLaunchKernelGGL(decoder_frag_kernel, dim3(13), dim3(64), 0, st, w1t, w2, frag);
#define SPI_BWD_LAUNCH(WG, RGBF) hipLaunchKernelGGL((decode_bwd_tiled_kernel<WG, RGBF>), dim3(grid), dim3(DT), 0, st, a, frag, b1, b2, d_rgb, d_rgb_scale, colors, d_sigma, part)
    if (wgrad) {
        if (d_rgb) SPI_BWD_LAUNCH(true, true); else SPI_BWD_LAUNCH(true, false);
        zero_decoder_grads(dw1, db1, dw2, db2, st);
        hipLaunchKernelGGL(decoder_partial_reduce_kernel, dim3((PART_DB2 + 33 + 255) / 256, 32), dim3(256), 0, st, part, (int)grid * 4, dw1, db1, dw2, db2);
    } else {
        if (d_rgb) SPI_BWD_LAUNCH(false, true); else SPI_BWD_LAUNCH(false, false);
    }
#undef SPI_BWD_LAUNCH
    if (!(a.dbg & 1)) {                                        // (tools/bench_render.py: 1 = decoder only)
        hipLaunchKernelGGL(tile_bbox_kernel, dim3((unsigned)((tiles * 3 + 255) / 256)), dim3(256), 0, st, a, bbox);
        hipLaunchKernelGGL(plane_scatter_kernel, dim3((unsigned)(N * a.patches), 3, 2), dim3(SCT), 0, st, a, bbox, d_planes_nhwc);
    }
    SPI_LAUNCH_CHECK("spi_triplane_decode_bwd_sorted");
    return SPI_OK;
}

int64_t spi_triplane_decode_bwd_sorted_ws(int N, int M, int S, int ray_w) {
    const bool p2d = (M % ray_w == 0) && (ray_w % 8 == 0) && ((M / ray_w) % 8 == 0);
    const int64_t patches = p2d ? M / 64 : (M + 63) / 64;
    const int64_t tiles = (int64_t)N * patches * ((S + 3) / 4);
    const int64_t counts = ((int64_t)N * patches + 31) & ~(int64_t)31;               // int32 per patch (regions kept 128-byte aligned)
    const int64_t order = (((int64_t)N * patches * 64 * S + 1) / 2 + 31) & ~(int64_t)31;     // uint16 per point, in floats
    const int64_t dfeat = (int64_t)N * patches * 64 * S * DEC_IN;                    // the d_feat rows handed from the decoder kernel to the scatter kernel
    return FRAG_TOTAL + 32 + std::min<int64_t>(tiles, BWD_MAX_GRID) * 4 * PART_ROW + counts + order + dfeat + tiles * 3 * 4;      // + the footprint boxes (int4 per tile and plane)
}

int spi_decoder_wgrad(const float* dump, int64_t cols, float* dw1, float* db1, float* dw2, float* db2, spi_stream_t stream) {
    SPI_REQUIRE(dump && dw1 && db1 && dw2 && db2 && cols > 0, "spi_decoder_wgrad: bad argument");
    SPI_REQUIRE(cols % 4 == 0 && ((uintptr_t)dump & 15) == 0, "spi_decoder_wgrad: dump must be 16-byte aligned with cols %% 4 == 0");
    hipStream_t st = as_stream(stream);
    zero_decoder_grads(dw1, db1, dw2, db2, st);
    int64_t chunk = (cols + 1023) / 1024;                    // ~1024 blocks
    chunk = std::max<int64_t>(256, ((chunk + WG_BK - 1) / WG_BK) * WG_BK);
    const unsigned grid = (unsigned)ceil_div64(cols, chunk);
    hipLaunchKernelGGL(decoder_wgrad_kernel, dim3(grid), dim3(256), 0, st, dump, cols, chunk, dw1, db1, dw2, db2);
    SPI_LAUNCH_CHECK("spi_decoder_wgrad");
    return SPI_OK;
}

int spi_minmax(const float* x, int64_t n, float* out2, spi_stream_t stream) {
    SPI_REQUIRE(x && out2 && n > 0, "spi_minmax: bad argument");
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, as_stream(stream), out2);
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(n, 256 * 8), 512);
    hipLaunchKernelGGL(minmax_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, n, out2);
    SPI_LAUNCH_CHECK("spi_minmax");
    return SPI_OK;
}

int spi_raymarch_fwd(const float* colors, const float* densities, const float* depths, const int32_t* perm,
                     const float* clamp2, int64_t R, int S, int S_store, int C, int white_back, float* rgb, float* depth,
                     float* weights, float* wsum, spi_stream_t stream) {
    SPI_REQUIRE(S_store >= S && S_store <= MAXS, "spi_raymarch_fwd: need S <= S_store <= %d (the ray's rows are staged in LDS), got S_store = %d", MAXS, S_store);
    SPI_REQUIRE(densities && depths && R > 0, "spi_raymarch_fwd: null tensor");
    SPI_REQUIRE(S >= 2 && S <= MAXS, "spi_raymarch_fwd: need 2 <= S <= %d, got %d", MAXS, S);
    SPI_REQUIRE(C == 32, "spi_raymarch_fwd: only C = 32 feature channels are supported, got %d", C);
    SPI_REQUIRE(rgb == nullptr || colors != nullptr, "spi_raymarch_fwd: rgb requested without colors");
    SPI_REQUIRE(depth == nullptr || clamp2 != nullptr, "spi_raymarch_fwd: depth requested without clamp range");
    dim3 grid((unsigned)ceil_div64(R, RM_WAVES)), block(64 * RM_WAVES);
    const int nch = (S + 63) / 64;
#define LAUNCH_FWD(NCH) hipLaunchKernelGGL(raymarch_fwd_kernel<NCH>, grid, block, 0, as_stream(stream), colors, densities, \
                                          depths, perm, clamp2, R, S, S_store, white_back, rgb, depth, weights, wsum)
    switch (nch) { case 1: LAUNCH_FWD(1); break; case 2: LAUNCH_FWD(2); break; case 3: LAUNCH_FWD(3); break; default: LAUNCH_FWD(4); }
#undef LAUNCH_FWD
    SPI_LAUNCH_CHECK("spi_raymarch_fwd");
    return SPI_OK;
}

int spi_raymarch_bwd(const float* colors, const float* densities, const float* depths, const int32_t* perm,
                     const float* clamp2, const float* d_rgb, const float* d_depth, const float* d_weights, int64_t R,
                     int S, int S_store, int C, int white_back, float* d_colors, float* d_color_scale, float* d_densities,
                     int32_t* ray_active, spi_stream_t stream) {
    SPI_REQUIRE(S_store >= S && S_store <= MAXS, "spi_raymarch_bwd: need S <= S_store <= %d (the ray's rows are staged in LDS), got S_store = %d", MAXS, S_store);
    SPI_REQUIRE(densities && depths && d_densities && R > 0 && (d_rgb == nullptr || colors != nullptr), "spi_raymarch_bwd: null tensor");
    SPI_REQUIRE(S >= 2 && S <= MAXS, "spi_raymarch_bwd: need 2 <= S <= %d, got %d", MAXS, S);
    SPI_REQUIRE(C == 32, "spi_raymarch_bwd: only C = 32 feature channels are supported, got %d", C);
    SPI_REQUIRE(d_depth == nullptr || clamp2 != nullptr, "spi_raymarch_bwd: d_depth given without clamp range");
    SPI_REQUIRE(d_rgb == nullptr || d_colors != nullptr || d_color_scale != nullptr,
                "spi_raymarch_bwd: d_rgb given without a d_colors or d_color_scale output");
    // rays per wave: 8 (one wave-instruction of gradient rows) unless that leaves fewer than 2 048 waves (256 CUs x 4 SIMDs x 2)
    const int rpw = (int)std::min<int64_t>(RM_RPW, std::max<int64_t>(1, R / 2048));
    dim3 grid((unsigned)ceil_div64(ceil_div64(R, rpw), RM_WAVES)), block(64 * RM_WAVES);
    const int nch = (S + 63) / 64;
#define LAUNCH_BWD(NCH) hipLaunchKernelGGL(raymarch_bwd_kernel<NCH>, grid, block, 0, as_stream(stream), colors, densities, \
                                          depths, perm, clamp2, d_rgb, d_depth, d_weights, R, S, S_store, white_back, rpw, d_colors, d_color_scale, d_densities, ray_active)
    switch (nch) { case 1: LAUNCH_BWD(1); break; case 2: LAUNCH_BWD(2); break; case 3: LAUNCH_BWD(3); break; default: LAUNCH_BWD(4); }
#undef LAUNCH_BWD
    SPI_LAUNCH_CHECK("spi_raymarch_bwd");
    return SPI_OK;
}

int spi_importance_sample(const float* depths, const float* weights, const float* u, int64_t R, int S, int Sf, float* fine,
                          int sort_out, spi_stream_t stream) {
    SPI_REQUIRE(depths && weights && u && fine && R > 0 && Sf > 0, "spi_importance_sample: bad argument");
    SPI_REQUIRE(S >= 4 && S <= MAXS && Sf <= MAXS, "spi_importance_sample: need 4 <= S <= %d and Sf <= %d", MAXS, MAXS);
    hipLaunchKernelGGL(importance_kernel, dim3((unsigned)ceil_div64(R, 4)), dim3(256), 0, as_stream(stream), depths, weights, u,
                       R, S, Sf, fine, sort_out);
    SPI_LAUNCH_CHECK("spi_importance_sample");
    return SPI_OK;
}

int spi_merge_sort_depths(const float* coarse, const float* fine, int64_t R, int Sc, int Sf, float* sorted, int32_t* perm,
                          spi_stream_t stream) {
    SPI_REQUIRE(coarse && fine && sorted && perm && R > 0 && Sc > 0 && Sf > 0, "spi_merge_sort_depths: bad argument");
    SPI_REQUIRE(Sc + Sf <= MAXS, "spi_merge_sort_depths: Sc + Sf must be <= %d", MAXS);
    hipLaunchKernelGGL(merge_sort_kernel, dim3((unsigned)ceil_div64(R, 4)), dim3(256), 0, as_stream(stream), coarse, fine, R, Sc,
                       Sf, sorted, perm);
    SPI_LAUNCH_CHECK("spi_merge_sort_depths");
    return SPI_OK;
}

int spi_decoder_gains(const float* w1, const float* b1, const float* w2, const float* b2, float g_w1, float g_b1, float g_w2, float g_b2,
                      float* o_w1, float* o_b1, float* o_w2, float* o_b2, int transpose_w1, spi_stream_t stream) {
    SPI_REQUIRE(w1 && b1 && w2 && b2 && o_w1 && o_b1 && o_w2 && o_b2, "spi_decoder_gains: null tensor");
    hipLaunchKernelGGL(decoder_gains_kernel, dim3(17), dim3(256), 0, as_stream(stream), w1, b1, w2, b2, g_w1, g_b1, g_w2, g_b2, o_w1, o_b1, o_w2, o_b2, transpose_w1);
    SPI_LAUNCH_CHECK("spi_decoder_gains");
    return SPI_OK;
}

}  // extern "C"
