/*
 * oracle/c/renderer_ref.c -- scalar C restatement of the volumetric renderer's arithmetic.
 * TEST INFRASTRUCTURE (see oracle/__init__.py): a second, independent checker beside the torch oracle,
 * written as plain loops so that every formula of SURVEY.md Appendix A is visible once more:
 *   A3 tri-plane gather     eg3d/training/volumetric_rendering/renderer.py:23-65
 *   A4 OSG decoder          eg3d/training/triplane.py:112-135, networks_stylegan2.py:114-127
 *   A5 ray march            eg3d/training/volumetric_rendering/ray_marcher.py:25-57
 *   A6 importance sampling  eg3d/training/volumetric_rendering/renderer.py:194-253
 * Pinned against the reference's golden vectors in tests/test_oracle_cpu.py.  Never linked into the product.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

/* planes [N,3,C,H,W] (NCHW as the backbone emits them), coords [N,P,3] -> rgb [N,P,32], sigma [N,P] */
void ref_gather_decode(const float* planes, const float* coords, int N, int64_t P, int C, int H, int W, float box_warp,
                       const float* w1 /*[64,C], gained*/, const float* b1, const float* w2 /*[33,64]*/, const float* b2,
                       float* rgb, float* sigma) {
    for (int n = 0; n < N; ++n)
        for (int64_t p = 0; p < P; ++p) {
            const float* q = coords + ((int64_t)n * P + p) * 3;
            const float s = 2.f / box_warp;
            const float x = q[0] * s, y = q[1] * s, z = q[2] * s;
            const float gxs[3] = {x, x, z}, gys[3] = {y, z, x};          /* planes sampled at (x,y), (x,z), (z,x) */
            float f[64];
            for (int c = 0; c < C; ++c) f[c] = 0.f;
            for (int pl = 0; pl < 3; ++pl) {
                const float ix = ((gxs[pl] + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gys[pl] + 1.f) * (float)H - 1.f) * 0.5f;
                const float fx = floorf(ix), fy = floorf(iy);
                const int x0 = (int)fx, y0 = (int)fy;
                const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
                for (int c = 0; c < C; ++c) {
                    const float* pc = planes + (((int64_t)n * 3 + pl) * C + c) * H * W;
                    float v = 0.f;
                    for (int cy = 0; cy < 2; ++cy)
                        for (int cx = 0; cx < 2; ++cx) {
                            const int xx = x0 + cx, yy = y0 + cy;
                            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                            v += pc[(int64_t)yy * W + xx] * ((cx ? wx1 : wx0) * (cy ? wy1 : wy0));
                        }
                    f[c] += v;
                }
            }
            for (int c = 0; c < C; ++c) f[c] /= 3.f;
            float h[64], yv[33];
            for (int j = 0; j < 64; ++j) {
                float a = b1[j];
                for (int c = 0; c < C; ++c) a += w1[j * C + c] * f[c];
                h[j] = softplus_f(a);
            }
            for (int o = 0; o < 33; ++o) {
                float a = b2[o];
                for (int j = 0; j < 64; ++j) a += w2[o * 64 + j] * h[j];
                yv[o] = a;
            }
            const int64_t g = (int64_t)n * P + p;
            sigma[g] = yv[0];
            for (int j = 0; j < 32; ++j) rgb[g * 32 + j] = 1.f / (1.f + expf(-yv[1 + j])) * 1.002f - 0.001f;
        }
}

/* colors [R,S,C], densities [R,S], depths [R,S] sorted -> rgb [R,C], depth [R], weights [R,S-1] */
void ref_ray_march(const float* colors, const float* densities, const float* depths, int64_t R, int S, int C, int white_back,
                   float* rgb, float* depth, float* weights) {
    float dmin = INFINITY, dmax = -INFINITY;
    for (int64_t i = 0; i < R * S; ++i) { if (depths[i] < dmin) dmin = depths[i]; if (depths[i] > dmax) dmax = depths[i]; }
    for (int64_t r = 0; r < R; ++r) {
        const float* c = colors + r * S * C; const float* sg = densities + r * S; const float* t = depths + r * S;
        float T = 1.f, wsum = 0.f, dnum = 0.f;
        for (int ch = 0; ch < C; ++ch) rgb[r * C + ch] = 0.f;
        for (int k = 0; k < S - 1; ++k) {
            const float delta = t[k + 1] - t[k];
            const float smid = softplus_f((sg[k] + sg[k + 1]) / 2.f - 1.f);
            const float alpha = 1.f - expf(-smid * delta);
            const float w = alpha * T;
            T *= 1.f - alpha + 1e-10f;
            weights[r * (S - 1) + k] = w;
            wsum += w; dnum += w * ((t[k] + t[k + 1]) / 2.f);
            for (int ch = 0; ch < C; ++ch) rgb[r * C + ch] += w * ((c[k * C + ch] + c[(k + 1) * C + ch]) / 2.f);
        }
        float d = dnum / wsum;
        if (d != d) d = INFINITY;
        d = d < dmin ? dmin : (d > dmax ? dmax : d);
        depth[r] = d;
        for (int ch = 0; ch < C; ++ch) rgb[r * C + ch] = (rgb[r * C + ch] + (white_back ? 1.f - wsum : 0.f)) * 2.f - 1.f;
    }
}

/* depths [R,S], weights [R,S-1], u [R,Sf] -> fine [R,Sf] in draw order */
void ref_importance(const float* depths, const float* weights, const float* u, int64_t R, int S, int Sf, float* fine) {
    const int L = S - 1, NP = L - 2;
    float* a = (float*)malloc(sizeof(float) * (L + 2));
    float* cdf = (float*)malloc(sizeof(float) * (NP + 1));
    float* bins = (float*)malloc(sizeof(float) * L);
    for (int64_t r = 0; r < R; ++r) {
        const float* w = weights + r * L; const float* t = depths + r * S;
        for (int i = 0; i < L; ++i) {                     /* max_pool1d(2,1,pad 1) then avg_pool1d(2,1), + 0.01 */
            const float m0 = fmaxf(i > 0 ? w[i - 1] : -INFINITY, w[i]);
            const float m1 = fmaxf(w[i], i + 1 < L ? w[i + 1] : -INFINITY);
            a[i] = 0.5f * (m0 + m1) + 0.01f;
            bins[i] = 0.5f * (t[i] + t[i + 1]);
        }
        float tot = 0.f;
        for (int j = 0; j < NP; ++j) tot += a[j + 1] + 1e-5f;
        cdf[0] = 0.f;
        float run = 0.f;
        for (int j = 0; j < NP; ++j) { run += (a[j + 1] + 1e-5f) / tot; cdf[j + 1] = run; }
        for (int j = 0; j < Sf; ++j) {
            const float uu = u[r * Sf + j];
            int idx = 0;
            while (idx < NP + 1 && cdf[idx] <= uu) ++idx;            /* searchsorted(right=True) */
            const int lo = idx - 1 < 0 ? 0 : idx - 1, hi = idx > NP ? NP : idx;
            float den = cdf[hi] - cdf[lo];
            if (den < 1e-5f) den = 1.f;
            fine[r * Sf + j] = bins[lo] + (uu - cdf[lo]) / den * (bins[hi] - bins[lo]);
        }
    }
    free(a); free(cdf); free(bins);
}
