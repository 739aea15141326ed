// micro-benchmark (round 6): v_mfma_f32_32x32x16_bf16 beside independent VALU work, against v_mfma_f32_32x32x2_f32 beside the same work.
// Question behind it: the decoder MLP kernels run fp32 MFMAs at ~50 % matrix-pipe busy with ~7.7 vector instructions per MFMA; on gfx950 the fp32 MFMA
// shares the FP32 lanes with the VALU (mfma_valu.hip: ~2.6 matrix cycles per vector instruction).  Does a bf16 MFMA (split-bf16 operands, 6 products)
// overlap with vector work, i.e. do the operand splits come for free in its shadow?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NV, bool BF>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a0 = lane * 0.01f, a1 = a0 + 1.f, b0 = 0.5f, b1 = 0.25f;
    bf16x8 A0, A1, B0, B1;
    for (int i = 0; i < 8; ++i) { A0[i] = (__bf16)(a0 + i); A1[i] = (__bf16)(a1 - i); B0[i] = (__bf16)(b0 * i); B1[i] = (__bf16)(b1 + i); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = lane + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (BF) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((m & 2) ? A1 : A0, (m & 1) ? B1 : B0, acc[m], 0, 0, 0);
                else acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32((m & 2) ? a1 : a0, (m & 1) ? b1 : b0, acc[m], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NV; ++q) v[q % 16] = __builtin_fmaf(v[q % 16], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool BF>
int run(float* d, int blocks) {
    const int iters = 1000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<NV, BF>), dim3(blocks), dim3(256), 0, 0, d, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<NV, BF>), dim3(blocks), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // per SIMD: (blocks / 256) waves x iters x 32 MFMAs each
    const double mfma_per_simd = (double)(blocks / 256) * iters * 32.0;
    const double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
    printf("%s  VALU per MFMA %2d, %d wave(s)/SIMD: %.3f ms -> %.1f SIMD-cycles per MFMA (+ its %d vector instructions)\n", BF ? "bf16 32x32x16" : "f32  32x32x2 ", NV,
           blocks / 256, ms, cyc, NV);
    return 0;
}

int main() {
    float* d; CK(hipMalloc(&d, 4096 * 256 * 4));
    for (int blocks : {256, 512}) {
        run<0, false>(d, blocks); run<4, false>(d, blocks); run<8, false>(d, blocks); run<16, false>(d, blocks);
        run<0, true>(d, blocks); run<2, true>(d, blocks); run<4, true>(d, blocks); run<8, true>(d, blocks); run<12, true>(d, blocks); run<16, true>(d, blocks);
    }
    return 0;
}
