// micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate (register-only) and with the LDS fragment reads of conv.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    __shared__ float As[2][16 * 132], Bs[2][16 * 132];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 16 * 132; i += 256) { (&As[0][0])[i] = 0.001f * (i % 97); (&Bs[0][0])[i] = 0.002f * (i % 89); }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float a0 = lane * 0.01f, a1 = a0 + 1.f, b0 = 0.5f, b1 = 0.25f;
    const int fr = lane & 31, fk = lane >> 5;
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (MODE == 1) {
                a0 = As[buf][(2 * kk + fk) * 132 + fr]; a1 = As[buf][(2 * kk + fk) * 132 + 32 + fr];
                b0 = Bs[buf][(2 * kk + fk) * 132 + fr]; b1 = Bs[buf][(2 * kk + fk) * 132 + 32 + fr];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (MODE == 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
int run(const char* name, float* d, int blocks) {
    const int iters = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double flop = (double)blocks * 4 /*waves*/ * iters * 32.0 * (2.0 * 32 * 32 * 2);
    printf("%-44s blocks %4d: %.3f ms -> %.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
    return 0;
}

int main() {
    float* d; CK(hipMalloc(&d, 4096 * 256 * 4));
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0>("registers only", d, blocks);
        run<1>("+ LDS fragment reads (conv.hip pattern)", d, blocks);
        run<2>("registers + __syncthreads per 32 MFMAs", d, blocks);
    }
    return 0;
}
