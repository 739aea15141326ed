// micro-benchmark: how many independent VALU instructions fit in the shadow of one v_mfma_f32_32x32x2_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NV>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a0 = lane * 0.01f, a1 = a0 + 1.f, b0 = 0.5f, b1 = 0.25f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = lane + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32((m & 2) ? a1 : a0, (m & 1) ? b1 : b0, acc[m], 0, 0, 0);
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

template <int NV>
int run(float* d, int blocks) {
    const int iters = 1000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, d, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double flop = (double)blocks * 4 * iters * 32.0 * 4096.0;
    printf("VALU per MFMA %2d, blocks %4d (%d waves/SIMD): %.3f ms -> %.1f TFLOP/s (MFMA only)\n", NV, blocks, blocks / 256, ms, flop / ms / 1e9);
    return 0;
}

int main() {
    float* d; CK(hipMalloc(&d, 4096 * 256 * 4));
    for (int blocks : {256, 512}) {
        run<0>(d, blocks); run<2>(d, blocks); run<4>(d, blocks); run<6>(d, blocks); run<8>(d, blocks); run<10>(d, blocks); run<12>(d, blocks); run<16>(d, blocks);
    }
    return 0;
}
