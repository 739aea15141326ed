// micro-benchmark: one wave per SIMD, 16 independent 32x32 accumulators in AGPRs (the Winograd kernel's shape):
// MFMA rate with register operands / with 16-byte LDS fragment reads / with a block barrier per 64 MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = (i % 17) * 0.01f;
    __syncthreads();
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float* Ub = lds + (((lane >> 5) * 64 + (wave >> 1) * 32 + (lane & 31)) << 2);
    const float* Vb = lds + 16384 + (((lane >> 5) * 64 + (wave & 1) * 32 + (lane & 31)) << 2);
    float4 a0 = make_float4(lane * 0.01f, 1.f, 2.f, 3.f), b0 = make_float4(0.5f, 0.25f, 0.125f, 1.f);
    for (int it = 0; it < iters; ++it) {
        const int buf = (it & 1) * 8192;
        float4 af[2][4], bf[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[0][i] = MODE >= 1 ? *reinterpret_cast<const float4*>(Ub + buf + i * 512) : a0;
            bf[0][i] = MODE >= 1 ? *reinterpret_cast<const float4*>(Vb + buf + i * 512) : b0;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = j * 4 + i;
            acc[g * 4 + i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][j], bf[g & 1][i][j], acc[g * 4 + i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < 4 && t < 8) {
                if (t < 4) af[(g + 1) & 1][t] = MODE >= 1 ? *reinterpret_cast<const float4*>(Ub + buf + ((g + 1) * 4 + t) * 512) : a0;
                else bf[(g + 1) & 1][t - 4] = MODE >= 1 ? *reinterpret_cast<const float4*>(Vb + buf + ((g + 1) * 4 + t - 4) * 512) : b0;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
int run(float* d, int blocks, int iters = 2000) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 131072, 0, d, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 131072, 0, d, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double flop = (double)blocks * 4 * iters * 64.0 * 4096.0;
    printf("mode %d (0 regs, 1 + LDS fragments, 2 + barrier), %d blocks x %d slabs: %.3f ms -> %.1f TFLOP/s\n", MODE, blocks, iters, ms, flop / ms / 1e9);
    return 0;
}

int main() {
    float* d; CK(hipMalloc(&d, 1024 * 256 * 4));
    for (int blocks : {256, 1024}) { run<0>(d, blocks); run<1>(d, blocks); run<2>(d, blocks); }
    // short blocks (one Winograd tile = 16 / 32 slabs): what the workgroup turnover costs with one block per CU
    run<2>(d, 256, 128); run<2>(d, 1024, 32); run<2>(d, 2048, 16); run<2>(d, 256, 64); run<2>(d, 256, 32);
    return 0;
}
