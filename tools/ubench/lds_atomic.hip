// micro-benchmark: LDS float atomic (ds_add_f32) vs plain read-modify-write vs integer atomic throughput
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, int stride) {
    __shared__ float win[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) win[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx = (wave * 64 + lane * stride) & 8191;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) atomicAdd(&win[idx], v);
        else if (MODE == 1) win[idx] += v;
        else if (MODE == 2) atomicAdd(reinterpret_cast<int*>(win) + idx, (int)v);
        else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long*>(win) + (idx >> 1), (unsigned long long)v);
        else if (MODE == 4) atomicAdd(reinterpret_cast<double*>(win) + (idx >> 1), (double)v);
        idx = (idx + 97 * 32) & 8191;           // move to another texel row each iteration
    }
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < 8192; i += 256) s += win[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, float* d, int stride) {
    const int iters = 4096, blocks = 512;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, stride);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, stride);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // 512 blocks over 256 CUs -> 2 blocks per CU; per CU: 2 blocks * 4 waves * iters wave-instructions
    double cyc_per_instr = ms * 1e-3 * 2.4e9 / (2.0 * 4 * iters);
    printf("%-28s stride %2d: %.3f ms  -> %.1f CU-cycles per wave-instruction (8 waves/CU sharing the LDS)\n", name, stride, ms, cyc_per_instr);
    return 0;
}

int main() {
    float* d; CK(hipMalloc(&d, 512 * 256 * 4));
    for (int stride : {1, 0, 2, 32}) {
        run<0>("ds_add_f32 (atomicAdd float)", d, stride);
        run<1>("plain RMW", d, stride);
        run<2>("ds_add_u32 (atomicAdd int)", d, stride);
        run<3>("ds_add_u64", d, stride);
        run<4>("ds_add_f64 (atomicAdd double)", d, stride);
    }
    return 0;
}
