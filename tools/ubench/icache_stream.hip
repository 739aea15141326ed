// Does a wave that walks through a LONG straight-line body (no inner loop to sit in the instruction buffer) issue at full rate?
// The ray-march backward executes ~16 KB of unrolled code per ray; this measures VALU issue for bodies of 0.5 ... 32 KB at 1-4 waves per SIMD.
//   build: hipcc --offload-arch=gfx950 -O3 -o icache_stream icache_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ void __launch_bounds__(256) body(float* out, int reps, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < N; ++i) {                      // 4 independent FMAs per step: N*4 VALU instructions, 8 bytes each (VOP3)
            asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
        }
    }
    if (x0 + x1 + x2 + x3 == 1234.5f) out[0] = x0;
}
template <int N>
void run(float* out, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;               // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    const int reps = (1 << 22) / (N * 4);                   // ~4 M instructions per wave
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(body<N>, dim3(blocks), dim3(256), 0, 0, out, reps, 1.0001f, 0.5f);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(body<N>, dim3(blocks), dim3(256), 0, 0, out, reps, 1.0001f, 0.5f);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = (double)reps * N * 4;
    printf("body %6d B  waves/SIMD %d : %8.1f us   %.2f ns per instruction per wave   %.2f ns per instruction per SIMD\n", N * 32, waves_per_simd, ms * 1e3,
           ms * 1e6 / instr, ms * 1e6 / instr / waves_per_simd);
}
int main() {
    float* out; (void)hipMalloc(&out, 4);
    for (int w = 1; w <= 4; ++w) { run<16>(out, w); run<128>(out, w); run<512>(out, w); run<1024>(out, w); run<2048>(out, w); }
    return 0;
}
