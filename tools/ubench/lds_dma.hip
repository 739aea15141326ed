#include <hip/hip_runtime.h>
__global__ void k(const float* __restrict__ g, float* out, int n) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, n * 4, 0x00020000);
    unsigned voff = (lane * 7 % 64) * 4;
    if (lane == 5) voff = 0x80000000u;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + wave * 256);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, %3 offen lds" :: "s"(dst), "v"(voff), "s"(rs), "s"(0) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}
int main() {
    float *g, *o; hipMalloc(&g, 4096); hipMalloc(&o, 1024);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i; hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, g, o, 1024);
    float r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int t = 0; t < 256; ++t) { int l = t & 63; float e = l == 5 ? 0.f : (float)(l * 7 % 64); if (r[t] != e) { if (bad < 5) printf("t=%d got %f exp %f\n", t, r[t], e); ++bad; } }
    printf("bad=%d\n", bad); return 0;
}
