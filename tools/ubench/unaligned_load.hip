// micro-benchmark: 16-byte global loads at 2-byte (odd-halfword) alignment -- do they work on gfx950 (unaligned access mode) and what do they cost?
// A wave reads 64 x 16 B = 1 KB per instruction from `base + shift` halves; checks the values and times a streaming read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct __attribute__((packed, aligned(2))) U8H { unsigned short h[8]; };

__global__ void __launch_bounds__(256) rd(const unsigned short* __restrict__ x, int64_t n8, int shift, unsigned long long* __restrict__ out, int check) {
    unsigned long long acc = 0ull, bad = 0ull;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n8; g += (int64_t)gridDim.x * 256) {
        const U8H v = *reinterpret_cast<const U8H*>(x + g * 8 + shift);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc += v.h[j];
            if (check && v.h[j] != (unsigned short)((g * 8 + shift + j) * 2654435761u >> 16)) ++bad;
        }
    }
    atomicAdd(out, acc);
    if (check) atomicAdd(out + 1, bad);
}

int main() {
    const int64_t n = (int64_t)1 << 28;            // 256 M halves = 512 MB
    unsigned short* d; unsigned long long* o;
    CK(hipMalloc(&d, (n + 64) * 2)); CK(hipMalloc(&o, 16));
    unsigned short* h = (unsigned short*)malloc((n + 64) * 2);
    for (int64_t i = 0; i < n + 64; ++i) h[i] = (unsigned short)((uint32_t)i * 2654435761u >> 16);
    CK(hipMemcpy(d, h, (n + 64) * 2, hipMemcpyHostToDevice));
    for (int shift : {0, 1, 2, 3, 4, 7}) {
        CK(hipMemset(o, 0, 16));
        hipLaunchKernelGGL(rd, dim3(4096), dim3(256), 0, 0, d, n / 8, shift, o, 1);
        CK(hipDeviceSynchronize());
        unsigned long long r[2]; CK(hipMemcpy(r, o, 16, hipMemcpyDeviceToHost));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(rd, dim3(4096), dim3(256), 0, 0, d, n / 8, shift, o, 0);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("shift %d halves: wrong values %llu, %.2f TB/s\n", shift, r[1], 5.0 * n * 2 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
