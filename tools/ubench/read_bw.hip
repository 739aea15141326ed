// Read-only HBM bandwidth of the access shapes the ray marcher uses: what "the roofline" is in practice for a cold 430 MB stream.
//   build: hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip ; run: ./read_bw
//   mode 0: grid-stride float4 stream, every wave reads 1 KB per instruction (64 lanes x 16 B), U loads in flight per lane
//   mode 1: the same with non-temporal loads
//   mode 2: ray-shaped: one wave per 24 KB "ray", 24 x 1 KB wave-loads issued back to back (the marcher's colour stream), NT
//   mode 3: mode 2, 8 rays per wave one after the other
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int U>
__global__ void __launch_bounds__(256) rd(const f32x4* __restrict__ p, size_t n4, float* out) {
    f32x4 acc = {0, 0, 0, 0};
    if (MODE <= 1) {
        size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
        const size_t stride = (size_t)gridDim.x * 256 * U;
        for (; i + (U - 1) * 256 < n4; i += stride) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = MODE ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    } else {
        const int lane = threadIdx.x & 63;
        const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
        const size_t rays = n4 / (24 * 64);
        for (size_t r = wave; r < rays; r += nw) {
            f32x4 v[24];
#pragma unroll
            for (int u = 0; u < 24; ++u) v[u] = __builtin_nontemporal_load(p + r * 24 * 64 + u * 64 + lane);
#pragma unroll
            for (int u = 0; u < 24; ++u) acc += v[u];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
int main() {
    const size_t bytes = 16384ull * 192 * 128 + 16384ull * 192 * 12;      // 430 MB
    const size_t n4 = bytes / 16;
    f32x4* p; float* out; char* flush;
    hipMalloc(&p, bytes); hipMalloc(&out, 4); hipMalloc(&flush, 512u << 20);
    hipMemset(p, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto launch) {
        std::vector<float> ts;
        for (int it = 0; it < 12; ++it) {
            hipMemsetAsync(flush, it, 512u << 20, 0);
            hipEventRecord(a, 0); launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-58s %7.1f us  %6.0f GB/s\n", name, ts[ts.size() / 2] * 1e3, bytes / ts[ts.size() / 2] / 1e6);
    };
    run("stream, 4 x 16 B per lane in flight, 2048 blocks", [&] { hipLaunchKernelGGL((rd<0, 4>), dim3(2048), dim3(256), 0, 0, p, n4, out); });
    run("stream, 8 x 16 B per lane in flight, 2048 blocks", [&] { hipLaunchKernelGGL((rd<0, 8>), dim3(2048), dim3(256), 0, 0, p, n4, out); });
    run("stream, 8 x 16 B, one pass (grid = n/2048)", [&] { hipLaunchKernelGGL((rd<0, 8>), dim3((unsigned)(n4 / 2048)), dim3(256), 0, 0, p, n4, out); });
    run("stream NT, 8 x 16 B per lane, 2048 blocks", [&] { hipLaunchKernelGGL((rd<1, 8>), dim3(2048), dim3(256), 0, 0, p, n4, out); });
    run("stream NT, 8 x 16 B, one pass", [&] { hipLaunchKernelGGL((rd<1, 8>), dim3((unsigned)(n4 / 2048)), dim3(256), 0, 0, p, n4, out); });
    run("ray-shaped NT, one wave per 24 KB ray (4096 blocks)", [&] { hipLaunchKernelGGL((rd<2, 1>), dim3(4096), dim3(256), 0, 0, p, n4, out); });
    run("ray-shaped NT, 8 rays per wave (512 blocks)", [&] { hipLaunchKernelGGL((rd<2, 1>), dim3(512), dim3(256), 0, 0, p, n4, out); });
    run("ray-shaped NT, 16 rays per wave... (256 blocks)", [&] { hipLaunchKernelGGL((rd<2, 1>), dim3(256), dim3(256), 0, 0, p, n4, out); });
    run("ray-shaped NT, 1024 blocks", [&] { hipLaunchKernelGGL((rd<2, 1>), dim3(1024), dim3(256), 0, 0, p, n4, out); });
    return 0;
}
