#!/usr/bin/env python3
"""Ablation probe of hconv_kernel (run on the GPU box): builds variants of spi_amd/csrc/hconv.hip with parts of the kernel removed by text
substitution (stores / MFMAs / in-loop loads) plus a tiny driver, and prints the time of each on the SR b512.conv1 shape.  Says where the
time goes; results are NOT valid convolutions.  Usage: python tools/ubench/hconv_probe.py [variant ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = open(os.path.join(ROOT, 'spi_amd', 'csrc', 'hconv.hip')).read()
DRIVER = r'''
#include <cstdio>
#include <vector>
void spi_set_error(const char*, ...) {}
#ifdef HW_DRIVER
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1, C = argc > 2 ? atoi(argv[2]) : 128, H = argc > 3 ? atoi(argv[3]) : 512;
    WinoParams Wp{}; Wp.N = N; Wp.nw = N; Wp.Mo = C; Wp.Ci = C; Wp.H = H; Wp.W = H; Wp.in_bs = (int64_t)C * H * H; Wp.out_bs = Wp.in_bs;
    Wp.wbs = (int64_t)C * C * 9; Wp.wsm = C * 9; Wp.wsc = 1; for (int t = 0; t < 9; ++t) Wp.widx[t] = t * C;
    _Float16 *x, *dy; float* dw;
    hipMalloc(&x, Wp.in_bs * N * 2); hipMalloc(&dy, Wp.out_bs * N * 2); hipMalloc(&dw, (size_t)N * C * C * 9 * 4);
    hipMemset(x, 0, Wp.in_bs * N * 2); hipMemset(dy, 0, Wp.out_bs * N * 2); hipMemset(dw, 0, (size_t)N * C * C * 9 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    void* ws = nullptr; const int64_t wsb = spi_hwgrad_workspace_bytes(Wp); if (argc > 4) hipMalloc(&ws, wsb);
    for (int i = 0; i < 3; ++i) spi_hwgrad_launch(Wp, x, dy, dw, ws, wsb, 0);
    hipEventRecord(e0);
    const int R = 20;
    for (int i = 0; i < R; ++i) spi_hwgrad_launch(Wp, x, dy, dw, ws, wsb, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * N * C * C * 9 * H * H;
    printf("%8.1f us  %7.1f TF/s  (%s)\n", ms * 1e3 / R, fl / (ms / R) / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
#else
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1, C = argc > 2 ? atoi(argv[2]) : 128, H = argc > 3 ? atoi(argv[3]) : 512;
    HConvParams P; P.N = N; P.nw = 1; P.Mo = C; P.Ci = C; P.H = H; P.W = H; P.tx = (H + HC_TX - 1) / HC_TX; P.ty = (H + HC_TY - 1) / HC_TY;
    P.in_bs = (int64_t)C * H * H; P.out_bs = P.in_bs; P.seg_flags = nullptr; P.out_flags = nullptr; P.nseg = 0;
    _Float16 *in, *out; u32x4_t* img;
    hipMalloc(&in, P.in_bs * N * 2); hipMalloc(&out, P.out_bs * N * 2); hipMalloc(&img, (size_t)C * C * 9 * 2);
    hipMemset(in, 0, P.in_bs * N * 2); hipMemset(img, 0, (size_t)C * C * 9 * 2);
    Epilogue ep{nullptr, nullptr, nullptr, 0, 0.f, 1.f, -1.f};
    dim3 grid(P.tx * P.ty, C / HC_BM, N);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(hconv_kernel, grid, dim3(HC_NT), 0, 0, P, in, img, out, ep);
    hipEventRecord(e0);
    const int R = 20;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(hconv_kernel, grid, dim3(HC_NT), 0, 0, P, in, img, out, ep);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * N * C * C * 9 * H * H;
    printf("%8.1f us  %7.1f TF/s  (%s)\n", ms * 1e3 / R, fl / (ms / R) / 1e9, hipGetErrorString(hipGetLastError()));
#ifdef HC_STAMPS
    std::vector<long long> st(512 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
    long long t0 = st[0]; for (int b = 0; b < 512; ++b) if (st[b * 16]) t0 = std::min(t0, st[b * 16]);
    for (int b : {0, 1, 7, 8, 100, 255, 256, 257, 300, 511}) {
        printf("blk %3d start %8lld |", b, st[b * 16] - t0);
        for (int i = 1; i < 11; ++i) printf(" %6lld", st[b * 16 + i] - st[b * 16 + i - 1]);
        printf(" | epi: regs->lds %lld barrier %lld lds->stores %lld drain %lld\n", st[b * 16 + 12] - st[b * 16 + 10], st[b * 16 + 13] - st[b * 16 + 12], st[b * 16 + 14] - st[b * 16 + 13], st[b * 16 + 11] - st[b * 16 + 14]);
    }
#endif
    return 0;
}
#endif
'''
STORE = 'if (wide) *reinterpret_cast<u32x4_t*>(dst) = u32x4_t{lo[0], lo[1], hi[0], hi[1]};'
MFMA = 'acc[r][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xq[r], wq[i], acc[r][i], 0, 0, 0);'
STAMP_SUBS = [
    ('__global__ void __launch_bounds__(HC_NT, 2) hconv_kernel(', '__device__ long long g_stamps[512 * 16];\n__global__ void __launch_bounds__(HC_NT, 2) hconv_kernel('),
    ('    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;', '    long long TS[12]; TS[0] = clock64();\n    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;'),
    ('    issue(0);\n    commit(0);\n    if (nchunk > 1) issue(1);\n    __syncthreads();', '    TS[1] = clock64(); issue(0);\n    commit(0);\n    if (nchunk > 1) issue(1);\n    __syncthreads(); TS[2] = clock64();'),
    ('        }\n        __syncthreads();', '        }\n        __syncthreads(); if (c < 8) TS[3 + c] = clock64();'),
    ('// ---- host side (called from conv.hip', '// host'),
]
STAMP_TAIL = ('}\n#undef HC_FENCE', '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TS[11] = clock64();\n    if (tid == 0) { for (int i = 0; i < 12; ++i) g_stamps[blockIdx.x * 16 + i] = TS[i]; g_stamps[blockIdx.x * 16 + 12] = __builtin_amdgcn_s_getreg(6164); }\n}\n#undef HC_FENCE')
EPI_SUBS = [('    __syncthreads();\n    {\n        typedef unsigned u32x2_t', '    TS[12] = clock64(); __syncthreads(); TS[13] = clock64();\n    {\n        typedef unsigned u32x2_t'),
            ('}\n#undef HC_FENCE', '    TS[14] = clock64();\n}\n#undef HC_FENCE')]
ATOM = 'atomicAdd(dt + (int64_t)co * P.wsm, acc[ky][kx][q]);'
WMFMA = [(f'acc[ky][{i}] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b{i}, acc[ky][{i}], 0, 0, 0);', f'acc[ky][{i}][0] += (float)a[0] * (float)b{i}[0];') for i in range(3)]
HW_VARIANTS = {
    'w_base': [],
    'w_noatomic': [(ATOM, 'if (acc[ky][kx][q] == 12345.678f) ' + ATOM)],
    'w_wgscope': [(ATOM, '__hip_atomic_fetch_add(dt + (int64_t)co * P.wsm, acc[ky][kx][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);')],
    'w_store': [(ATOM, 'dt[(int64_t)co * P.wsm] = acc[ky][kx][q];')],
    'w_nomfma': WMFMA,
    'w_noload': [('if (t + 2 < t_end) issue(t + 2);', 'if (t + 2 < t_end && P.N == 12345) issue(t + 2);')],
    'w_nocommit': [('if (t + 1 < t_end) commit(buf ^ 1);', 'if (t + 1 < t_end && P.N == 12345) commit(buf ^ 1);')],
}
VARIANTS = {
    'stamps': [(a, b.replace('long long TS[12]', 'long long TS[16]')) for a, b in STAMP_SUBS[:4]] + EPI_SUBS + [(STAMP_TAIL[0], STAMP_TAIL[1].replace('i < 12; ++i) g_stamps[blockIdx.x * 16 + i]', 'i < 15; ++i) g_stamps[blockIdx.x * 16 + i]').replace('g_stamps[blockIdx.x * 16 + 12] = __builtin_amdgcn_s_getreg(6164);', ''))],
    'base': [],
    'nostore': [(STORE, STORE.replace('if (wide)', 'if (wide && lo[0] == 0x12345678u)'))],
    'nomfma': [(MFMA, 'acc[r][i][0] += (float)wq[i][0] * (float)xq[r][0];')],
    'noloopload': [('if (c + 2 < nchunk) issue(c + 2);', 'if (c + 2 < nchunk && P.N == 12345) issue(c + 2);')],
    'nocommit': [('if (c + 1 < nchunk) commit(buf ^ 1);', 'if (c + 1 < nchunk && P.N == 12345) commit(buf ^ 1);')],
    'noload_nocommit': [('if (c + 2 < nchunk) issue(c + 2);', 'if (c + 2 < nchunk && P.N == 12345) issue(c + 2);'),
                        ('if (c + 1 < nchunk) commit(buf ^ 1);', 'if (c + 1 < nchunk && P.N == 12345) commit(buf ^ 1);')],
}


def main():
    VARIANTS.update(HW_VARIANTS)
    names = sys.argv[1:] or [v for v in VARIANTS if not v.startswith('w_')]
    os.makedirs('/tmp/hcp', exist_ok=True)
    for nm in names:
        s = SRC
        for a, b in VARIANTS[nm]:
            assert a in s, (nm, a)
            s = s.replace(a, b)
        path = f'/tmp/hcp/{nm}.hip'
        open(path, 'w').write(s + DRIVER)
        exe = f'/tmp/hcp/{nm}'
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-mllvm', '-amdgpu-mfma-vgpr-form=1', '-Wno-pass-failed',
                               '-Wno-unused-result', *(['-DHC_STAMPS'] if nm == 'stamps' else []), *(['-DHW_DRIVER'] if nm.startswith('w_') else []), '-Wno-unused-value', '-Wno-comment', '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'spi_amd', 'csrc'), path, '-o', exe])
        for args in ((['1', '128', '512'], ['1', '128', '512', 'parts'], ['1', '256', '256', 'parts']) if nm.startswith('w_') else (['1', '128', '512'], ['1', '256', '256'])):
            out = subprocess.run([exe] + args, capture_output=True, text=True).stdout.strip()
            print(f'{nm:18s} {"x".join(args):18s} {out}', flush=True)
            if nm == 'stamps':
                break


if __name__ == '__main__':
    main()
