// How fast can 256-thread blocks stream a [4096][25088] bf16 matrix (205 MB), as a function of the access pattern of a wave's 16-byte
// loads?  A: the MFMA A-fragment pattern on the row-major matrix (16 rows x 64 B per instruction, 8 instructions = 512 B per row);
// B: fragment-major storage (1 KB contiguous per instruction); C: row-major, one row per wave instruction (1 KB contiguous per row).
// hipcc --offload-arch=gfx950 -O3 scratch/wstream_bench.hip -o scratch/wstream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int N = 4096, KP = 25088, KS = 1024;
template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ w, uint4* __restrict__ out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 128 + wave * 32, k0 = blockIdx.y * KS;
    const int klen = min(KS, KP - k0);
    const int frow = lane & 15, g = lane >> 4;
    uint4 acc = make_uint4(0, 0, 0, 0);
    constexpr int U = 8;
    for (int q = 0; q < KS / 32 / U; ++q) {
        uint4 a[2][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = q * U + u;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                long idx;       // in uint4 (8 bf16) units
                if (MODE == 0) idx = ((long)(n0 + i * 16 + frow) * KP + k0 + kk * 32 + g * 8) / 8;
                else if (MODE == 1) idx = (((long)((n0 >> 4) + i) * (KP / 32) + (k0 / 32 + kk)) * 64 + lane);          // fragment-major: [n tile][k step][lane]
                else idx = ((long)(n0 + i * 16 + (kk & 15)) * KP + k0 + (kk >> 4) * 512 + lane * 8) / 8;                  // a row's 1 KB per instruction
                const bool ok = kk * 32 < klen;
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok) { const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + idx)); v = make_uint4(t.x, t.y, t.z, t.w); }
                a[i][u] = v;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) { acc.x ^= a[i][u].x; acc.y += a[i][u].y; acc.z ^= a[i][u].z; acc.w += a[i][u].w; }
    }
    if (acc.x == 0x12345678u) out[blockIdx.x] = acc;
}
int main() {
    const size_t bytes = (size_t)N * KP * 2;
    uint4 *w, *out, *flush;
    CK(hipMalloc(&w, bytes)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&flush, 512 << 20));
    CK(hipMemset(w, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 grid(N / 128, (KP + KS - 1) / KS);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(flush, rep, 512 << 20));
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, grid, dim3(256), 0, 0, w, out);
            else if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, grid, dim3(256), 0, 0, w, out);
            else hipLaunchKernelGGL(stream_kernel<2>, grid, dim3(256), 0, 0, w, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("mode %d: %.1f us, %.2f TB/s\n", mode, best * 1e3, bytes / best / 1e9);
    }
    return 0;
}
