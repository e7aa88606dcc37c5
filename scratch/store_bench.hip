// Micro-benchmark: what does a wave pay to ISSUE its epilogue stores, as a function of the lane -> address pattern?
// 256 blocks x 512 threads (one block per CU, like the persistent conv kernels), every wave stores 8 x 1 KB per item, items are
// separated by ~20 K cycles of MFMA work.  Patterns: 0 = 16 B per lane at a 256-byte stride (today's NHWC epilogue, Cout = 128),
// 1 = 64-byte runs, 2 = 128-byte runs (one full line per 8 lanes), 3 = 1 KB contiguous, 4 = 512-byte stride, 5 = as 0 but only
// waves 0-3 store (the ping-pong kernel: one half at a time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k(char* out, long long* stamps, int pattern, int items, int work) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4f acc = {0, 0, 0, 0};
    v8s a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    long long t_store = 0;
    for (int it = 0; it < items; ++it) {
        for (int i = 0; i < work; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        __builtin_amdgcn_s_barrier();
        // the block's tile: 128 channels (256 B per pixel) x 256 px = 64 KB; wave w owns 64 channels x 64 px = 8 KB = 8 stores
        char* tile = out + ((long)(it * gridDim.x + blockIdx.x)) * 65536;
        long long t0 = __builtin_readcyclecounter();
        if (pattern < 5 || wave < 4) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                long off;
                const int px_base = (wave >> 1) * 64, co_base = (wave & 1) * 128;      // bytes within the pixel's 256 B
                if (pattern == 0 || pattern == 5) off = (long)(px_base + lane) * 256 + co_base + s * 16;
                else if (pattern == 1) off = (long)(px_base + (s & 3) * 16 + (lane >> 2)) * 256 + co_base + (s >> 2) * 64 + (lane & 3) * 16;
                else if (pattern == 2) off = (long)(px_base + s * 8 + (lane >> 3)) * 256 + co_base + (lane & 7) * 16;
                else if (pattern == 3) off = (long)wave * 8192 + s * 1024 + lane * 16;
                else if (pattern == 4) off = (long)(px_base + lane) * 512 % 65536 + co_base + s * 16;
                else if (pattern == 6) off = (long)(px_base + s * 8 + (lane >> 3)) * 256 + co_base + (lane & 7) * 16;          // full lines, half the waves
                else if (pattern == 7) off = (long)(px_base + (s & 3) * 16 + (lane & 15)) * 256 + co_base + (s >> 2) * 64 + (lane >> 4) * 16;   // the real epilogue: lane % 16 = pixel
                else if (pattern == 8) off = (long)(px_base + (s & 3) * 16 + (lane >> 2)) * 256 + co_base + (s >> 2) * 64 + (lane & 3) * 16;    // 64-byte runs, adjacent lanes
                else off = (long)wave * 8192 + s * 1024 + lane * 16;                                                             // 9: contiguous KB, half the waves
                v4f v = acc; v.x += s;
                *(v4f*)(tile + off) = v;
            }
        }
        long long t1 = __builtin_readcyclecounter();
        t_store += t1 - t0;
    }
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = t_store;
    if (acc.x == 12345.f) out[0] = 1;
}

int main(int argc, char** argv) {
    const int items = 40; const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    char* out; long long* st;
    hipMalloc(&out, (size_t)items * 256 * 65536 + 65536);
    hipMalloc(&st, 256 * 8 * sizeof(long long)); printf("blocks %d\n", blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int work : {0, 1200}) for (int p = 0; p < 10; ++p) {
        float best = 1e9; double cyc = 0;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, st, p, items, work);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            std::vector<long long> h(blocks * 8);
            hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
            double s = 0; int n = 0;
            for (int i = 0; i < blocks * 8; ++i) if (p < 5 || (i & 7) < 4) { s += h[i]; ++n; }
            cyc = s / n / items / 8;
        }
        printf("work %5d pattern %d: %8.3f ms per launch, %7.1f s_memtime ticks per store instruction\n", work, p, best, cyc);
    }
    return 0;
}
