// Sustained MFMA rate under the chip's power budget: v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16, random operands,
// two waves per SIMD on every CU, 64 accumulator registers per wave (the ping-pong kernel's register tile).
//   hipcc -O3 --offload-arch=gfx950 -o scratch/mfma_rate scratch/mfma_rate.hip && scratch/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(512) void rate_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    uint4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 0xffff]; b[i] = src[(t * 8 + 4 + i) & 0xffff]; }
    float s = 0.f;
    if constexpr (KIND == 0) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[i >> 2]), acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(i + r) & 3]), __builtin_bit_cast(bf16x8, b[(i * 3 + r) & 3]), acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    }
    if (s == 12345.678f) out[t] = s;
}

int main() {
    std::vector<unsigned short> h(65536 * 8);
    srand(1);
    for (auto& v : h) {                                  // bf16 normals in [-2, 2): sign, exponent 124..127, random mantissa
        unsigned e = 124 + rand() % 4, m = rand() & 0x7f, sg = rand() & 1;
        v = (unsigned short)((sg << 15) | (e << 7) | m);
    }
    uint4* src; float* out;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;                              // 48 x 16384 (or 24 x 32768) FLOP per wave and iteration
    for (int round = 0; round < 4; ++round)
        for (int kind = 0; kind < 2; ++kind) {
            hipEventRecord(e0);
            for (int rep = 0; rep < 4; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(256), dim3(512), 0, 0, src, out, iters);
                else hipLaunchKernelGGL(rate_kernel<1>, dim3(256), dim3(512), 0, 0, src, out, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 4.0 * 256 * 8 * (double)iters * 48 * 16384;
            printf("round %d  %s  %8.2f ms  %7.1f TFLOP/s\n", round, kind == 0 ? "16x16x32" : "32x32x16", ms, flop / ms / 1e9);
        }
    return 0;
}
