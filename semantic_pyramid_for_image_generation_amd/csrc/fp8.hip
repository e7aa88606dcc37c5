// fp8 (OCP e4m3, gfx950) helpers of the SP_F8 convolution path - BASELINE.json config 5: activations enter the fp8 chain of the
// frozen VGG-16 pyramid through sp_quantize_fp8, its filters are packed once by sp_pack_weight_fp8 (one scale per output
// channel), and the per-tensor activation scales follow the running maxima of the previous call (sp_fp8_update_scales:
// delayed scaling, no host round trip).  The convolution itself is conv_pp.hip's kernel instantiated for e4m3 operands.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, pk, true);
    return (unsigned)pk;
}

template <typename T> struct Load16;
template <> struct Load16<bf16> {
    static __device__ __forceinline__ void ld(const bf16* p, float (&o)[16]) {
        VecIO<bf16, 8>::ld(p, *reinterpret_cast<float(*)[8]>(&o[0]));
        VecIO<bf16, 8>::ld(p + 8, *reinterpret_cast<float(*)[8]>(&o[8]));
    }
};
template <> struct Load16<float> {
    static __device__ __forceinline__ void ld(const float* p, float (&o)[16]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) VecIO<float, 4>::ld(p + 4 * k, *reinterpret_cast<float(*)[4]>(&o[4 * k]));
    }
};

template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const T* __restrict__ x, uint4* __restrict__ q, long groups,
                                                           const float* __restrict__ inv_scale, float* amax) {
    const float s = inv_scale[0];
    float m = 0.f;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long)gridDim.x * 256) {
        float v[16];
        Load16<T>::ld(x + g * 16, v);
        unsigned w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(v[k * 4 + e]));
            w[k] = pack4_e4m3(v[k * 4] * s, v[k * 4 + 1] * s, v[k * 4 + 2] * s, v[k * 4 + 3] * s);
        }
        q[g] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (amax != nullptr) {
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(m));
    }
}

// one block per output channel: max |w[co]| -> scale, then [tap][cin_p] e4m3 (pad channels zero)
__global__ __launch_bounds__(256) void pack_weight_fp8_kernel(const float* __restrict__ w, int cin, int cin_p, uint8_t* __restrict__ out,
                                                              float* __restrict__ w_scale) {
    __shared__ float red[4];
    const int co = blockIdx.x;
    const float* src = w + (long)co * cin * 9;             // [cin][3][3]
    float m = 0.f;
    for (int i = threadIdx.x; i < cin * 9; i += 256) m = fmaxf(m, fabsf(src[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = m > 0.f ? m / 448.f : 1.f;
    const float inv = 1.f / scale;
    if (threadIdx.x == 0) w_scale[co] = scale;
    uint8_t* dst = out + (long)co * 9 * cin_p;
    for (int i = threadIdx.x; i < 9 * cin_p / 4; i += 256) {
        const int tap = (i * 4) / cin_p, c0 = (i * 4) - tap * cin_p;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c0 + e < cin ? src[(long)(c0 + e) * 9 + tap] * inv : 0.f;
        reinterpret_cast<unsigned*>(dst)[i] = pack4_e4m3(v[0], v[1], v[2], v[3]);
    }
}

__global__ void fp8_update_scales_kernel(float* amax, float* scale, float* inv_scale, int n, float margin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = amax[i];
    if (a > 0.f) {
        const float s = margin * a / 448.f;
        scale[i] = s;
        inv_scale[i] = 1.f / s;
    }
    amax[i] = 0.f;
}

}  // namespace

extern "C" int sp_quantize_fp8(const void* x, void* q, int64_t numel, const float* inv_scale, float* amax, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && q && inv_scale && numel > 0 && numel % 16 == 0, "sp_quantize_fp8: bad args (numel must be a multiple of 16)");
    SP_CHECK_ARG(dtype == SP_F32 || dtype == SP_BF16, "sp_quantize_fp8: source dtype must be SP_F32 or SP_BF16");
    const long groups = numel / 16;
    long blocks = (groups + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SP_BF16)
        hipLaunchKernelGGL(quantize_fp8_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const bf16*>(x), reinterpret_cast<uint4*>(q), groups, inv_scale, amax);
    else
        hipLaunchKernelGGL(quantize_fp8_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float*>(x), reinterpret_cast<uint4*>(q), groups, inv_scale, amax);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_pack_weight_fp8(const float* w, int32_t cout, int32_t cin, int32_t cin_p, void* out, float* w_scale, sp_stream_t stream) {
    SP_CHECK_ARG(w && out && w_scale && cout > 0 && cin > 0 && cin_p >= cin && cin_p % 16 == 0, "sp_pack_weight_fp8: bad args (cin_p: multiple of 16, >= cin)");
    hipLaunchKernelGGL(pack_weight_fp8_kernel, dim3((unsigned)cout), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, cin, cin_p,
                       reinterpret_cast<uint8_t*>(out), w_scale);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_fp8_update_scales(float* amax, float* scale, float* inv_scale, int32_t n, float margin, sp_stream_t stream) {
    SP_CHECK_ARG(amax && scale && inv_scale && n > 0 && margin >= 1.f, "sp_fp8_update_scales: bad args (margin >= 1)");
    hipLaunchKernelGGL(fp8_update_scales_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), amax, scale, inv_scale, n, margin);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
