// Multi-tensor Adam: every parameter of a network in ONE launch (torch.optim.Adam at main.py:64-65, .step() at
// model_wrapper.py:162,190 - the reference runs torch's per-tensor / foreach kernels: hundreds of launches per step).
// The host splits the tensors into chunks of <= 65536 elements; one block per chunk, so small tensors (biases,
// u / v vectors) cost one short block each and the big weight matrices spread over many.  Update rule and
// operation order follow torch/optim/adam.py (_single_tensor_adam, amsgrad = False, maximize = False):
//   g'  = g + wd * p
//   m  += (g' - m) * (1 - beta1)                      (Tensor.lerp_)
//   v   = v * beta2 + (1 - beta2) * g' * g'           (mul_ + addcmul_)
//   p  -= step_size * m / (sqrt(v) / sqrt(bc2) + eps), step_size = lr / bc1  (per-tensor scalars: the step count is
//                                                      per parameter, so they travel in the chunk table)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adam_multi_kernel(const sp_adam_chunk* __restrict__ chunks, float omb1, float beta2, float omb2,
                                                         float eps, float wd, const float* __restrict__ skip) {
    if (skip != nullptr && *skip != 0.f) return;          // sp_adam_multi_guarded: non-finite gradients were found - no update at all
    const sp_adam_chunk c = chunks[blockIdx.x];
    const bool vec = (((uintptr_t)c.p | (uintptr_t)c.g | (uintptr_t)c.m | (uintptr_t)c.v) & 15) == 0;
    const int n4 = vec ? c.n >> 2 : 0;
    for (int i = threadIdx.x; i < n4; i += 256) {
        float4 p = reinterpret_cast<float4*>(c.p)[i];
        const float4 g4 = reinterpret_cast<const float4*>(c.g)[i];
        float4 m = reinterpret_cast<float4*>(c.m)[i], v = reinterpret_cast<float4*>(c.v)[i];
        float pp[4] = {p.x, p.y, p.z, p.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w}, mm[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float g = wd != 0.f ? gg[r] + wd * pp[r] : gg[r];
            mm[r] = mm[r] + (g - mm[r]) * omb1;
            vv[r] = vv[r] * beta2 + omb2 * g * g;
            const float denom = sqrtf(vv[r]) * c.inv_sqrt_bc2 + eps;
            pp[r] = pp[r] - c.step_size * (mm[r] / denom);
        }
        reinterpret_cast<float4*>(c.p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        reinterpret_cast<float4*>(c.m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(c.v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    for (int i = n4 * 4 + threadIdx.x; i < c.n; i += 256) {
        const float g = wd != 0.f ? c.g[i] + wd * c.p[i] : c.g[i];
        const float m = c.m[i] + (g - c.m[i]) * omb1;
        const float v = c.v[i] * beta2 + omb2 * g * g;
        const float denom = sqrtf(v) * c.inv_sqrt_bc2 + eps;
        c.p[i] = c.p[i] - c.step_size * (m / denom);
        c.m[i] = m;
        c.v[i] = v;
    }
}

__global__ __launch_bounds__(256) void scale_f32_kernel(float* __restrict__ x, long n4, long n, float f, const float* __restrict__ fdev) {
    if (fdev != nullptr) f = *fdev;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<float4*>(x)[i];
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        reinterpret_cast<float4*>(x)[i] = v;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] *= f;
}

// any non-finite value in x -> *found = 1 (every thread that sees one stores the same value: no atomics, no reset here)
__global__ __launch_bounds__(256) void check_finite_kernel(const float* __restrict__ x, long n4, long n, float* __restrict__ found) {
    const long stride = (long)gridDim.x * 256;
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        // x - x is 0 for every finite x and NaN for +-inf / NaN
        const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
        bad |= !(t == 0.f);
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) bad |= !((x[i] - x[i]) == 0.f);
    if (bad) *found = 1.f;
}

// state: [0] scale, [1] 1 / scale, [2] clean optimizer steps since the scale last changed, [3] non-finite gradients found (consumed here),
// [4] optimizer steps skipped so far - torch.cuda.amp.GradScaler's policy, entirely on the device
__global__ void loss_scale_update_kernel(float* __restrict__ st, float growth, float backoff, float interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float scale = st[0], clean = st[2];
    if (st[3] != 0.f) {
        scale = fmaxf(scale * backoff, 1.f);
        clean = 0.f;
        st[4] += 1.f;
    } else {
        clean += 1.f;
        if (clean >= interval) { scale = fminf(scale * growth, 16777216.f); clean = 0.f; }
    }
    st[0] = scale; st[1] = 1.f / scale; st[2] = clean; st[3] = 0.f;
}

}  // namespace

extern "C" int sp_check_finite(const float* x, int64_t numel, float* found, sp_stream_t stream) {
    SP_CHECK_ARG(x && found && numel > 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "sp_check_finite: bad args (16-byte aligned pointer, numel > 0)");
    long blocks = (numel / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(check_finite_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, (long)(numel / 4), (long)numel, found);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_loss_scale_update(float* state, float growth, float backoff, int32_t interval, sp_stream_t stream) {
    SP_CHECK_ARG(state && growth >= 1.f && backoff > 0.f && backoff <= 1.f && interval > 0, "sp_loss_scale_update: bad args");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), state, growth, backoff, (float)interval);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_scale_f32_dev(float* x, int64_t numel, const float* factor_dev, sp_stream_t stream) {
    SP_CHECK_ARG(x && factor_dev && numel > 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "sp_scale_f32_dev: bad args (16-byte aligned pointer, numel > 0)");
    long blocks = (numel / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, (long)(numel / 4),
                       (long)numel, 1.f, factor_dev);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_scale_f32(float* x, int64_t numel, float factor, sp_stream_t stream) {
    SP_CHECK_ARG(x && numel > 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "sp_scale_f32: bad args (16-byte aligned pointer, numel > 0)");
    long blocks = (numel / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, (long)(numel / 4),
                       (long)numel, factor, (const float*)nullptr);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_adam_multi_guarded(const sp_adam_chunk* chunks_dev, int32_t n_chunks, double beta1, double beta2, double eps,
                                     double weight_decay, const float* skip_if_nonzero, sp_stream_t stream) {
    SP_CHECK_ARG(chunks_dev && n_chunks > 0, "sp_adam_multi: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, s, chunks_dev, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)eps, (float)weight_decay, skip_if_nonzero);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_adam_multi(const sp_adam_chunk* chunks_dev, int32_t n_chunks, double beta1, double beta2, double eps,
                             double weight_decay, sp_stream_t stream) {
    return sp_adam_multi_guarded(chunks_dev, n_chunks, beta1, beta2, eps, weight_decay, nullptr, stream);
}
