// Memory-bound glue kernels: image ingest, mask application, activations, small reductions.
// All of them are single-pass, channel-vectorised (4 elements per lane, 8/16-byte accesses) and
// grid-stride over at most 2048 blocks.
#include "common.h"

namespace {

inline int grid_for(long work_items) {
    long b = (work_items + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- image ingest: strided 3-channel source -> NHWC with zero-padded channels, optional affine ------
template <typename TS, typename TD>
__global__ void ingest_kernel(const TS* __restrict__ src, long sn, long sc, long sh, long sw, TD* __restrict__ dst,
                              int N, int C, int H, int W, int CP, float3 scale, float3 shift) {
    const long total = (long)N * H * W;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
        const int w = (int)(p % W);
        const long q = p / W;
        const int h = (int)(q % H);
        const int n = (int)(q / H);
        const TS* s = src + n * sn + h * sh + w * sw;
        const float sc3[3] = {scale.x, scale.y, scale.z}, sf3[3] = {shift.x, shift.y, shift.z};
        for (int c = 0; c < CP; ++c) {
            float v = 0.f;
            if (c < C) v = Elem<TS>::ld(s + c * sc) * sc3[c % 3] + sf3[c % 3];
            Elem<TD>::st(dst + p * CP + c, v);
        }
    }
}

// backward of ingest: dsrc[n,h,w,c] (NHWC, C channels, pitch C) = dy[n,h,w,c] * scale[c]
template <typename T>
__global__ void ingest_bwd_kernel(const T* __restrict__ dy, int CP, T* __restrict__ dsrc, int C, long pixels, float3 scale) {
    const float sc3[3] = {scale.x, scale.y, scale.z};
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long)gridDim.x * 256)
        for (int c = 0; c < C; ++c) Elem<T>::st(dsrc + p * C + c, Elem<T>::ld(dy + p * CP + c) * sc3[c % 3]);
}

// ---- feature * mask, concatenated with the mask channel (models.py:94), padded to CP channels ------
template <typename T>
__global__ void mask_concat_kernel(const T* __restrict__ feat, const float* __restrict__ mask, T* __restrict__ out,
                                   long pixels, int C, int CP) {
    const int vpp = CP / 4;
    const long total = pixels * vpp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / vpp;
        const int c = (int)(i - p * vpp) * 4;
        const float m = mask[p];
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c + 3 < C) {
            Elem<T>::ld4(feat + p * C + c, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= m;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (c + r < C) v[r] = Elem<T>::ld(feat + p * C + c + r) * m;
                else if (c + r == C) v[r] = m;
            }
        }
        Elem<T>::st4(out + p * CP + c, v);
    }
}

// ---- 2-D feature * mask (models.py:78,80) -----------------------------------------------------------
template <typename T>
__global__ void mask_mul_2d_kernel(const T* __restrict__ feat, int ldf, const float* __restrict__ mask, T* __restrict__ out,
                                   int ldo, int B, int K) {
    const long total = (long)B * ldo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / ldo), k = (int)(i % ldo);
        float v = 0.f;
        if (k < K) v = Elem<T>::ld(feat + (long)b * ldf + k) * mask[(long)b * K + k];
        Elem<T>::st(out + i, v);
    }
}

// ---- activation forward / backward from the POST-activation value ----------------------------------
template <typename T>
__global__ void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n4, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float v[4];
        Elem<T>::ld4(x + i * 4, v);
        apply_act_vec<4>(v, act);
        Elem<T>::st4(y + i * 4, v);
    }
}

__device__ __forceinline__ float act_grad(float dy, float y, int act) {
    if (act == SP_ACT_LRELU) return y > 0.f ? dy : 0.2f * dy;
    if (act == SP_ACT_RELU) return y > 0.f ? dy : 0.f;
    if (act == SP_ACT_TANH) return dy * (1.f - y * y);
    return dy;
}

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dz, long n4, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float d[4], v[4];
        Elem<T>::ld4(dy + i * 4, d);
        Elem<T>::ld4(y + i * 4, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = act_grad(d[r], v[r], act);
        Elem<T>::st4(dz + i * 4, d);
    }
}

// narrow-channel variant with re-pitch: dy,y pitch C -> dz pitch CP (zero padded).  Used for the tanh head.
template <typename T>
__global__ void act_bwd_pad_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dz, long pixels,
                                   int C, int CP, int act) {
    const long total = pixels * CP;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / CP;
        const int c = (int)(i - p * CP);
        float v = 0.f;
        if (c < C) v = act_grad(Elem<T>::ld(dy + p * C + c), Elem<T>::ld(y + p * C + c), act);
        Elem<T>::st(dz + i, v);
    }
}

// ---- y = g[0] * a + b (attention residual, models.py:274) and its backward -------------------------
template <typename T>
__global__ void scale_add_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ g,
                                 T* __restrict__ y, long n4) {
    const float gv = g[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float av[4], bv[4];
        Elem<T>::ld4(a + i * 4, av);
        Elem<T>::ld4(b + i * 4, bv);
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = gv * av[r] + bv[r];
        Elem<T>::st4(y + i * 4, av);
    }
}

// ---- y = (x - bias[c]) * (sigma_a / sigma_b) + bias[c]: the output of a spectral-normalised layer under ANOTHER forward's sigma -----
// (conv(x, W / sigma_b) + b = (conv(x, W / sigma_a) + b - b) * sigma_a / sigma_b + b: the layer is not run again, sp_rescale_bias)
template <typename T>
__global__ void rescale_bias_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, int c, int ldx, int ldy,
                                    const float* __restrict__ bias, const float* __restrict__ sig_a, const float* __restrict__ sig_b) {
    const float r = sig_a[0] * sig_b[1];                    // {sigma, 1 / sigma} pairs of the two forwards
    const int c4 = c / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * c4; i += (long)gridDim.x * 256) {
        const long row = i / c4;
        const int col = (int)(i - row * c4) * 4;
        float v[4];
        Elem<T>::ld4(x + row * ldx + col, v);
        const float4 bv = bias != nullptr ? *reinterpret_cast<const float4*>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[0] = (v[0] - bv.x) * r + bv.x; v[1] = (v[1] - bv.y) * r + bv.y; v[2] = (v[2] - bv.z) * r + bv.z; v[3] = (v[3] - bv.w) * r + bv.w;
        Elem<T>::st4(y + row * ldy + col, v);
    }
}

template <typename T>
__global__ void rescale_bias_scalar_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, int c, int ldx, int ldy,
                                           const float* __restrict__ bias, const float* __restrict__ sig_a, const float* __restrict__ sig_b) {
    const float r = sig_a[0] * sig_b[1];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * c; i += (long)gridDim.x * 256) {
        const long row = i / c;
        const int col = (int)(i - row * c);
        const float bv = bias != nullptr ? bias[col] : 0.f;
        Elem<T>::st(y + row * ldy + col, (Elem<T>::ld(x + row * ldx + col) - bv) * r + bv);
    }
}

template <typename T>
__global__ void scale_add_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ a, const float* __restrict__ g,
                                     T* __restrict__ da, float* __restrict__ dg_part, long n4) {
    __shared__ float red[4];
    const float gv = g[0];
    float part = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float d[4], av[4];
        Elem<T>::ld4(dy + i * 4, d);
        Elem<T>::ld4(a + i * 4, av);
#pragma unroll
        for (int r = 0; r < 4; ++r) { part += d[r] * av[r]; d[r] *= gv; }
        Elem<T>::st4(da + i * 4, d);
    }
    const float tot = block_sum_256(part, red);
    if (threadIdx.x == 0) dg_part[blockIdx.x] = tot;          // column_sum_kernel adds the blocks in order
}

// out[col] = sum over rows of part[row][col], rows taken in a fixed order (the second stage of the reductions of this file:
// per-block partial sums instead of atomics, so the results are bit-reproducible).  One block per column.
__global__ __launch_bounds__(256) void column_sum_kernel(const float* __restrict__ part, int rows, int cols, float* __restrict__ out) {
    __shared__ float red[4];
    const int col = blockIdx.x;
    float t = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) t += part[(long)r * cols + col];
    const float tot = block_sum_256(t, red);
    if (threadIdx.x == 0) out[col] = tot;
}

// ---- (B, C*HW) <-> (B, HW*C) permutation (NCHW flatten order <-> NHWC) ------------------------------
template <typename T>
__global__ void permute_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int C, int HW, int to_hwc) {
    const long total = (long)B * C * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / ((long)C * HW));
        const int r = (int)(i - (long)b * C * HW);
        // i enumerates the DESTINATION
        long s;
        if (to_hwc) { const int hw = r / C, c = r - hw * C; s = (long)b * C * HW + (long)c * HW + hw; }
        else { const int c = r / HW, hw = r - c * HW; s = (long)b * C * HW + (long)hw * C + c; }
        dst[i] = src[s];
    }
}

// ---- per-channel sum over pixels (bias gradients) ---------------------------------------------------
// thread = (4-channel group, pixel lane); pixel lanes are combined in LDS, then one partial row per block (out = [blocks][C]).
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ x, int ld, long pixels, int C,
                                                          float* __restrict__ out) {
    __shared__ float red[256 * 4];
    const int ngroups = (C + 3) / 4;
    const int lanes_per_pix = ngroups < 256 ? ngroups : 256;
    const int pix_par = 256 / lanes_per_pix;
    const int cg = threadIdx.x % lanes_per_pix, pl = threadIdx.x / lanes_per_pix;
    const bool vec = (ld & 3) == 0;
    for (int cbase = 0; cbase < ngroups; cbase += lanes_per_pix) {
        const int c = (cbase + cg) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const bool live = c < C && pl < pix_par;
        if (live) {
            for (long p = (long)blockIdx.x * pix_par + pl; p < pixels; p += (long)gridDim.x * pix_par) {
                if (vec) {
                    float v[4];
                    Elem<T>::ld4(x + p * ld + c, v);
                    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
                } else {
                    for (int r = 0; r < 4 && c + r < C; ++r) acc[r] += Elem<T>::ld(x + p * ld + c + r);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) red[threadIdx.x * 4 + r] = acc[r];
        __syncthreads();
        if (live && pl == 0) {
            for (int r = 0; r < 4 && c + r < C; ++r) {
                float t = 0.f;
                for (int k = 0; k < pix_par; ++k) t += red[(k * lanes_per_pix + cg) * 4 + r];
                out[(long)blockIdx.x * C + c + r] = t;
            }
        }
    }
}

}  // namespace

#define SP_DT_SWITCH(dtype, CALL_F32, CALL_BF16) do { if ((dtype) == SP_F32) { CALL_F32; } else { CALL_BF16; } } while (0)

extern "C" int sp_ingest_image(const void* src, int32_t src_dtype, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                               void* dst, int32_t n, int32_t c, int32_t h, int32_t w_, int32_t cp, const float* scale3,
                               const float* shift3, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(src && dst && c <= 3 && cp >= c, "sp_ingest_image: bad args");
    SP_CHECK_ARG((src_dtype == SP_F32 || src_dtype == SP_BF16) && (dtype == SP_F32 || dtype == SP_BF16), "sp_ingest_image: bad dtype");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float3 sc3 = make_float3(1.f, 1.f, 1.f), sf3 = make_float3(0.f, 0.f, 0.f);
    if (scale3) sc3 = make_float3(scale3[0], scale3[1], scale3[2]);     // host pointers (3 floats)
    if (shift3) sf3 = make_float3(shift3[0], shift3[1], shift3[2]);
    const int g = grid_for((long)n * h * w_);
    if (src_dtype == SP_F32 && dtype == SP_F32)
        hipLaunchKernelGGL((ingest_kernel<float, float>), dim3(g), dim3(256), 0, s, (const float*)src, sn, sc, sh, sw, (float*)dst, n, c, h, w_, cp, sc3, sf3);
    else if (src_dtype == SP_F32 && dtype == SP_BF16)
        hipLaunchKernelGGL((ingest_kernel<float, bf16>), dim3(g), dim3(256), 0, s, (const float*)src, sn, sc, sh, sw, (bf16*)dst, n, c, h, w_, cp, sc3, sf3);
    else if (src_dtype == SP_BF16 && dtype == SP_BF16)
        hipLaunchKernelGGL((ingest_kernel<bf16, bf16>), dim3(g), dim3(256), 0, s, (const bf16*)src, sn, sc, sh, sw, (bf16*)dst, n, c, h, w_, cp, sc3, sf3);
    else
        hipLaunchKernelGGL((ingest_kernel<bf16, float>), dim3(g), dim3(256), 0, s, (const bf16*)src, sn, sc, sh, sw, (float*)dst, n, c, h, w_, cp, sc3, sf3);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_ingest_image_bwd(const void* dy, int32_t cp, void* dsrc, int32_t c, int64_t pixels, const float* scale3,
                                   int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(dy && dsrc && c <= 3 && cp >= c, "sp_ingest_image_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float3 sc3 = make_float3(1.f, 1.f, 1.f);
    if (scale3) sc3 = make_float3(scale3[0], scale3[1], scale3[2]);
    const int g = grid_for(pixels);
    SP_DT_SWITCH(dtype,
                 hipLaunchKernelGGL(ingest_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)dy, cp, (float*)dsrc, c, (long)pixels, sc3),
                 hipLaunchKernelGGL(ingest_bwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)dy, cp, (bf16*)dsrc, c, (long)pixels, sc3));
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_mask_concat(const void* feat, const float* mask, void* out, int64_t pixels, int32_t c, int32_t cp,
                              int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(feat && mask && out && cp > c && cp % 4 == 0 && c % 4 == 0, "sp_mask_concat: bad args (c=%d cp=%d)", c, cp);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int g = grid_for(pixels * (cp / 4));
    SP_DT_SWITCH(dtype,
                 hipLaunchKernelGGL(mask_concat_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)feat, mask, (float*)out, (long)pixels, c, cp),
                 hipLaunchKernelGGL(mask_concat_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)feat, mask, (bf16*)out, (long)pixels, c, cp));
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_mask_mul_2d(const void* feat, int32_t ldf, const float* mask, void* out, int32_t ldo, int32_t batch,
                              int32_t k, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(feat && mask && out && ldf >= k && ldo >= k, "sp_mask_mul_2d: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int g = grid_for((long)batch * ldo);
    SP_DT_SWITCH(dtype,
                 hipLaunchKernelGGL(mask_mul_2d_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)feat, ldf, mask, (float*)out, ldo, batch, k),
                 hipLaunchKernelGGL(mask_mul_2d_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)feat, ldf, mask, (bf16*)out, ldo, batch, k));
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_act_fwd(const void* x, void* y, int64_t numel, int32_t act, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && y && numel % 4 == 0, "sp_act_fwd: numel must be a multiple of 4");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int g = grid_for(numel / 4);
    SP_DT_SWITCH(dtype,
                 hipLaunchKernelGGL(act_fwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, (float*)y, (long)(numel / 4), act),
                 hipLaunchKernelGGL(act_fwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)y, (long)(numel / 4), act));
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_act_bwd(const void* dy, const void* y, void* dz, int64_t pixels, int32_t c, int32_t cp, int32_t act,
                          int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(dy && y && dz && cp >= c, "sp_act_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (c == cp && (pixels * c) % 4 == 0) {
        const long n4 = pixels * c / 4;
        const int g = grid_for(n4);
        SP_DT_SWITCH(dtype,
                     hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)dy, (const float*)y, (float*)dz, n4, act),
                     hipLaunchKernelGGL(act_bwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)dy, (const bf16*)y, (bf16*)dz, n4, act));
    } else {
        const int g = grid_for(pixels * cp);
        SP_DT_SWITCH(dtype,
                     hipLaunchKernelGGL(act_bwd_pad_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)dy, (const float*)y, (float*)dz, (long)pixels, c, cp, act),
                     hipLaunchKernelGGL(act_bwd_pad_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)dy, (const bf16*)y, (bf16*)dz, (long)pixels, c, cp, act));
    }
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_scale_add(const void* a, const void* b, const float* g, void* y, int64_t numel, int32_t dtype,
                            sp_stream_t stream) {
    SP_CHECK_ARG(a && b && g && y && numel % 4 == 0, "sp_scale_add: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int gr = grid_for(numel / 4);
    SP_DT_SWITCH(dtype,
                 hipLaunchKernelGGL(scale_add_kernel<float>, dim3(gr), dim3(256), 0, s, (const float*)a, (const float*)b, g, (float*)y, (long)(numel / 4)),
                 hipLaunchKernelGGL(scale_add_kernel<bf16>, dim3(gr), dim3(256), 0, s, (const bf16*)a, (const bf16*)b, g, (bf16*)y, (long)(numel / 4)));
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_rescale_bias(const void* x, void* y, int64_t rows, int32_t c, int32_t ldx, int32_t ldy, const float* bias,
                               const float* sig_a, const float* sig_b, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && y && sig_a && sig_b && rows > 0 && c > 0 && ldx >= c && ldy >= c, "sp_rescale_bias: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if ((c & 3) || (ldx & 3) || (ldy & 3)) {                 // odd widths (the 365-wide linear mapping): element by element
        const int g1 = grid_for(rows * c);
        SP_DT_SWITCH(dtype,
                     hipLaunchKernelGGL(rescale_bias_scalar_kernel<float>, dim3(g1), dim3(256), 0, s, (const float*)x, (float*)y, (long)rows, c, ldx, ldy, bias, sig_a, sig_b),
                     hipLaunchKernelGGL(rescale_bias_scalar_kernel<bf16>, dim3(g1), dim3(256), 0, s, (const bf16*)x, (bf16*)y, (long)rows, c, ldx, ldy, bias, sig_a, sig_b));
        SP_LAUNCH_CHECK();
        return SP_OK;
    }
    const int gr = grid_for(rows * (c / 4));
    SP_DT_SWITCH(dtype,
                 hipLaunchKernelGGL(rescale_bias_kernel<float>, dim3(gr), dim3(256), 0, s, (const float*)x, (float*)y, (long)rows, c, ldx, ldy, bias, sig_a, sig_b),
                 hipLaunchKernelGGL(rescale_bias_kernel<bf16>, dim3(gr), dim3(256), 0, s, (const bf16*)x, (bf16*)y, (long)rows, c, ldx, ldy, bias, sig_a, sig_b));
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_scale_add_bwd(const void* dy, const void* a, const float* g, void* da, float* dg, float* partials, int64_t numel,
                                int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(dy && a && g && da && dg && partials && numel % 4 == 0, "sp_scale_add_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // <= 512 grid-striding blocks, one partial sum each (thousands of blocks ending in one same-address atomic serialised
    // there: 31 us for a 31 MB pass); the second kernel adds them in block order
    int gr = grid_for(numel / 4);
    if (gr > 512) gr = 512;
    SP_DT_SWITCH(dtype,
                 hipLaunchKernelGGL(scale_add_bwd_kernel<float>, dim3(gr), dim3(256), 0, s, (const float*)dy, (const float*)a, g, (float*)da, partials, (long)(numel / 4)),
                 hipLaunchKernelGGL(scale_add_bwd_kernel<bf16>, dim3(gr), dim3(256), 0, s, (const bf16*)dy, (const bf16*)a, g, (bf16*)da, partials, (long)(numel / 4)));
    hipLaunchKernelGGL(column_sum_kernel, dim3(1), dim3(256), 0, s, partials, gr, 1, dg);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_permute_chw_hwc(const void* src, void* dst, int32_t batch, int32_t c, int32_t hw, int32_t to_hwc,
                                  int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(src && dst && batch > 0 && c > 0 && hw > 0, "sp_permute_chw_hwc: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int g = grid_for((long)batch * c * hw);
    SP_DT_SWITCH(dtype,
                 hipLaunchKernelGGL(permute_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)src, (float*)dst, batch, c, hw, to_hwc),
                 hipLaunchKernelGGL(permute_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)src, (bf16*)dst, batch, c, hw, to_hwc));
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_channel_sum(const void* x, int32_t ld, int64_t pixels, int32_t c, float* out, float* partials, int32_t dtype,
                              sp_stream_t stream) {
    SP_CHECK_ARG(x && out && partials && ld >= c && c > 0, "sp_channel_sum: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int groups = (c + 3) / 4, lanes = groups < 256 ? groups : 256, pix_par = 256 / lanes;
    long blocks = pixels / ((long)pix_par * 32);
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    SP_DT_SWITCH(dtype,
                 hipLaunchKernelGGL(channel_sum_kernel<float>, dim3((int)blocks), dim3(256), 0, s, (const float*)x, ld, (long)pixels, c, partials),
                 hipLaunchKernelGGL(channel_sum_kernel<bf16>, dim3((int)blocks), dim3(256), 0, s, (const bf16*)x, ld, (long)pixels, c, partials));
    hipLaunchKernelGGL(column_sum_kernel, dim3(c), dim3(256), 0, s, partials, (int)blocks, c, out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Training masks on the device (replaces misc.get_masks_for_training, /root/reference/misc.py:13-68, which
// runs per sample in DataLoader workers).  One block per sample; all randomness is INTEGER arithmetic on a counter-based
// generator (splitmix64 of (seed, sample, draw index)), so oracle/sempyr_oracle.py restates it bit for bit:
//   draw 0: stage = [0,1,2,3,4,5,6,0,1][r % 9]                (misc.py:28: random.choice(list(range(7)) + [0, 1]), deep end first)
//   draw 1: spatial = (r >> 40) < p_random * 2^24  and  0 < stage < 6        (misc.py:32-34)
//   draw 2: number of shapes 1 + r % 4                                          (misc.py:38-39: min_shapes 1, max_shapes 4)
//   draws 3 + 4k .. 6 + 4k (shape k): bounding box h = lo + r % (base - lo + 1), w likewise, y0 = r % (base - h + 1), x0 likewise,
//            on the level just finer than the stage (side `base`, lo = min(8, base / 2): misc.py:37-41)
//   draw 19 + k: kind of shape k = r % 4 - rectangle, circle, triangle, ellipse, the four kinds skimage.draw.random_shapes chooses
//            from (misc.py:37; skimage itself is not available offline, so its generator cannot be restated - the mask CONTRACT is
//            the reference's: zeros inside shapes, ones outside, nearest-neighbour expansion to every finer level, misc.py:45,55).
//            Inside its bounding box (dy, dx from the box corner, pixel centres, all in integers so the oracle restates it exactly):
//            rectangle: every pixel;  ellipse (inscribed): (2dy+1-h)^2 w^2 + (2dx+1-w)^2 h^2 <= h^2 w^2;
//            circle (inscribed, diameter d = min(h, w), centred): (2dy+1-h)^2 + (2dx+1-w)^2 <= d^2;
//            triangle (apex top centre, base = bottom edge): |2dx+1-w| * 2h <= w * (2dy+1)
// Level idx (0 = the 365-vector ... 6 = 128 x 128): ones where idx == stage, zeros where idx < stage; idx > stage: zeros, or
// with `spatial` the shape map read through the nearest-neighbour index i * base / side.  Values are exact 0.0f / 1.0f.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long sp_mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned long long sp_draw(unsigned long long seed, unsigned sample, unsigned k) {
    return sp_mix64(sp_mix64(seed ^ ((unsigned long long)sample << 32)) + k);
}

struct sp_mask_ptrs { float* m[7]; };          // list order of the reference: 128^2, 64^2, 32^2, 16^2, 8^2, 4096, 365

static __global__ __launch_bounds__(256) void training_masks_kernel(sp_mask_ptrs out, unsigned long long seed, unsigned thresh24) {
    const unsigned b = blockIdx.x;
    const int stage_tab[9] = {0, 1, 2, 3, 4, 5, 6, 0, 1};
    const int side_of[7] = {1, 1, 8, 16, 32, 64, 128};         // side of level idx (counted from the deep end); 0 / 1 are vectors
    const int numel_of[7] = {365, 4096, 64, 256, 1024, 4096, 16384};
    const int stage = stage_tab[sp_draw(seed, b, 0) % 9];
    const bool spatial = (unsigned)(sp_draw(seed, b, 1) >> 40) < thresh24 && stage > 0 && stage < 6;
    const int base = side_of[stage + 1 > 6 ? 6 : stage + 1];
    const int lo = base / 2 < 8 ? base / 2 : 8;
    int nrect = 0, ry[4], rx[4], rh[4], rw[4], kind[4];
    if (spatial) {
        nrect = 1 + (int)(sp_draw(seed, b, 2) % 4);
        for (int k = 0; k < 4; ++k) {
            rh[k] = lo + (int)(sp_draw(seed, b, 3 + 4 * k) % (unsigned)(base - lo + 1));
            rw[k] = lo + (int)(sp_draw(seed, b, 4 + 4 * k) % (unsigned)(base - lo + 1));
            ry[k] = (int)(sp_draw(seed, b, 5 + 4 * k) % (unsigned)(base - rh[k] + 1));
            rx[k] = (int)(sp_draw(seed, b, 6 + 4 * k) % (unsigned)(base - rw[k] + 1));
            kind[k] = (int)(sp_draw(seed, b, 19 + k) % 4);
        }
    }
    auto inside = [&](int k, int y, int x) {
        const int dy = y - ry[k], dx = x - rx[k], h = rh[k], w = rw[k];
        if (dy < 0 || dy >= h || dx < 0 || dx >= w) return false;
        const long long ey = 2 * dy + 1 - h, ex = 2 * dx + 1 - w;
        if (kind[k] == 0) return true;                                                              // rectangle
        if (kind[k] == 1) { const long long d = h < w ? h : w; return ey * ey + ex * ex <= d * d; }   // circle
        if (kind[k] == 2) return (ex < 0 ? -ex : ex) * 2 * h <= (long long)w * (2 * dy + 1);        // triangle
        return ey * ey * w * w + ex * ex * h * h <= (long long)h * h * w * w;                        // ellipse
    };
    for (int idx = 0; idx < 7; ++idx) {
        float* dst = out.m[6 - idx] + (long)b * numel_of[idx];
        const int n = numel_of[idx], side = side_of[idx];
        if (idx <= stage || !spatial) {
            const float v = idx == stage ? 1.f : 0.f;
            for (int i = threadIdx.x; i < n; i += 256) dst[i] = v;
            continue;
        }
        for (int i = threadIdx.x; i < n; i += 256) {
            const int y = (i / side) * base / side, x = (i % side) * base / side;
            bool hit = false;
            for (int k = 0; k < nrect; ++k) hit |= inside(k, y, x);
            dst[i] = hit ? 0.f : 1.f;
        }
    }
}

// kornia.normalize_min_max(image, -1, 1) of the data pipeline (/root/reference/data.py:53):
//   y = (hi - lo) * (x - min) / (max - min + eps) + lo,  min / max over one plane (per_channel) or over the whole image.
// One block per (image, channel) or per image; every operation in the reference's order and in fp32: bit-identical to torch.
static __global__ __launch_bounds__(256) void minmax_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, long plane, int planes,
                                                                      float lo, float hi, float eps) {
    __shared__ float smin[4], smax[4];
    const long n = plane * planes;
    const float* src = x + (long)blockIdx.x * n;
    float* dst = y + (long)blockIdx.x * n;
    float mn = INFINITY, mx = -INFINITY;
    for (long i = threadIdx.x; i < n; i += 256) { const float v = src[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    mn = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    const float range = hi - lo;
    const float den = (mx - mn) + eps;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float t = range * (src[i] - mn);
        dst[i] = __fadd_rn(__fdiv_rn(t, den), lo);          // explicit roundings: no contraction into an fma, IEEE division
    }
}

extern "C" int sp_minmax_normalize(const float* x, float* y, int32_t batch, int32_t channels, int64_t hw, float lo, float hi, float eps,
                                   int32_t per_channel, sp_stream_t stream) {
    SP_CHECK_ARG(x && y && batch > 0 && channels > 0 && hw > 0, "sp_minmax_normalize: bad args");
    const int blocks = per_channel ? batch * channels : batch;
    hipLaunchKernelGGL(minmax_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, (long)hw,
                       per_channel ? 1 : channels, lo, hi, eps);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_training_masks(float* m128, float* m64, float* m32, float* m16, float* m8, float* m4096, float* m365,
                                 int32_t batch, uint64_t seed, float p_random_mask, sp_stream_t stream) {
    SP_CHECK_ARG(m128 && m64 && m32 && m16 && m8 && m4096 && m365 && batch > 0, "sp_training_masks: bad args");
    SP_CHECK_ARG(p_random_mask >= 0.f && p_random_mask <= 1.f, "sp_training_masks: p_random_mask outside [0, 1]");
    sp_mask_ptrs o;
    o.m[0] = m128; o.m[1] = m64; o.m[2] = m32; o.m[3] = m16; o.m[4] = m8; o.m[5] = m4096; o.m[6] = m365;
    const unsigned thresh = (unsigned)((double)p_random_mask * 16777216.0);
    hipLaunchKernelGGL(training_masks_kernel, dim3((unsigned)batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), o,
                       (unsigned long long)seed, thresh);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
