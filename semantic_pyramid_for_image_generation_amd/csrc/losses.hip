// Discriminator head and the four losses of the training step (fp32 math, tiny tensors except the
// reconstruction / diversity reductions, which are single-pass streaming reads).
//   D head        models.py:149-155   pred[i][j][c] = x[j][c] * E[cls[i]][c] + (wc . x[j] + bc)   -> (B,B,F) quirk kept
//   LSGAN         lossfunction.py:137,164         0.5 * mean((p - t)^2)
//   reconstruction lossfunction.py:31-68           sum_levels mean(|maxpool2(real) - maxpool2(fake)| * maxpool2(mask))
//   diversity     lossfunction.py:92-110          mean|z1 - z2| / (mean|img1 - img2| + 1e-8)
#include "common.h"

namespace {

inline int ew_grid(long items) { long b = (items + 255) / 256; return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b)); }
// reductions that end in one atomic per block on a single address: few, long-running blocks (the atomics serialise)
inline int red_grid(long items) { const int g = ew_grid(items); return g > 256 ? 256 : g; }

// ---------------- discriminator head ----------------
// one block per class row i of pred[i][j][c] (the (B,B,F) broadcast of discriminator.py's projection head)
template <typename T>
__global__ void dhead_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ E, const int64_t* __restrict__ cls,
                                 const float* __restrict__ wc, const float* __restrict__ bc, float* __restrict__ pred, int B, int F) {
    extern __shared__ float cl[];       // [B] classification logits (recomputed by every block: B*F MACs)
    for (int j = threadIdx.x >> 6; j < B; j += 4) {
        float a = 0.f;
        for (int c = threadIdx.x & 63; c < F; c += 64) a += wc[c] * Elem<T>::ld(x + (long)j * ldx + c);
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) cl[j] = a + bc[0];
    }
    __syncthreads();
    const int i = blockIdx.x;
    const float* Ei = E + cls[i] * F;
    for (int e = threadIdx.x; e < B * F; e += 256) {
        const int j = e / F, c = e - j * F;
        pred[(long)i * B * F + e] = Elem<T>::ld(x + (long)j * ldx + c) * Ei[c] + cl[j];
    }
}

// one block per sample b: dx row b, the embedding-gradient row of class cls[b] and sdp[b] = sum_{i,c} dpred[i][b][c]; a second,
// single-block kernel forms dwc / dbc from all sdp (a first version let block 0 compute all B sums itself - a serial chain of
// B*B*F/64 dependent loads, 55 of the kernel's 59 us; a second one used a process-global "last block" ticket, which made the
// entry point non-reentrant across streams).  sdp lives in caller scratch.
// Embedding gradient without atomics: samples of the same class share a row of dE; the FIRST sample of a class sums the
// contributions of all its samples in batch order and is the only writer of that row.
template <typename T>
__global__ void dhead_bwd_kernel(const float* __restrict__ dpred, const T* __restrict__ x, int ldx, const float* __restrict__ E,
                                 const int64_t* __restrict__ cls, const float* __restrict__ wc, T* __restrict__ dx, int lddx,
                                 float* __restrict__ dE, float* __restrict__ sdp_out, int B, int F) {
    // Every loop below walks the batch with two loads per trip; written plainly each trip waited for its own loads (~50 dependent
    // round trips, 18 - 23 us per launch for a few hundred kilobytes).  The class indices sit in LDS and the loads of FOUR trips are
    // issued together; the additions keep their order (results are bit-identical).
    __shared__ float red[4];
    __shared__ int cls_s[1024];                                // (the entry point admits batch <= 1024)
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < B; i += 256) cls_s[i] = (int)cls[i];
    {
        float a = 0.f;
        const int total = B * F;
        for (int e0 = threadIdx.x; e0 < total; e0 += 4 * 256) {       // (i, c) pairs of column b
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 256;
                const int i = e / F, c = e - i * F;
                v[u] = e < total ? dpred[((long)i * B + b) * F + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (e0 + u * 256 < total) a += v[u];
        }
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    }
    __syncthreads();
    const float sdp_b = red[0] + red[1] + red[2] + red[3];
    const int my_cls = cls_s[b];
    bool first = true;
    for (int i = 0; i < b; ++i) first = first && cls_s[i] != my_cls;      // block-uniform
    for (int c = threadIdx.x; c < F; c += 256) {
        float a = wc[c] * sdp_b;                               // dx[b][c]
        for (int i0 = 0; i0 < B; i0 += 4) {
            float dp[4], ev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                dp[u] = i < B ? dpred[((long)i * B + b) * F + c] : 0.f;
                ev[u] = i < B ? E[(long)cls_s[i] * F + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i0 + u < B) a += dp[u] * ev[u];
        }
        Elem<T>::st(dx + (long)b * lddx + c, a);
        if (first) {
            float g = 0.f;                                     // dE[k][c] = sum_{i: cls[i] = k} sum_j dpred[i][j][c] * x[j][c]
            for (int i = b; i < B; ++i) {
                if (cls_s[i] != my_cls) continue;
                for (int j0 = 0; j0 < B; j0 += 4) {
                    float dp[4], xv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = j0 + u;
                        dp[u] = j < B ? dpred[((long)i * B + j) * F + c] : 0.f;
                        xv[u] = j < B ? Elem<T>::ld(x + (long)j * ldx + c) : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (j0 + u < B) g += dp[u] * xv[u];
                }
            }
            dE[(long)my_cls * F + c] += g;
        }
    }
    if (threadIdx.x == 0) sdp_out[b] = sdp_b;
}

template <typename T>
__global__ void dhead_bwd_cls_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ sdp_in, float* __restrict__ dwc,
                                     float* __restrict__ dbc, int B, int F) {
    extern __shared__ float sdp[];      // [B]
    for (int j = threadIdx.x; j < B; j += 256) sdp[j] = sdp_in[j];
    __syncthreads();
    for (int c = threadIdx.x; c < F; c += 256) {
        float a = 0.f;
        for (int j = 0; j < B; ++j) a += Elem<T>::ld(x + (long)j * ldx + c) * sdp[j];
        dwc[c] = a;
    }
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int j = 0; j < B; ++j) a += sdp[j];
        dbc[0] = a;
    }
}

// ---------------- 0.5 * mean((p - t)^2) ----------------
__global__ void sqerr_fwd_kernel(const float* __restrict__ p, long n, float target, double* __restrict__ acc) {
    __shared__ float red[4];
    float part = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float d = p[i] - target; part += d * d; }
    const float tot = block_sum_256(part, red);
    if (threadIdx.x == 0) atomicAdd(acc, (double)tot * 0.5 / (double)n);
}
// small inputs (the discriminator's (B, B, F) prediction): ONE block, fixed summation order, no accumulator / second launch.
// Four independent 16-byte loads per thread and trip (a single dependent load per trip made this 14 us for 51 200 values).
__global__ __launch_bounds__(1024) void sqerr_fwd_small_kernel(const float* __restrict__ p, int n, float target, float* __restrict__ loss) {
    __shared__ double red[16];
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    const int n4 = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? n / 4 : 0;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    int i = threadIdx.x;
    for (; i + 3 * 1024 < n4; i += 4 * 1024) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p4[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float a = v[u].x - target, b = v[u].y - target, c = v[u].z - target, d = v[u].w - target;
            part[u] += (a * a + b * b) + (c * c + d * d);
        }
    }
    for (; i < n4; i += 1024) {
        const float4 v = p4[i];
        const float a = v.x - target, b = v.y - target, c = v.z - target, d = v.w - target;
        part[0] += (a * a + b * b) + (c * c + d * d);
    }
    for (int j = n4 * 4 + threadIdx.x; j < n; j += 1024) { const float d = p[j] - target; part[1] += d * d; }
    double t = ((double)part[0] + (double)part[1]) + ((double)part[2] + (double)part[3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < 16; ++k) tot += red[k];
        loss[0] = (float)(tot * 0.5 / (double)n);
    }
}
__global__ void sqerr_bwd_kernel(const float* __restrict__ p, long n, float target, const float* __restrict__ gout, float* __restrict__ dp) {
    const float g = gout[0] / (float)n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dp[i] = g * (p[i] - target);
}
__global__ void dbl_to_f32_kernel(const double* __restrict__ a, float* __restrict__ o, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = (float)a[i];
}

// ---------------- semantic reconstruction, 4-D level ----------------
template <typename T>
__device__ __forceinline__ void max4(const T* b, long W, int C, float o[4]) {
    float t[4];
    Elem<T>::ld4(b, o);
    Elem<T>::ld4(b + C, t); for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], t[r]);
    Elem<T>::ld4(b + W * C, t); for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], t[r]);
    Elem<T>::ld4(b + W * C + C, t); for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], t[r]);
}

// this thread's share of sum |maxpool(real) - maxpool(fake)| * maxpool(mask) when block `blk` of `nblk` walks the level
template <typename T>
__device__ __forceinline__ float rec4d_fwd_part(const T* __restrict__ real, const T* __restrict__ fake, const float* __restrict__ mask, int N,
                                                int H, int W, int C, int blk, int nblk) {
    const int OH = H / 2, OW = W / 2, vpp = C / 4;
    const long total = (long)N * OH * OW * vpp;
    float part = 0.f;
    for (long i = (long)blk * 256 + threadIdx.x; i < total; i += (long)nblk * 256) {
        const int c = (int)(i % vpp) * 4;
        const long pp = i / vpp;
        const int ow = (int)(pp % OW);
        const long q = pp / OW;
        const int oh = (int)(q % OH), n = (int)(q / OH);
        const long pix = ((long)n * H + oh * 2) * W + ow * 2;
        const float m = fmaxf(fmaxf(mask[pix], mask[pix + 1]), fmaxf(mask[pix + W], mask[pix + W + 1]));
        if (m == 0.f) continue;
        float r[4], f[4];
        max4(real + pix * C + c, (long)W, C, r);
        max4(fake + pix * C + c, (long)W, C, f);
        for (int k = 0; k < 4; ++k) part += fabsf((r[k] - f[k]) * m);
    }
    return part;
}
template <typename T>
__global__ void rec4d_fwd_kernel(const T* __restrict__ real, const T* __restrict__ fake, const float* __restrict__ mask, int N, int H,
                                 int W, int C, double* __restrict__ acc) {
    __shared__ float red[4];
    const float part = rec4d_fwd_part<T>(real, fake, mask, N, H, W, C, (int)blockIdx.x, (int)gridDim.x);
    const float tot = block_sum_256(part, red);
    if (threadIdx.x == 0) atomicAdd(acc, (double)tot / (double)((long)N * (H / 2) * (W / 2) * C));
}

// dfake: -g * sign((r - f) * m) * m / count at the first maximum of each fake window, 0 elsewhere
template <typename T>
__device__ __forceinline__ void rec4d_bwd_part(const T* __restrict__ real, const T* __restrict__ fake, const float* __restrict__ mask,
                                               float gup, T* __restrict__ dfake, int N, int H, int W, int C, int blk, int nblk) {
    const int OH = H / 2, OW = W / 2, vpp = C / 4;
    const long total = (long)N * OH * OW * vpp;
    const float g = gup / (float)((long)N * OH * OW * C);
    for (long i = (long)blk * 256 + threadIdx.x; i < total; i += (long)nblk * 256) {
        const int c = (int)(i % vpp) * 4;
        const long pp = i / vpp;
        const int ow = (int)(pp % OW);
        const long q = pp / OW;
        const int oh = (int)(q % OH), n = (int)(q / OH);
        const long pix = ((long)n * H + oh * 2) * W + ow * 2;
        const long offs[4] = {0, C, (long)W * C, (long)W * C + C};
        const float m = fmaxf(fmaxf(mask[pix], mask[pix + 1]), fmaxf(mask[pix + W], mask[pix + W + 1]));
        float o[4][4];
        float fv[4][4], r[4];
        for (int k = 0; k < 4; ++k) Elem<T>::ld4(fake + pix * C + c + offs[k], fv[k]);
        max4(real + pix * C + c, (long)W, C, r);
        for (int e = 0; e < 4; ++e) {
            int best = 0;
            float fm = fv[0][e];
            for (int k = 1; k < 4; ++k) if (fv[k][e] > fm) { fm = fv[k][e]; best = k; }
            const float d = (r[e] - fm) * m;
            const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            for (int k = 0; k < 4; ++k) o[k][e] = (k == best) ? -g * s * m : 0.f;
        }
        for (int k = 0; k < 4; ++k) Elem<T>::st4(dfake + pix * C + c + offs[k], o[k]);
    }
}
template <typename T>
__global__ void rec4d_bwd_kernel(const T* __restrict__ real, const T* __restrict__ fake, const float* __restrict__ mask,
                                 const float* __restrict__ gout, T* __restrict__ dfake, int N, int H, int W, int C) {
    rec4d_bwd_part<T>(real, fake, mask, gout[0], dfake, N, H, W, C, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------- semantic reconstruction, 2-D level (MaxPool1d(2) over pairs) ----------------
template <typename T>
__device__ __forceinline__ float rec2d_fwd_part(const T* __restrict__ real, int ldr, const T* __restrict__ fake, int ldf,
                                                const float* __restrict__ mask, int B, int K, int blk, int nblk) {
    const int KH = K / 2;
    float part = 0.f;
    for (long i = (long)blk * 256 + threadIdx.x; i < (long)B * KH; i += (long)nblk * 256) {
        const int b = (int)(i / KH), k = (int)(i % KH) * 2;
        const float m = fmaxf(mask[(long)b * K + k], mask[(long)b * K + k + 1]);
        const float r = fmaxf(Elem<T>::ld(real + (long)b * ldr + k), Elem<T>::ld(real + (long)b * ldr + k + 1));
        const float f = fmaxf(Elem<T>::ld(fake + (long)b * ldf + k), Elem<T>::ld(fake + (long)b * ldf + k + 1));
        part += fabsf((r - f) * m);
    }
    return part;
}
template <typename T>
__global__ void rec2d_fwd_kernel(const T* __restrict__ real, int ldr, const T* __restrict__ fake, int ldf, const float* __restrict__ mask,
                                 int B, int K, double* __restrict__ acc) {
    __shared__ float red[4];
    const float part = rec2d_fwd_part<T>(real, ldr, fake, ldf, mask, B, K, (int)blockIdx.x, (int)gridDim.x);
    const float tot = block_sum_256(part, red);
    if (threadIdx.x == 0) atomicAdd(acc, (double)tot / (double)((long)B * (K / 2)));
}

template <typename T>
__device__ __forceinline__ void rec2d_bwd_part(const T* __restrict__ real, int ldr, const T* __restrict__ fake, int ldf,
                                               const float* __restrict__ mask, float gup, T* __restrict__ dfake, int ldd, int B, int K, int blk,
                                               int nblk) {
    const int KH = K / 2;
    const float g = gup / (float)((long)B * KH);
    for (long i = (long)blk * 256 + threadIdx.x; i < (long)B * ((K + 1) / 2); i += (long)nblk * 256) {
        const int b = (int)(i / ((K + 1) / 2)), k = (int)(i % ((K + 1) / 2)) * 2;
        if (k + 1 >= K) { Elem<T>::st(dfake + (long)b * ldd + k, 0.f); continue; }     // odd tail is dropped by the pool
        const float m = fmaxf(mask[(long)b * K + k], mask[(long)b * K + k + 1]);
        const float r = fmaxf(Elem<T>::ld(real + (long)b * ldr + k), Elem<T>::ld(real + (long)b * ldr + k + 1));
        const float f0 = Elem<T>::ld(fake + (long)b * ldf + k), f1 = Elem<T>::ld(fake + (long)b * ldf + k + 1);
        const int best = f1 > f0 ? 1 : 0;
        const float d = (r - (best ? f1 : f0)) * m;
        const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        Elem<T>::st(dfake + (long)b * ldd + k + best, -g * s * m);
        Elem<T>::st(dfake + (long)b * ldd + k + 1 - best, 0.f);
    }
}
template <typename T>
__global__ void rec2d_bwd_kernel(const T* __restrict__ real, int ldr, const T* __restrict__ fake, int ldf, const float* __restrict__ mask,
                                 const float* __restrict__ gout, T* __restrict__ dfake, int ldd, int B, int K) {
    rec2d_bwd_part<T>(real, ldr, fake, ldf, mask, gout[0], dfake, ldd, B, K, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------- all pyramid levels of the semantic reconstruction loss in one launch ----------------
// The level descriptors travel as a kernel argument; block b serves the level whose block range holds it.
constexpr int REC_MAX_LEVELS = 8;
struct RecLevels { sp_rec_level lv[REC_MAX_LEVELS]; int first_block[REC_MAX_LEVELS + 1]; int n; };

template <typename T>
__global__ void rec_fwd_levels_kernel(RecLevels a, double* __restrict__ acc) {
    __shared__ float red[4];
    int l = 0;
    while (l + 1 < a.n && (int)blockIdx.x >= a.first_block[l + 1]) ++l;
    const int blk = (int)blockIdx.x - a.first_block[l], nblk = a.first_block[l + 1] - a.first_block[l];
    const sp_rec_level& L = a.lv[l];
    float part;
    double count;
    if (L.h > 1 || L.w_ > 1) {
        part = rec4d_fwd_part<T>((const T*)L.real, (const T*)L.fake, L.mask, L.n, L.h, L.w_, L.c, blk, nblk);
        count = (double)((long)L.n * (L.h / 2) * (L.w_ / 2) * L.c);
    } else {
        part = rec2d_fwd_part<T>((const T*)L.real, L.ld_real, (const T*)L.fake, L.ld_fake, L.mask, L.n, L.c, blk, nblk);
        count = (double)((long)L.n * (L.c / 2));
    }
    const float tot = block_sum_256(part, red);
    if (threadIdx.x == 0) atomicAdd(acc, (double)tot / count);
}
template <typename T>
__global__ void rec_bwd_levels_kernel(RecLevels a, const float* __restrict__ gout, float weight) {
    int l = 0;
    while (l + 1 < a.n && (int)blockIdx.x >= a.first_block[l + 1]) ++l;
    const int blk = (int)blockIdx.x - a.first_block[l], nblk = a.first_block[l + 1] - a.first_block[l];
    const sp_rec_level& L = a.lv[l];
    const float gup = gout[0] * weight;
    if (L.h > 1 || L.w_ > 1) rec4d_bwd_part<T>((const T*)L.real, (const T*)L.fake, L.mask, gup, (T*)L.dfake, L.n, L.h, L.w_, L.c, blk, nblk);
    else rec2d_bwd_part<T>((const T*)L.real, L.ld_real, (const T*)L.fake, L.ld_fake, L.mask, gup, (T*)L.dfake, L.ld_dfake, L.n, L.c, blk, nblk);
}
// out[i] = weight * (float)acc[i]; the accumulators go back to zero (the contract of the one-launch forms: zero on entry and exit)
__global__ void loss_finalize_kernel(double* __restrict__ acc, float* __restrict__ out, int n, float weight) {
    const int i = threadIdx.x;
    if (i < n) { out[i] = (float)acc[i] * weight; acc[i] = 0.0; }
}

// ---------------- diversity loss ----------------
template <typename T>
__global__ void div_fwd_kernel(const T* __restrict__ img, long half_elems, const float* __restrict__ z, long half_z,
                               double* __restrict__ acc /* [0]=sum|img1-img2| [1]=sum|z1-z2| */) {
    __shared__ float red[4];
    float pi = 0.f, pz = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < half_elems; i += (long)gridDim.x * 256)
        pi += fabsf(Elem<T>::ld(img + i) - Elem<T>::ld(img + half_elems + i));
    if (blockIdx.x == 0)
        for (long i = threadIdx.x; i < half_z; i += 256) pz += fabsf(z[i] - z[half_z + i]);
    const float ti = block_sum_256(pi, red);
    const float tz = block_sum_256(pz, red);
    if (threadIdx.x == 0) { atomicAdd(acc, (double)ti); if (blockIdx.x == 0) atomicAdd(acc + 1, (double)tz); }
}
__global__ void div_finalize_kernel(double* __restrict__ acc, long half_elems, long half_z, float* __restrict__ out, float weight, int rezero) {
    const double den = acc[0] / (double)half_elems, num = acc[1] / (double)half_z;
    out[0] = (float)(num / (den + 1e-8)) * weight;
    out[1] = (float)(-num / ((den + 1e-8) * (den + 1e-8)) / (double)half_elems) * weight;     // d loss / d |img1-img2| element
    if (rezero) { acc[0] = 0.0; acc[1] = 0.0; }
}
template <typename T>
__global__ void div_bwd_kernel(const T* __restrict__ img, long half_elems, const float* __restrict__ fwd_out, const float* __restrict__ gout,
                               T* __restrict__ dimg) {
    const float coef = gout[0] * fwd_out[1];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < half_elems; i += (long)gridDim.x * 256) {
        const float d = Elem<T>::ld(img + i) - Elem<T>::ld(img + half_elems + i);
        const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        Elem<T>::st(dimg + i, coef * s);
        Elem<T>::st(dimg + half_elems + i, -coef * s);
    }
}

}  // namespace

extern "C" int sp_dhead_fwd(const void* x, int32_t ldx, const float* emb_sn, const int64_t* cls, const float* wc,
                            const float* bc, float* pred, int32_t batch, int32_t f, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && emb_sn && cls && wc && bc && pred && batch > 0 && batch <= 1024 && f > 0, "sp_dhead_fwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SP_F32) hipLaunchKernelGGL(dhead_fwd_kernel<float>, dim3(batch), dim3(256), batch * 4, s, (const float*)x, ldx, emb_sn, cls, wc, bc, pred, batch, f);
    else hipLaunchKernelGGL(dhead_fwd_kernel<bf16>, dim3(batch), dim3(256), batch * 4, s, (const bf16*)x, ldx, emb_sn, cls, wc, bc, pred, batch, f);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_dhead_bwd(const float* dpred, const void* x, int32_t ldx, const float* emb_sn, const int64_t* cls,
                            const float* wc, void* dx, int32_t lddx, float* demb, int32_t num_classes, float* dwc,
                            float* dbc, float* scratch, int32_t batch, int32_t f, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(dpred && x && emb_sn && cls && wc && dx && demb && dwc && dbc && scratch && batch > 0 && batch <= 1024 && f > 0, "sp_dhead_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(demb, 0, sizeof(float) * (size_t)num_classes * f, s);
    if (e != hipSuccess) { sp_set_error("sp_dhead_bwd: memset failed"); return SP_ERR_LAUNCH; }
    if (dtype == SP_F32) {
        hipLaunchKernelGGL(dhead_bwd_kernel<float>, dim3(batch), dim3(256), 0, s, dpred, (const float*)x, ldx, emb_sn, cls, wc, (float*)dx, lddx, demb, scratch, batch, f);
        hipLaunchKernelGGL(dhead_bwd_cls_kernel<float>, dim3(1), dim3(256), batch * 4, s, (const float*)x, ldx, scratch, dwc, dbc, batch, f);
    } else {
        hipLaunchKernelGGL(dhead_bwd_kernel<bf16>, dim3(batch), dim3(256), 0, s, dpred, (const bf16*)x, ldx, emb_sn, cls, wc, (bf16*)dx, lddx, demb, scratch, batch, f);
        hipLaunchKernelGGL(dhead_bwd_cls_kernel<bf16>, dim3(1), dim3(256), batch * 4, s, (const bf16*)x, ldx, scratch, dwc, dbc, batch, f);
    }
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_sqerr_loss_fwd(const float* p, int64_t numel, float target, double* acc_tmp, float* loss,
                                 sp_stream_t stream) {
    SP_CHECK_ARG(p && acc_tmp && loss && numel > 0, "sp_sqerr_loss_fwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (numel <= (1 << 18)) {
        hipLaunchKernelGGL(sqerr_fwd_small_kernel, dim3(1), dim3(1024), 0, s, p, (int)numel, target, loss);
        SP_LAUNCH_CHECK();
        return SP_OK;
    }
    if (hipMemsetAsync(acc_tmp, 0, sizeof(double), s) != hipSuccess) { sp_set_error("sp_sqerr_loss_fwd: memset failed"); return SP_ERR_LAUNCH; }
    hipLaunchKernelGGL(sqerr_fwd_kernel, dim3(red_grid(numel)), dim3(256), 0, s, p, (long)numel, target, acc_tmp);
    hipLaunchKernelGGL(dbl_to_f32_kernel, dim3(1), dim3(256), 0, s, acc_tmp, loss, 1);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_sqerr_loss_bwd(const float* p, int64_t numel, float target, const float* gout, float* dp,
                                 sp_stream_t stream) {
    SP_CHECK_ARG(p && gout && dp && numel > 0, "sp_sqerr_loss_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sqerr_bwd_kernel, dim3(ew_grid(numel)), dim3(256), 0, s, p, (long)numel, target, gout, dp);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_rec_loss_fwd(const void* real, int32_t ld_real, const void* fake, int32_t ld_fake, const float* mask,
                               int32_t n, int32_t h, int32_t w_, int32_t c, double* acc, int32_t dtype,
                               sp_stream_t stream) {
    SP_CHECK_ARG(real && fake && mask && acc && n > 0 && c > 0, "sp_rec_loss_fwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (h > 1 || w_ > 1) {
        SP_CHECK_ARG(c % 4 == 0 && h % 2 == 0 && w_ % 2 == 0 && ld_real == c && ld_fake == c, "sp_rec_loss_fwd: 4-D level needs even H,W and dense C%%4==0");
        const int g = red_grid((long)n * (h / 2) * (w_ / 2) * (c / 4));
        if (dtype == SP_F32) hipLaunchKernelGGL(rec4d_fwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)real, (const float*)fake, mask, n, h, w_, c, acc);
        else hipLaunchKernelGGL(rec4d_fwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)real, (const bf16*)fake, mask, n, h, w_, c, acc);
    } else {
        const int g = red_grid((long)n * (c / 2));
        if (dtype == SP_F32) hipLaunchKernelGGL(rec2d_fwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)real, ld_real, (const float*)fake, ld_fake, mask, n, c, acc);
        else hipLaunchKernelGGL(rec2d_fwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)real, ld_real, (const bf16*)fake, ld_fake, mask, n, c, acc);
    }
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_rec_loss_bwd(const void* real, int32_t ld_real, const void* fake, int32_t ld_fake, const float* mask,
                               const float* gout, void* dfake, int32_t ld_dfake, int32_t n, int32_t h, int32_t w_,
                               int32_t c, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(real && fake && mask && gout && dfake && n > 0 && c > 0, "sp_rec_loss_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (h > 1 || w_ > 1) {
        SP_CHECK_ARG(c % 4 == 0 && h % 2 == 0 && w_ % 2 == 0 && ld_real == c && ld_fake == c && ld_dfake == c, "sp_rec_loss_bwd: 4-D level needs even H,W and dense C%%4==0");
        const int g = ew_grid((long)n * (h / 2) * (w_ / 2) * (c / 4));
        if (dtype == SP_F32) hipLaunchKernelGGL(rec4d_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)real, (const float*)fake, mask, gout, (float*)dfake, n, h, w_, c);
        else hipLaunchKernelGGL(rec4d_bwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)real, (const bf16*)fake, mask, gout, (bf16*)dfake, n, h, w_, c);
    } else {
        const int g = ew_grid((long)n * ((c + 1) / 2));
        if (dtype == SP_F32) hipLaunchKernelGGL(rec2d_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)real, ld_real, (const float*)fake, ld_fake, mask, gout, (float*)dfake, ld_dfake, n, c);
        else hipLaunchKernelGGL(rec2d_bwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)real, ld_real, (const bf16*)fake, ld_fake, mask, gout, (bf16*)dfake, ld_dfake, n, c);
    }
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_f64_to_f32(const double* src, float* dst, int32_t n, sp_stream_t stream) {
    SP_CHECK_ARG(src && dst && n > 0, "sp_f64_to_f32: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(dbl_to_f32_kernel, dim3(sp_div_up(n, 256)), dim3(256), 0, s, src, dst, n);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_div_loss_fwd(const void* img, int64_t half_elems, const float* z, int64_t half_z, double* acc_tmp,
                               float* out2, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(img && z && acc_tmp && out2 && half_elems > 0 && half_z > 0, "sp_div_loss_fwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(acc_tmp, 0, 2 * sizeof(double), s) != hipSuccess) { sp_set_error("sp_div_loss_fwd: memset failed"); return SP_ERR_LAUNCH; }
    const int g = red_grid(half_elems);
    if (dtype == SP_F32) hipLaunchKernelGGL(div_fwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)img, (long)half_elems, z, (long)half_z, acc_tmp);
    else hipLaunchKernelGGL(div_fwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)img, (long)half_elems, z, (long)half_z, acc_tmp);
    hipLaunchKernelGGL(div_finalize_kernel, dim3(1), dim3(1), 0, s, acc_tmp, (long)half_elems, (long)half_z, out2, 1.0f, 0);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_div_loss_fwd_w(const void* img, int64_t half_elems, const float* z, int64_t half_z, double* acc,
                                 float* out2, float weight, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(img && z && acc && out2 && half_elems > 0 && half_z > 0, "sp_div_loss_fwd_w: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int g = red_grid(half_elems);
    if (dtype == SP_F32) hipLaunchKernelGGL(div_fwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)img, (long)half_elems, z, (long)half_z, acc);
    else hipLaunchKernelGGL(div_fwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)img, (long)half_elems, z, (long)half_z, acc);
    hipLaunchKernelGGL(div_finalize_kernel, dim3(1), dim3(1), 0, s, acc, (long)half_elems, (long)half_z, out2, weight, 1);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

static int rec_levels_plan(const sp_rec_level* levels, int n_levels, bool bwd, RecLevels& a, const char* who) {
    SP_CHECK_ARG(levels && n_levels > 0 && n_levels <= REC_MAX_LEVELS, "%s: 1..%d levels", who, REC_MAX_LEVELS);
    a.n = n_levels;
    a.first_block[0] = 0;
    for (int l = 0; l < n_levels; ++l) {
        const sp_rec_level& L = levels[l];
        SP_CHECK_ARG(L.real && L.fake && L.mask && L.n > 0 && L.c > 0 && (!bwd || L.dfake), "%s: level %d: bad args", who, l);
        long items;
        if (L.h > 1 || L.w_ > 1) {
            SP_CHECK_ARG(L.c % 4 == 0 && L.h % 2 == 0 && L.w_ % 2 == 0 && L.ld_real == L.c && L.ld_fake == L.c && (!bwd || L.ld_dfake == L.c),
                         "%s: level %d: a 4-D level needs even H, W and dense C %% 4 == 0", who, l);
            items = (long)L.n * (L.h / 2) * (L.w_ / 2) * (L.c / 4);
        } else {
            items = (long)L.n * (bwd ? (L.c + 1) / 2 : L.c / 2);
        }
        a.lv[l] = L;
        a.first_block[l + 1] = a.first_block[l] + (bwd ? ew_grid(items) : red_grid(items));
    }
    return SP_OK;
}

extern "C" int sp_rec_loss_fwd_levels(const sp_rec_level* levels, int32_t n_levels, double* acc, float* loss, float weight,
                                      int32_t dtype, sp_stream_t stream) {
    RecLevels a;
    const int rc = rec_levels_plan(levels, n_levels, false, a, "sp_rec_loss_fwd_levels");
    if (rc != SP_OK) return rc;
    SP_CHECK_ARG(acc && loss, "sp_rec_loss_fwd_levels: null pointer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SP_F32) hipLaunchKernelGGL(rec_fwd_levels_kernel<float>, dim3(a.first_block[a.n]), dim3(256), 0, s, a, acc);
    else hipLaunchKernelGGL(rec_fwd_levels_kernel<bf16>, dim3(a.first_block[a.n]), dim3(256), 0, s, a, acc);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, acc, loss, 1, weight);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_rec_loss_bwd_levels(const sp_rec_level* levels, int32_t n_levels, const float* gout, float weight,
                                      int32_t dtype, sp_stream_t stream) {
    RecLevels a;
    const int rc = rec_levels_plan(levels, n_levels, true, a, "sp_rec_loss_bwd_levels");
    if (rc != SP_OK) return rc;
    SP_CHECK_ARG(gout, "sp_rec_loss_bwd_levels: null pointer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SP_F32) hipLaunchKernelGGL(rec_bwd_levels_kernel<float>, dim3(a.first_block[a.n]), dim3(256), 0, s, a, gout, weight);
    else hipLaunchKernelGGL(rec_bwd_levels_kernel<bf16>, dim3(a.first_block[a.n]), dim3(256), 0, s, a, gout, weight);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_div_loss_bwd(const void* img, int64_t half_elems, const float* fwd_out2, const float* gout, void* dimg,
                               int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(img && fwd_out2 && gout && dimg && half_elems > 0, "sp_div_loss_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int g = ew_grid(half_elems);
    if (dtype == SP_F32) hipLaunchKernelGGL(div_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)img, (long)half_elems, fwd_out2, gout, (float*)dimg);
    else hipLaunchKernelGGL(div_bwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)img, (long)half_elems, fwd_out2, gout, (bf16*)dimg);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
