// (Conditional) BatchNorm in training mode, NHWC.  Replaces nn.BatchNorm2d at models.py:53,484 and the
// class-gathered affine of ConditionalBatchNorm.forward (models.py:498-506), plus their autograd.
//   stats  : per-channel sum / sum-of-squares.  Thread = (16-byte channel group, pixel lane): every wave reads
//            whole pixels (coalesced 16-byte loads), accumulates in fp32 registers; pixel lanes are combined in
//            LDS and each block writes ONE partial row [2C]; a second tiny kernel adds the <= 256 partial rows
//            in fp64 (deterministic, no atomics, no memset).
//   apply  : y = act(scale[n,c] * (x - mean) * invstd + bias[n,c])  with (scale,bias) = emb[cls[n]] (CBN,
//            embedding row = [scale(C) | bias(C)]) or (gamma, beta) (plain BN); LeakyReLU fused.
//   bwd    : the two per-(sample, channel) reductions use the same partial-row scheme, then one elementwise pass.
#include "common.h"

namespace {

constexpr int BN_MAX_PARTS = 1024;     // partial rows = blocks of the reduction kernels: 4 per CU (256 left them at 2.5 TB/s)

struct Affine {
    const float* gamma;     // plain BN: gamma[c], beta[c]
    const float* beta;
    const float* emb;       // CBN: emb[cls[n]][c], emb[cls[n]][C + c]
    const int64_t* cls;
    __device__ __forceinline__ void get(int n, int c, int C, float& s, float& b) const {
        if (emb) { const float* row = emb + (long)cls[n] * 2 * C; s = row[c]; b = row[C + c]; }
        else { s = gamma ? gamma[c] : 1.f; b = beta ? beta[c] : 0.f; }
    }
    // V consecutive channels (c % 4 == 0, C % 4 == 0: 16-byte loads - one per four channels instead of one per channel and array)
    template <int V>
    __device__ __forceinline__ void getv(int n, int c, int C, float (&s)[V], float (&b)[V]) const;
};
template <int V> __device__ __forceinline__ void ldv(const float* p, float (&o)[V]) {
#pragma unroll
    for (int k = 0; k < V / 4; ++k) { const float4 t = *reinterpret_cast<const float4*>(p + 4 * k); o[4 * k] = t.x; o[4 * k + 1] = t.y; o[4 * k + 2] = t.z; o[4 * k + 3] = t.w; }
}
template <int V>
__device__ __forceinline__ void Affine::getv(int n, int c, int C, float (&s)[V], float (&b)[V]) const {
    if (emb) { const float* row = emb + (long)cls[n] * 2 * C; ldv<V>(row + c, s); ldv<V>(row + C + c, b); return; }
#pragma unroll
    for (int r = 0; r < V; ++r) { s[r] = 1.f; b[r] = 0.f; }
    if (gamma) ldv<V>(gamma + c, s);
    if (beta) ldv<V>(beta + c, b);
}

// thread layout: lanes_per_pix channel groups side by side, pix_par pixels per block step
struct Lay { int cg, pl, lanes_per_pix, pix_par; };
__device__ __forceinline__ Lay make_lay(int ngroups) {
    Lay l;
    l.lanes_per_pix = ngroups < 256 ? ngroups : 256;
    l.pix_par = 256 / l.lanes_per_pix;
    l.cg = threadIdx.x % l.lanes_per_pix;
    l.pl = threadIdx.x / l.lanes_per_pix;
    return l;
}

// part[blockIdx.x][c][0..1] = (sum x, sum x^2) over this block's pixels
// blockIdx.y = group of a two-group batch (sp_bn_stats_pair): images [0, split) / [split, n) of one tensor, each with its own
// statistics - pixels_b / part_stride describe the second group (a one-group launch has gridDim.y == 1)
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, long pixels, int C, float* __restrict__ part,
                                                       long pixels_b = 0, long part_stride = 0) {
    __shared__ float red[256 * 2 * V];
    if (blockIdx.y == 1) { x += pixels * C; part += part_stride; pixels = pixels_b; }
    const int ngroups = C / V;
    const Lay L = make_lay(ngroups);
    for (int cbase = 0; cbase < ngroups; cbase += L.lanes_per_pix) {
        const int c = (cbase + L.cg) * V;
        float s[V], q[V];
#pragma unroll
        for (int r = 0; r < V; ++r) { s[r] = 0.f; q[r] = 0.f; }
        const bool live = c < C && L.pl < L.pix_par;
        if (live) {
#pragma unroll 4
            for (long p = (long)blockIdx.x * L.pix_par + L.pl; p < pixels; p += (long)gridDim.x * L.pix_par) {
                float v[V];
                VecIO<T, V>::ld(x + p * C + c, v);
#pragma unroll
                for (int r = 0; r < V; ++r) { s[r] += v[r]; q[r] += v[r] * v[r]; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < V; ++r) { red[threadIdx.x * 2 * V + r] = s[r]; red[threadIdx.x * 2 * V + V + r] = q[r]; }
        __syncthreads();
        if (live && L.pl == 0) {
            for (int r = 0; r < V; ++r) {
                float ts = 0.f, tq = 0.f;
                for (int k = 0; k < L.pix_par; ++k) {
                    ts += red[(k * L.lanes_per_pix + L.cg) * 2 * V + r];
                    tq += red[(k * L.lanes_per_pix + L.cg) * 2 * V + V + r];
                }
                part[((long)blockIdx.x * C + c + r) * 2] = ts;
                part[((long)blockIdx.x * C + c + r) * 2 + 1] = tq;
            }
        }
    }
}

// mean / invstd from the batch (training; adds the partial rows in fp64) or the running statistics (eval);
// running-stat update follows torch: running = (1-m)*running + m*batch, with the UNBIASED batch variance
// (models.py:484 momentum=0.001).  Block = 8 channels x 32 partial lanes: the partial-row walk is a chain of dependent fp64
// additions, so its length (rows / lanes) and the number of blocks (C / 8) decide the time, not the 64-byte row segments
// (32 channels x 8 lanes: 9.4 us per launch, 33 launches per step).
constexpr int FIN_CL = 8, FIN_NL = 256 / FIN_CL;
constexpr int BN_EMB_MAXN = 256;         // largest batch of the conditional (class-embedding) backward
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nparts, long count, int C, float eps,
                                                          float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, int training,
                                                          float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                          int groups = 1, long count_b = 0, long part_stride = 0, int first_group = 0) {
    __shared__ double red[256 * 2];
    const int cl = threadIdx.x % FIN_CL, bl = threadIdx.x / FIN_CL;
    const int c = blockIdx.x * FIN_CL + cl;
    if (groups == 2) {
        // two-group batch (sp_bn_stats_pair): both groups' statistics in this launch; the running statistics take the two batches ONE
        // AFTER THE OTHER, first_group first - the order in which the reference runs the two forwards (training mode only)
        for (int step = 0; step < 2; ++step) {
            const int g = step == 0 ? first_group : 1 - first_group;
            const float2* p2 = reinterpret_cast<const float2*>(part + (long)g * part_stride);
            const long cnt = g == 0 ? count : count_b;
            double gs = 0.0, gq = 0.0;
            if (c < C) {
#pragma unroll 8
                for (int b = bl; b < nparts; b += FIN_NL) { const float2 t = p2[(long)b * C + c]; gs += t.x; gq += t.y; }
            }
            __syncthreads();
            red[threadIdx.x * 2] = gs;
            red[threadIdx.x * 2 + 1] = gq;
            __syncthreads();
            if (bl == 0 && c < C) {
                for (int k = 1; k < FIN_NL; ++k) { gs += red[(k * FIN_CL + cl) * 2]; gq += red[(k * FIN_CL + cl) * 2 + 1]; }
                const double m = gs / (double)cnt;
                double var = gq / (double)cnt - m * m;
                if (var < 0.0) var = 0.0;
                mean_out[(long)g * C + c] = (float)m;
                invstd_out[(long)g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
                if (running_mean) {
                    const double unb = cnt > 1 ? var * (double)cnt / (double)(cnt - 1) : var;
                    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
                    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
                }
            }
        }
        return;
    }
    double s = 0.0, q = 0.0;
    if (training && c < C)
    {
        // 8-byte loads, eight in flight (the plain loop waited for every load: 32 dependent round trips, 7 us per launch)
        const float2* p2 = reinterpret_cast<const float2*>(part);
#pragma unroll 8
        for (int b = bl; b < nparts; b += FIN_NL) { const float2 t = p2[(long)b * C + c]; s += t.x; q += t.y; }
    }
    red[threadIdx.x * 2] = s;
    red[threadIdx.x * 2 + 1] = q;
    __syncthreads();
    if (bl != 0 || c >= C) return;
    if (training) {
        for (int k = 1; k < FIN_NL; ++k) { s += red[(k * FIN_CL + cl) * 2]; q += red[(k * FIN_CL + cl) * 2 + 1]; }
        const double m = s / (double)count;
        double var = q / (double)count - m * m;
        if (var < 0.0) var = 0.0;
        mean_out[c] = (float)m;
        invstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
        }
    } else {
        mean_out[c] = running_mean[c];
        invstd_out[c] = 1.f / sqrtf(running_var[c] + eps);
    }
}

// grid (pixel slabs, samples): a thread keeps ONE channel group for the whole kernel, so mean / invstd / scale / bias
// are loaded once and the loop body is load -> fma -> store.
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, long hw, int C,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd, Affine aff,
                                                       int act, int split = 0x7fffffff) {
    const int n = blockIdx.y;
    if (n >= split) { mean += C; invstd += C; }              // second group of a two-group batch: its own statistics ([2][C] arrays)
    const int ngroups = C / V;
    const Lay L = make_lay(ngroups);
    const T* xn = x + (long)n * hw * C;
    T* yn = y + (long)n * hw * C;
    for (int cbase = 0; cbase < ngroups; cbase += L.lanes_per_pix) {
        const int c = (cbase + L.cg) * V;
        if (c >= C || L.pl >= L.pix_par) continue;
        float a[V], b[V];                                    // y = a * x + b
        {
            float mu[V], is[V];
            aff.template getv<V>(n, c, C, a, b);
            ldv<V>(mean + c, mu); ldv<V>(invstd + c, is);
#pragma unroll
            for (int r = 0; r < V; ++r) { a[r] *= is[r]; b[r] -= mu[r] * a[r]; }
        }
#pragma unroll 4
        for (long p = (long)blockIdx.x * L.pix_par + L.pl; p < hw; p += (long)gridDim.x * L.pix_par) {
            float v[V];
            VecIO<T, V>::ld(xn + p * C + c, v);
#pragma unroll
            for (int r = 0; r < V; ++r) v[r] = fmaf(a[r], v[r], b[r]);
            apply_act_vec<V>(v, act);
            VecIO<T, V>::st(yn + p * C + c, v);
        }
    }
}

// per (sample, block): part[n][blockIdx.x][c][0] = sum_hw dz, [1] = sum_hw dz * xhat; dz = dy * act'(z)
// BatchNorm apply (+ activation) fused with the bilinear x2 upsampling that follows it in a generator block (models.py:
// 296-298: CBN -> LeakyReLU -> UpsamplingBilinear2d -> conv): each output pixel normalises and activates its four source
// pixels on the fly, so the activated low-resolution tensor is neither written nor read back.
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_apply_upsample2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 Affine aff, int act, int split = 0x7fffffff) {
    // (split: samples >= split take the second row of [2][C] statistics - a batch of two groups, sp_bn_apply_upsample2_pair)
    // grid (column slabs, N * OH): a block owns (part of) ONE output row, so the sample, the two source rows and the row weight
    // are block-uniform, and a thread keeps one channel group: its parameters are loaded once (16-byte loads), the loop body is
    // four loads -> fma / activation -> blend -> store without any integer division (the flat-index form decoded (n, oh, ow, c)
    // per output vector and reloaded 4 V scalars: 0.36 TB/s)
    const int OH = 2 * H, OW = 2 * W;
    const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    for (int row = blockIdx.y; row < N * OH; row += gridDim.y) {
    const int n = row / OH, oh = row - n * OH;
    const float fh = sh * oh;
    const int h0 = (int)fh;
    const int h1 = h0 + (h0 < H - 1 ? 1 : 0);
    const float lh = fh - h0;
    const int ngroups = C / V;
    const Lay L = make_lay(ngroups);
    for (int cbase = 0; cbase < ngroups; cbase += L.lanes_per_pix) {
        const int c = (cbase + L.cg) * V;
        if (c >= C || L.pl >= L.pix_par) continue;
        float a[V], b[V];
        {
            float mu[V], is[V];
            aff.template getv<V>(n, c, C, a, b);
            const int go = n >= split ? C : 0;
            ldv<V>(mean + go + c, mu); ldv<V>(invstd + go + c, is);
#pragma unroll
            for (int r = 0; r < V; ++r) { a[r] *= is[r]; b[r] -= mu[r] * a[r]; }
        }
        const T* r0 = x + ((long)n * H + h0) * W * C + c;
        const T* r1 = x + ((long)n * H + h1) * W * C + c;
        T* yr = y + ((long)n * OH + oh) * OW * C + c;
        // a thread produces the four output columns 4q - 3 ... 4q from the THREE source columns 2q - 2 ... 2q they interpolate between
        // (bn_up2_kernel below explains the pattern): 6 loads and 6 normalisations per 4 outputs instead of 16 each.  Rows are blended
        // first, then columns: the reference's order up to fp32 rounding.
        for (int q = blockIdx.x * L.pix_par + L.pl; 4 * q - 3 < OW; q += gridDim.x * L.pix_par) {
            float t[3][V];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int col = 2 * q - 2 + k;
                col = col < 0 ? 0 : (col < W ? col : W - 1);
                float s0[V], s1[V];
                VecIO<T, V>::ld(r0 + (long)col * C, s0);
                VecIO<T, V>::ld(r1 + (long)col * C, s1);
#pragma unroll
                for (int r = 0; r < V; ++r) { s0[r] = fmaf(a[r], s0[r], b[r]); s1[r] = fmaf(a[r], s1[r], b[r]); }
                apply_act_vec<V>(s0, act); apply_act_vec<V>(s1, act);
#pragma unroll
                for (int r = 0; r < V; ++r) t[k][r] = (1.f - lh) * s0[r] + lh * s1[r];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ow = 4 * q - 3 + j;
                if (ow < 0 || ow >= OW) continue;
                const int w0 = 2 * q - 2 + (j >> 1);
                const float lw = ow == 0 ? 0.f : sw * ow - (float)w0;
                float o[V];
#pragma unroll
                for (int r = 0; r < V; ++r) o[r] = (1.f - lw) * t[j >> 1][r] + lw * t[(j >> 1) + 1][r];
                VecIO<T, V>::st(yr + (long)ow * C, o);
            }
        }
    }
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x, long hw, int C,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            Affine aff, int act, float* __restrict__ part) {
    __shared__ float red[256 * 2 * V];
    const int n = blockIdx.y;
    const int ngroups = C / V;
    const Lay L = make_lay(ngroups);
    const T* dyn = dy + (long)n * hw * C;
    const T* xn = x + (long)n * hw * C;
    float* prow = part + ((long)n * gridDim.x + blockIdx.x) * C * 2;
    for (int cbase = 0; cbase < ngroups; cbase += L.lanes_per_pix) {
        const int c = (cbase + L.cg) * V;
        const bool live = c < C && L.pl < L.pix_par;
        float a[V], b[V];
#pragma unroll
        for (int r = 0; r < V; ++r) { a[r] = 0.f; b[r] = 0.f; }
        if (live) {
            float sc[V], bi[V], mu[V], is[V];
            aff.template getv<V>(n, c, C, sc, bi);
            ldv<V>(mean + c, mu); ldv<V>(invstd + c, is);
#pragma unroll 2
            for (long p = (long)blockIdx.x * L.pix_par + L.pl; p < hw; p += (long)gridDim.x * L.pix_par) {
                float d[V], v[V];
                VecIO<T, V>::ld(dyn + p * C + c, d);
                VecIO<T, V>::ld(xn + p * C + c, v);
#pragma unroll
                for (int r = 0; r < V; ++r) {
                    const float xh = (v[r] - mu[r]) * is[r];
                    float dz = d[r];
                    if (act == SP_ACT_LRELU) dz = (sc[r] * xh + bi[r]) > 0.f ? dz : 0.2f * dz;
                    a[r] += dz;
                    b[r] += dz * xh;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < V; ++r) { red[threadIdx.x * 2 * V + r] = a[r]; red[threadIdx.x * 2 * V + V + r] = b[r]; }
        __syncthreads();
        if (live && L.pl == 0) {
            for (int r = 0; r < V; ++r) {
                float ta = 0.f, tb = 0.f;
                for (int k = 0; k < L.pix_par; ++k) {
                    ta += red[(k * L.lanes_per_pix + L.cg) * 2 * V + r];
                    tb += red[(k * L.lanes_per_pix + L.cg) * 2 * V + V + r];
                }
                prow[(c + r) * 2] = ta;
                prow[(c + r) * 2 + 1] = tb;
            }
        }
    }
}

// per channel: c1 = sum_n scale*A / M, c2 = sum_n scale*B / M; parameter gradients.  Block = 8 channels x 32 lanes,
// the lanes split the samples.
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nparts, int N, int C, long count,
                                                              Affine aff, float* __restrict__ c1, float* __restrict__ c2,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ demb, int num_classes) {
    __shared__ double red[256 * 4];
    __shared__ float embv[BN_EMB_MAXN][FIN_CL][2];       // per-sample (dscale, dbias) of this block's channels
    const int cl = threadIdx.x % FIN_CL, bl = threadIdx.x / FIN_CL;
    const int c = blockIdx.x * FIN_CL + cl;
    if (demb) {
        // the embedding gradient is dense [classes][2C] with at most N non-zero rows: this block clears ITS channels of every
        // row before it accumulates into them (it is their only writer) - no separate fill launch
        if (c < C)
            for (int k = bl; k < num_classes; k += FIN_NL) { demb[(long)k * 2 * C + c] = 0.f; demb[(long)k * 2 * C + C + c] = 0.f; }
        __syncthreads();
    }
    double s1 = 0.0, s2 = 0.0, ga = 0.0, gb = 0.0;
    if (c < C) {
        for (int n = bl; n < N; n += FIN_NL) {
            float sc, bi;
            aff.get(n, c, C, sc, bi);
            double a = 0.0, b = 0.0;
            const float2* p2 = reinterpret_cast<const float2*>(part) + (long)n * nparts * C + c;
#pragma unroll 8
            for (int k = 0; k < nparts; ++k) { const float2 t = p2[(long)k * C]; a += t.x; b += t.y; }
            s1 += sc * a;
            s2 += sc * b;
            ga += b;
            gb += a;
            if (demb) { embv[n][cl][0] = (float)b; embv[n][cl][1] = (float)a; }
        }
    }
    if (demb) {
        // samples of one class share a row: the FIRST sample of a class adds up all of them in batch order and stores the row
        // once (no atomics, reproducible; plain independent stores - a read-modify-write per sample serialised 20 dependent
        // memory round trips per launch).  The lanes of a channel split the samples.
        __syncthreads();
        if (c < C)
            for (int n = bl; n < N; n += FIN_NL) {
                const long k = aff.cls[n];
                bool first = true;
                for (int m = 0; m < n; ++m) first = first && aff.cls[m] != k;
                if (!first) continue;
                float s0 = embv[n][cl][0], s1 = embv[n][cl][1];
                for (int m = n + 1; m < N; ++m)
                    if (aff.cls[m] == k) { s0 += embv[m][cl][0]; s1 += embv[m][cl][1]; }
                demb[k * 2 * C + c] = s0;
                demb[k * 2 * C + C + c] = s1;
            }
    }
    red[threadIdx.x * 4] = s1; red[threadIdx.x * 4 + 1] = s2; red[threadIdx.x * 4 + 2] = ga; red[threadIdx.x * 4 + 3] = gb;
    __syncthreads();
    if (bl != 0 || c >= C) return;
    for (int k = 1; k < FIN_NL; ++k) {
        s1 += red[(k * FIN_CL + cl) * 4]; s2 += red[(k * FIN_CL + cl) * 4 + 1]; ga += red[(k * FIN_CL + cl) * 4 + 2]; gb += red[(k * FIN_CL + cl) * 4 + 3];
    }
    c1[c] = (float)(s1 / (double)count);
    c2[c] = (float)(s2 / (double)count);
    if (dgamma) dgamma[c] = (float)ga;
    if (dbeta) dbeta[c] = (float)gb;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx, long hw,
                                                           int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           Affine aff, int act, const float* __restrict__ c1,
                                                           const float* __restrict__ c2) {
    const int n = blockIdx.y;
    const int ngroups = C / V;
    const Lay L = make_lay(ngroups);
    const T* dyn = dy + (long)n * hw * C;
    const T* xn = x + (long)n * hw * C;
    T* dxn = dx + (long)n * hw * C;
    for (int cbase = 0; cbase < ngroups; cbase += L.lanes_per_pix) {
        const int c = (cbase + L.cg) * V;
        if (c >= C || L.pl >= L.pix_par) continue;
        float sc[V], bi[V], mu[V], is[V], k1[V], k2[V];
        aff.template getv<V>(n, c, C, sc, bi);
        ldv<V>(mean + c, mu); ldv<V>(invstd + c, is); ldv<V>(c1 + c, k1); ldv<V>(c2 + c, k2);
#pragma unroll 4
        for (long p = (long)blockIdx.x * L.pix_par + L.pl; p < hw; p += (long)gridDim.x * L.pix_par) {
            float d[V], v[V];
            VecIO<T, V>::ld(dyn + p * C + c, d);
            VecIO<T, V>::ld(xn + p * C + c, v);
#pragma unroll
            for (int r = 0; r < V; ++r) {
                const float xh = (v[r] - mu[r]) * is[r];
                float dz = d[r];
                if (act == SP_ACT_LRELU) dz = (sc[r] * xh + bi[r]) > 0.f ? dz : 0.2f * dz;
                d[r] = is[r] * (dz * sc[r] - k1[r] - xh * k2[r]);
            }
            VecIO<T, V>::st(dxn + p * C + c, d);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// BatchNorm of a tensor that is the bilinear x2 expansion (align_corners) of a stored one - the generator's final block,
// UpsamplingBilinear2d -> BatchNorm2d -> LeakyReLU (models.py:52-54) on 64 channels at 256 x 256: the expansion u = up2(x) is never
// written.  The separate passes move x + 4x (upsample) + 4x (statistics) + 8x (apply) = 17 units of the low-resolution tensor
// through HBM, these kernels x (statistics) + x + 4x (apply) = 6; the backward reads x instead of u twice.  One kernel body, four
// modes; every mode walks output rows with four output columns per thread step (8 loads per 4 interpolated vectors).
//   grid (column slabs, row slabs, samples); a thread keeps one 16-byte channel group (C / V <= 256 groups)
//   STATS      part[((n * gy + by) * gx + bx)][C][2] = (sum u, sum u^2) of the block's pixels      -> bn_finalize_kernel
//   APPLY      out = act(a u + b)
//   BWD_REDUCE part[n][by * gx + bx][C][2] = (sum dz, sum dz xhat), dz = dy act'(z)                 -> bn_bwd_finalize_kernel
//   BWD_APPLY  out = d loss / d u = invstd (dz scale - c1 - xhat c2)         (sp_upsample2_bwd then folds it back onto x)
// ------------------------------------------------------------------------------------------------------------------------------
enum { UP2_STATS = 0, UP2_APPLY = 1, UP2_BWD_REDUCE = 2, UP2_BWD_APPLY = 3 };
template <typename T, int V, int MODE>
__global__ __launch_bounds__(256) void bn_up2_kernel(const T* __restrict__ x, int H, int W, int C, const T* __restrict__ dy, T* __restrict__ out,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd, Affine aff, int act,
                                                     const float* __restrict__ c1, const float* __restrict__ c2, float* __restrict__ part) {
    __shared__ float red[(MODE == UP2_STATS || MODE == UP2_BWD_REDUCE) ? 256 * 2 * V : 1];
    const int OH = 2 * H, OW = 2 * W;
    const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const int n = blockIdx.z;
    const Lay L = make_lay(C / V);
    const int c = L.cg * V;
    const bool live = c < C && L.pl < L.pix_par;
    float acc0[V], acc1[V];
#pragma unroll
    for (int r = 0; r < V; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if (live) {
        float sc[V], bi[V], mu[V], is[V], k1[V], k2[V], a[V], b[V];
#pragma unroll
        for (int r = 0; r < V; ++r) { sc[r] = 1.f; bi[r] = 0.f; mu[r] = 0.f; is[r] = 1.f; k1[r] = 0.f; k2[r] = 0.f; }
        if constexpr (MODE != UP2_STATS) {
            aff.template getv<V>(n, c, C, sc, bi);
            ldv<V>(mean + c, mu); ldv<V>(invstd + c, is);
        }
        if constexpr (MODE == UP2_BWD_APPLY) { ldv<V>(c1 + c, k1); ldv<V>(c2 + c, k2); }
#pragma unroll
        for (int r = 0; r < V; ++r) { a[r] = sc[r] * is[r]; b[r] = bi[r] - mu[r] * a[r]; }
        for (int oh = blockIdx.y; oh < OH; oh += gridDim.y) {
            const float fh = sh * oh;
            const int h0 = (int)fh;
            const int h1 = h0 + (h0 < H - 1 ? 1 : 0);
            const float lh = fh - h0;
            const T* r0 = x + ((long)n * H + h0) * W * C + c;
            const T* r1 = x + ((long)n * H + h1) * W * C + c;
            const long orow = ((long)n * OH + oh) * OW;
            // output columns 2k + 1 and 2k + 2 both interpolate between source columns k and k + 1 (align_corners: column ow sits at
            // ow (W - 1) / (2W - 1)), so a thread takes the four outputs 4q - 3 ... 4q from THREE source columns 2q - 2, 2q - 1, 2q
            // with a fixed pattern - 6 loads per 4 vectors and no per-element selects (the first form, 8 loads and two select chains
            // per element, was bound by its VALU work: 80 - 110 us per pass on the 256 x 256 layer)
            for (int q = blockIdx.x * L.pix_par + L.pl; 4 * q - 3 < OW; q += gridDim.x * L.pix_par) {
                float t[3][V];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    int col = 2 * q - 2 + k;
                    col = col < 0 ? 0 : (col < W ? col : W - 1);
                    float s0[V], s1[V];
                    VecIO<T, V>::ld(r0 + (long)col * C, s0);
                    VecIO<T, V>::ld(r1 + (long)col * C, s1);
#pragma unroll
                    for (int r = 0; r < V; ++r) t[k][r] = (1.f - lh) * s0[r] + lh * s1[r];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ow = 4 * q - 3 + j;
                    if (ow < 0 || ow >= OW) continue;
                    const int w0 = 2 * q - 2 + (j >> 1);                       // = floor(ow (W - 1) / (2W - 1)) for ow >= 1
                    const float lw = ow == 0 ? 0.f : sw * ow - (float)w0;
                    float u[V];
#pragma unroll
                    for (int r = 0; r < V; ++r) u[r] = (1.f - lw) * t[j >> 1][r] + lw * t[(j >> 1) + 1][r];
                    if constexpr (sizeof(T) == 2) {
                        // the separate passes stored u in the 16-bit type and normalised THAT: round here too, so that the statistics, the
                        // output and the backward see one and the same tensor whether or not it is materialised
#pragma unroll
                        for (int r = 0; r < V; r += 2) {
                            const uint32_t w2 = f32x2_to_bf16x2(u[r], u[r + 1]);
                            u[r] = h16_lo_to_f32(w2);
                            u[r + 1] = h16_hi_to_f32(w2);
                        }
                    }
                    const long off = (orow + ow) * C + c;
                    if constexpr (MODE == UP2_STATS) {
#pragma unroll
                        for (int r = 0; r < V; ++r) { acc0[r] += u[r]; acc1[r] += u[r] * u[r]; }
                    } else if constexpr (MODE == UP2_APPLY) {
#pragma unroll
                        for (int r = 0; r < V; ++r) u[r] = fmaf(a[r], u[r], b[r]);
                        apply_act_vec<V>(u, act);
                        VecIO<T, V>::st(out + off, u);
                    } else {
                        float d[V];
                        VecIO<T, V>::ld(dy + off, d);
#pragma unroll
                        for (int r = 0; r < V; ++r) {
                            const float xh = (u[r] - mu[r]) * is[r];
                            float dz = d[r];
                            if (act == SP_ACT_LRELU) dz = (sc[r] * xh + bi[r]) > 0.f ? dz : 0.2f * dz;
                            if constexpr (MODE == UP2_BWD_REDUCE) { acc0[r] += dz; acc1[r] += dz * xh; }
                            else d[r] = is[r] * (dz * sc[r] - k1[r] - xh * k2[r]);
                        }
                        if constexpr (MODE == UP2_BWD_APPLY) VecIO<T, V>::st(out + off, d);
                    }
                }
            }
        }
    }
    if constexpr (MODE == UP2_STATS || MODE == UP2_BWD_REDUCE) {
#pragma unroll
        for (int r = 0; r < V; ++r) { red[threadIdx.x * 2 * V + r] = acc0[r]; red[threadIdx.x * 2 * V + V + r] = acc1[r]; }
        __syncthreads();
        if (live && L.pl == 0) {
            const long prow = ((long)n * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;     // both layouts: sample-major, then the sample's blocks
            for (int r = 0; r < V; ++r) {
                float ta = 0.f, tb = 0.f;
                for (int k = 0; k < L.pix_par; ++k) {
                    ta += red[(k * L.lanes_per_pix + L.cg) * 2 * V + r];
                    tb += red[(k * L.lanes_per_pix + L.cg) * 2 * V + V + r];
                }
                part[(prow * C + c + r) * 2] = ta;
                part[(prow * C + c + r) * 2 + 1] = tb;
            }
        }
    }
}

// grid of the up2 kernels: column slabs x row slabs x samples; reductions keep samples * gx * gy <= BN_MAX_PARTS partial rows
inline dim3 up2_grid(int n, int h, int w, int c, int v, bool reduction) {
    const int groups = c / v, lanes = groups < 256 ? groups : 256, pix_par = 256 / lanes;
    int gx = ((2 * w + 2) / 4 + 1 + pix_par - 1) / pix_par;             // column groups {0}, {1..4}, {5..8}, ...
    if (gx < 1) gx = 1;
    int gy = 2 * h;
    if (reduction) {
        int budget = BN_MAX_PARTS / n;
        if (budget < 1) budget = 1;
        if (gx > budget) gx = budget;
        gy = budget / gx;
        if (gy > 2 * h) gy = 2 * h;
        if (gy < 1) gy = 1;
    }
    return dim3((unsigned)gx, (unsigned)gy, (unsigned)n);
}

inline int ew_grid(long items) { long b = (items + 255) / 256; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

// elementwise passes: blocks per sample so that a thread walks ~iters pixels (SP_TUNE_BN_ITERS)
inline int apply_blocks(long hw, int c, int v, int n) {
    const int groups = c / v, lanes = groups < 256 ? groups : 256, pix_par = 256 / lanes;
    const int iters = sp_tune(SP_TUNE_BN_ITERS, 2) < 1 ? 1 : sp_tune(SP_TUNE_BN_ITERS, 2);      // 2 / 4 / 8 / 16 / 32 measured (scratch/bw_probe.py): 2 is at the rate of a device copy
    long bx = (hw + (long)pix_par * iters - 1) / ((long)pix_par * iters);
    if (bx * n > 4096) bx = 4096 / n > 0 ? 4096 / n : 1;
    if (bx < 1) bx = 1;
    return (int)bx;
}
inline int stat_blocks(long pixels, int c, int v) {
    const int groups = c / v, lanes = groups < 256 ? groups : 256, pix_par = 256 / lanes;
    long blocks = pixels / ((long)pix_par * 8);
    if (blocks > BN_MAX_PARTS) blocks = BN_MAX_PARTS;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

// vector width: 8 bf16 (16 bytes) when the channel count allows, else 4
#define SP_BN_DISPATCH(dtype, c, KERNEL, GRID, ...)                                                                      \
    do {                                                                                                                   \
        if ((dtype) == SP_F32) hipLaunchKernelGGL((KERNEL<float, 4>), GRID, dim3(256), 0, s, __VA_ARGS__);               \
        else if ((c) % 8 == 0) hipLaunchKernelGGL((KERNEL<bf16, 8>), GRID, dim3(256), 0, s, __VA_ARGS__);                \
        else hipLaunchKernelGGL((KERNEL<bf16, 4>), GRID, dim3(256), 0, s, __VA_ARGS__);                                  \
    } while (0)

extern "C" int sp_bn_stats(const void* x, int32_t n, int64_t hw, int32_t c, float* partials, float eps, float momentum,
                           float* running_mean, float* running_var, int32_t training, float* mean_out,
                           float* invstd_out, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && partials && mean_out && invstd_out && c % 4 == 0 && n > 0 && hw > 0, "sp_bn_stats: bad args (c=%d)", c);
    SP_CHECK_ARG(training || (running_mean && running_var), "sp_bn_stats: eval mode needs running statistics");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long pixels = (long)n * hw;
    int nparts = 0;
    if (training) {
        const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
        nparts = stat_blocks(pixels, c, v);
        if (dtype == SP_F32) hipLaunchKernelGGL((bn_stats_kernel<float, 4>), dim3(nparts), dim3(256), 0, s, (const float*)x, pixels, c, partials);
        else if (v == 8) hipLaunchKernelGGL((bn_stats_kernel<bf16, 8>), dim3(nparts), dim3(256), 0, s, (const bf16*)x, pixels, c, partials);
        else hipLaunchKernelGGL((bn_stats_kernel<bf16, 4>), dim3(nparts), dim3(256), 0, s, (const bf16*)x, pixels, c, partials);
        SP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(sp_div_up(c, FIN_CL)), dim3(256), 0, s, partials, nparts, pixels, c, eps, momentum,
                       running_mean, running_var, training, mean_out, invstd_out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_apply(const void* x, void* y, int32_t n, int64_t hw, int32_t c, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, const float* emb,
                           const int64_t* cls, int32_t act, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && y && mean && invstd && c % 4 == 0, "sp_bn_apply: bad args");
    SP_CHECK_ARG(!emb || cls, "sp_bn_apply: conditional mode needs class indices");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Affine aff{gamma, beta, emb, cls};
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    const dim3 g(apply_blocks(hw, c, v, n), n);
    if (dtype == SP_F32) hipLaunchKernelGGL((bn_apply_kernel<float, 4>), g, dim3(256), 0, s, (const float*)x, (float*)y, (long)hw, c, mean, invstd, aff, act);
    else if (v == 8) hipLaunchKernelGGL((bn_apply_kernel<bf16, 8>), g, dim3(256), 0, s, (const bf16*)x, (bf16*)y, (long)hw, c, mean, invstd, aff, act);
    else hipLaunchKernelGGL((bn_apply_kernel<bf16, 4>), g, dim3(256), 0, s, (const bf16*)x, (bf16*)y, (long)hw, c, mean, invstd, aff, act);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_stats_pair(const void* x, int32_t n, int32_t split, int64_t hw, int32_t c, float* partials, float eps, float momentum,
                                float* running_mean, float* running_var, int32_t first_group, float* mean2, float* invstd2,
                                int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && partials && mean2 && invstd2 && c % 4 == 0 && split > 0 && split < n && hw > 0 && (first_group == 0 || first_group == 1),
                 "sp_bn_stats_pair: bad args (c=%d, n=%d, split=%d)", c, n, split);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long pix_a = (long)split * hw, pix_b = (long)(n - split) * hw;
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    int nparts = stat_blocks(pix_a > pix_b ? pix_a : pix_b, c, v);
    if (nparts > BN_MAX_PARTS / 2) nparts = BN_MAX_PARTS / 2;         // both groups' partial rows share the caller's 1024 * 2 * c floats
    const long stride = (long)nparts * 2 * c;
    const dim3 grid(nparts, 2);
    if (dtype == SP_F32) hipLaunchKernelGGL((bn_stats_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)x, pix_a, c, partials, pix_b, stride);
    else if (v == 8) hipLaunchKernelGGL((bn_stats_kernel<bf16, 8>), grid, dim3(256), 0, s, (const bf16*)x, pix_a, c, partials, pix_b, stride);
    else hipLaunchKernelGGL((bn_stats_kernel<bf16, 4>), grid, dim3(256), 0, s, (const bf16*)x, pix_a, c, partials, pix_b, stride);
    SP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(sp_div_up(c, FIN_CL)), dim3(256), 0, s, partials, nparts, pix_a, c, eps, momentum,
                       running_mean, running_var, 1, mean2, invstd2, 2, pix_b, stride, first_group);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_apply_pair(const void* x, void* y, int32_t n, int32_t split, int64_t hw, int32_t c, const float* mean2,
                                const float* invstd2, const float* gamma, const float* beta, const float* emb, const int64_t* cls,
                                int32_t act, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && y && mean2 && invstd2 && c % 4 == 0 && split > 0 && split < n, "sp_bn_apply_pair: bad args");
    SP_CHECK_ARG(!emb || cls, "sp_bn_apply_pair: conditional mode needs class indices");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Affine aff{gamma, beta, emb, cls};
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    const dim3 g(apply_blocks(hw, c, v, n), n);
    if (dtype == SP_F32) hipLaunchKernelGGL((bn_apply_kernel<float, 4>), g, dim3(256), 0, s, (const float*)x, (float*)y, (long)hw, c, mean2, invstd2, aff, act, split);
    else if (v == 8) hipLaunchKernelGGL((bn_apply_kernel<bf16, 8>), g, dim3(256), 0, s, (const bf16*)x, (bf16*)y, (long)hw, c, mean2, invstd2, aff, act, split);
    else hipLaunchKernelGGL((bn_apply_kernel<bf16, 4>), g, dim3(256), 0, s, (const bf16*)x, (bf16*)y, (long)hw, c, mean2, invstd2, aff, act, split);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_apply_upsample2(const void* x, void* y, int32_t n, int32_t h, int32_t w_, int32_t c, const float* mean,
                                     const float* invstd, const float* gamma, const float* beta, const float* emb,
                                     const int64_t* cls, int32_t act, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && y && mean && invstd && c % 4 == 0 && n > 0 && h > 0 && w_ > 0, "sp_bn_apply_upsample2: bad args");
    SP_CHECK_ARG(!emb || cls, "sp_bn_apply_upsample2: conditional mode needs class indices");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Affine aff{gamma, beta, emb, cls};
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    const int groups = c / v, lanes = groups < 256 ? groups : 256, pix_par = 256 / lanes;
    int bx = ((2 * w_ + 2) / 4 + 1 + pix_par - 1) / pix_par;        // a thread: output columns 4q - 3 ... 4q
    if (bx < 1) bx = 1;
    const dim3 g(bx, (long)n * 2 * h < 65535 ? n * 2 * h : 65535);
    if (dtype == SP_F32) hipLaunchKernelGGL((bn_apply_upsample2_kernel<float, 4>), g, dim3(256), 0, s, (const float*)x, (float*)y, n, h, w_, c, mean, invstd, aff, act);
    else if (v == 8) hipLaunchKernelGGL((bn_apply_upsample2_kernel<bf16, 8>), g, dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w_, c, mean, invstd, aff, act);
    else hipLaunchKernelGGL((bn_apply_upsample2_kernel<bf16, 4>), g, dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w_, c, mean, invstd, aff, act);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_apply_upsample2_pair(const void* x, void* y, int32_t n, int32_t split, int32_t h, int32_t w_, int32_t c, const float* mean2,
                                          const float* invstd2, const float* gamma, const float* beta, const float* emb,
                                          const int64_t* cls, int32_t act, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && y && mean2 && invstd2 && c % 4 == 0 && n > 0 && h > 0 && w_ > 0 && split > 0 && split < n, "sp_bn_apply_upsample2_pair: bad args");
    SP_CHECK_ARG(!emb || cls, "sp_bn_apply_upsample2_pair: conditional mode needs class indices");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Affine aff{gamma, beta, emb, cls};
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    const int groups = c / v, lanes = groups < 256 ? groups : 256, pix_par = 256 / lanes;
    int bx = ((2 * w_ + 2) / 4 + 1 + pix_par - 1) / pix_par;
    if (bx < 1) bx = 1;
    const dim3 g(bx, (long)n * 2 * h < 65535 ? n * 2 * h : 65535);
    if (dtype == SP_F32) hipLaunchKernelGGL((bn_apply_upsample2_kernel<float, 4>), g, dim3(256), 0, s, (const float*)x, (float*)y, n, h, w_, c, mean2, invstd2, aff, act, split);
    else if (v == 8) hipLaunchKernelGGL((bn_apply_upsample2_kernel<bf16, 8>), g, dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w_, c, mean2, invstd2, aff, act, split);
    else hipLaunchKernelGGL((bn_apply_upsample2_kernel<bf16, 4>), g, dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w_, c, mean2, invstd2, aff, act, split);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_stats_up2(const void* x, int32_t n, int32_t h, int32_t w_, int32_t c, float* partials, float eps, float momentum,
                               float* running_mean, float* running_var, float* mean_out, float* invstd_out, int32_t dtype,
                               sp_stream_t stream) {
    SP_CHECK_ARG(x && partials && mean_out && invstd_out && c % 4 == 0 && n > 0 && n <= BN_MAX_PARTS && h > 0 && w_ > 0, "sp_bn_stats_up2: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    SP_CHECK_ARG(c / v <= 256, "sp_bn_stats_up2: at most %d channels", 256 * v);
    const dim3 g = up2_grid(n, h, w_, c, v, true);
    Affine none{nullptr, nullptr, nullptr, nullptr};
    if (dtype == SP_F32) hipLaunchKernelGGL((bn_up2_kernel<float, 4, UP2_STATS>), g, dim3(256), 0, s, (const float*)x, h, w_, c, (const float*)nullptr, (float*)nullptr, nullptr, nullptr, none, 0, nullptr, nullptr, partials);
    else if (v == 8) hipLaunchKernelGGL((bn_up2_kernel<bf16, 8, UP2_STATS>), g, dim3(256), 0, s, (const bf16*)x, h, w_, c, (const bf16*)nullptr, (bf16*)nullptr, nullptr, nullptr, none, 0, nullptr, nullptr, partials);
    else hipLaunchKernelGGL((bn_up2_kernel<bf16, 4, UP2_STATS>), g, dim3(256), 0, s, (const bf16*)x, h, w_, c, (const bf16*)nullptr, (bf16*)nullptr, nullptr, nullptr, none, 0, nullptr, nullptr, partials);
    SP_LAUNCH_CHECK();
    const long pixels = (long)n * 4 * h * w_;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(sp_div_up(c, FIN_CL)), dim3(256), 0, s, partials, (int)(g.x * g.y * g.z), pixels, c, eps, momentum,
                       running_mean, running_var, 1, mean_out, invstd_out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_apply_up2(const void* x, void* y, int32_t n, int32_t h, int32_t w_, int32_t c, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, const float* emb, const int64_t* cls, int32_t act, int32_t dtype,
                               sp_stream_t stream) {
    SP_CHECK_ARG(x && y && mean && invstd && c % 4 == 0 && n > 0 && n <= 65535, "sp_bn_apply_up2: bad args");
    SP_CHECK_ARG(!emb || cls, "sp_bn_apply_up2: conditional mode needs class indices");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    SP_CHECK_ARG(c / v <= 256, "sp_bn_apply_up2: at most %d channels", 256 * v);
    Affine aff{gamma, beta, emb, cls};
    const dim3 g = up2_grid(n, h, w_, c, v, false);
    if (dtype == SP_F32) hipLaunchKernelGGL((bn_up2_kernel<float, 4, UP2_APPLY>), g, dim3(256), 0, s, (const float*)x, h, w_, c, (const float*)nullptr, (float*)y, mean, invstd, aff, act, nullptr, nullptr, nullptr);
    else if (v == 8) hipLaunchKernelGGL((bn_up2_kernel<bf16, 8, UP2_APPLY>), g, dim3(256), 0, s, (const bf16*)x, h, w_, c, (const bf16*)nullptr, (bf16*)y, mean, invstd, aff, act, nullptr, nullptr, nullptr);
    else hipLaunchKernelGGL((bn_up2_kernel<bf16, 4, UP2_APPLY>), g, dim3(256), 0, s, (const bf16*)x, h, w_, c, (const bf16*)nullptr, (bf16*)y, mean, invstd, aff, act, nullptr, nullptr, nullptr);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_backward_up2(const void* dy, const void* x, void* du, int32_t n, int32_t h, int32_t w_, int32_t c, const float* mean,
                                  const float* invstd, const float* gamma, const float* beta, const float* emb, const int64_t* cls,
                                  int32_t act, float* partials, float* c_tmp, float* dgamma, float* dbeta, float* demb,
                                  int32_t num_classes, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(dy && x && du && mean && invstd && partials && c_tmp && c % 4 == 0 && n > 0 && n <= BN_MAX_PARTS, "sp_bn_backward_up2: bad args");
    SP_CHECK_ARG(!emb || (cls && (!demb || num_classes > 0)), "sp_bn_backward_up2: conditional mode needs class indices");
    SP_CHECK_ARG(!demb || n <= BN_EMB_MAXN, "sp_bn_backward_up2: the conditional backward supports batches up to %d", BN_EMB_MAXN);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    SP_CHECK_ARG(c / v <= 256, "sp_bn_backward_up2: at most %d channels", 256 * v);
    Affine aff{gamma, beta, emb, cls};
    const dim3 rg = up2_grid(n, h, w_, c, v, true);
    if (dtype == SP_F32) hipLaunchKernelGGL((bn_up2_kernel<float, 4, UP2_BWD_REDUCE>), rg, dim3(256), 0, s, (const float*)x, h, w_, c, (const float*)dy, (float*)nullptr, mean, invstd, aff, act, nullptr, nullptr, partials);
    else if (v == 8) hipLaunchKernelGGL((bn_up2_kernel<bf16, 8, UP2_BWD_REDUCE>), rg, dim3(256), 0, s, (const bf16*)x, h, w_, c, (const bf16*)dy, (bf16*)nullptr, mean, invstd, aff, act, nullptr, nullptr, partials);
    else hipLaunchKernelGGL((bn_up2_kernel<bf16, 4, UP2_BWD_REDUCE>), rg, dim3(256), 0, s, (const bf16*)x, h, w_, c, (const bf16*)dy, (bf16*)nullptr, mean, invstd, aff, act, nullptr, nullptr, partials);
    SP_LAUNCH_CHECK();
    const long pixels = (long)n * 4 * h * w_;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(sp_div_up(c, FIN_CL)), dim3(256), 0, s, partials, (int)(rg.x * rg.y), n, c, pixels, aff, c_tmp, c_tmp + c,
                       dgamma, dbeta, demb, num_classes);
    SP_LAUNCH_CHECK();
    const dim3 g = up2_grid(n, h, w_, c, v, false);
    if (dtype == SP_F32) hipLaunchKernelGGL((bn_up2_kernel<float, 4, UP2_BWD_APPLY>), g, dim3(256), 0, s, (const float*)x, h, w_, c, (const float*)dy, (float*)du, mean, invstd, aff, act, c_tmp, c_tmp + c, nullptr);
    else if (v == 8) hipLaunchKernelGGL((bn_up2_kernel<bf16, 8, UP2_BWD_APPLY>), g, dim3(256), 0, s, (const bf16*)x, h, w_, c, (const bf16*)dy, (bf16*)du, mean, invstd, aff, act, c_tmp, c_tmp + c, nullptr);
    else hipLaunchKernelGGL((bn_up2_kernel<bf16, 4, UP2_BWD_APPLY>), g, dim3(256), 0, s, (const bf16*)x, h, w_, c, (const bf16*)dy, (bf16*)du, mean, invstd, aff, act, c_tmp, c_tmp + c, nullptr);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_backward(const void* dy, const void* x, void* dx, int32_t n, int64_t hw, int32_t c,
                              const float* mean, const float* invstd, const float* gamma, const float* beta,
                              const float* emb, const int64_t* cls, int32_t act, float* partials, float* c_tmp,
                              float* dgamma, float* dbeta, float* demb, int32_t num_classes, int32_t dtype,
                              sp_stream_t stream) {
    SP_CHECK_ARG(dy && x && dx && mean && invstd && partials && c_tmp && c % 4 == 0, "sp_bn_backward: bad args");
    SP_CHECK_ARG(!emb || (cls && (!demb || num_classes > 0)), "sp_bn_backward: conditional mode needs class indices");
    SP_CHECK_ARG(!demb || n <= BN_EMB_MAXN, "sp_bn_backward: the conditional backward supports batches up to %d", BN_EMB_MAXN);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Affine aff{gamma, beta, emb, cls};
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    int nparts = stat_blocks(hw, c, v);
    if ((long)nparts * n > BN_MAX_PARTS) nparts = BN_MAX_PARTS / n > 0 ? BN_MAX_PARTS / n : 1;
    const dim3 rgrid(nparts, n);
    if (dtype == SP_F32) hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, 4>), rgrid, dim3(256), 0, s, (const float*)dy, (const float*)x, (long)hw, c, mean, invstd, aff, act, partials);
    else if (v == 8) hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16, 8>), rgrid, dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (long)hw, c, mean, invstd, aff, act, partials);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16, 4>), rgrid, dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (long)hw, c, mean, invstd, aff, act, partials);
    SP_LAUNCH_CHECK();
    const long pixels = (long)n * hw;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(sp_div_up(c, FIN_CL)), dim3(256), 0, s, partials, nparts, n, c, pixels, aff, c_tmp, c_tmp + c,
                       dgamma, dbeta, demb, num_classes);
    SP_LAUNCH_CHECK();
    const dim3 g(apply_blocks(hw, c, v, n), n);
    if (dtype == SP_F32) hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 4>), g, dim3(256), 0, s, (const float*)dy, (const float*)x, (float*)dx, (long)hw, c, mean, invstd, aff, act, c_tmp, c_tmp + c);
    else if (v == 8) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, 8>), g, dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (bf16*)dx, (long)hw, c, mean, invstd, aff, act, c_tmp, c_tmp + c);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, 4>), g, dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (bf16*)dx, (long)hw, c, mean, invstd, aff, act, c_tmp, c_tmp + c);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
