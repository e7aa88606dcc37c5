// Pooling / upsampling on NHWC tensors, forward and backward as GATHERS (deterministic, no atomics).
//   avg-pool 2x2      nn.AvgPool2d((2,2))             models.py:406,451
//   max-pool 2x2/2    nn.MaxPool2d(2,2)               models.py:245; VGG features (models.py:203)
//   adaptive avg-pool nn.AdaptiveAvgPool2d            models.py:126 (-> 1x1), VGG avgpool 8->7 (models.py:206)
//   bilinear x2       nn.UpsamplingBilinear2d(2)      models.py:52,298,308 (align_corners=True)
// One lane handles 4 channels of one output element (8/16-byte accesses).
#include "common.h"

namespace {

// (0: more items than the kernels' 32-bit indexing covers - the entry points reject that)
inline int ew_grid(long items) { if (items >= (1L << 31)) return 0; long b = (items + 255) / 256; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }

// 32-bit item index (the first version walked and decoded a 64-bit index: four 64-bit divisions per item - the resampling
// kernels ran at 2.0 - 2.7 TB/s); two items per thread in flight
#define SP_FOR_VEC(total) _Pragma("unroll 2") for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)(total); i += gridDim.x * 256u)

// decode i -> (n, oh, ow, c4) for an output of OH x OW x C
#define SP_DECODE(i, OH, OW, C)                              \
    const unsigned vpp = (unsigned)((C) / V);                 \
    const unsigned pp_ = (i) / vpp;                           \
    const int c = (int)((i) - pp_ * vpp) * V;                 \
    const unsigned q_ = pp_ / (unsigned)(OW);                 \
    const int ow = (int)(pp_ - q_ * (unsigned)(OW));          \
    const int n = (int)(q_ / (unsigned)(OH));                 \
    const int oh = (int)(q_ - (unsigned)n * (unsigned)(OH));

template <typename T, int V>
__global__ void avgpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int act2,
                                    T* __restrict__ y2) {
    const int OH = H / 2, OW = W / 2;
    const long total = (long)N * OH * OW * (C / V);
    SP_FOR_VEC(total) {
        SP_DECODE(i, OH, OW, C)
        const T* b = x + (((long)n * H + oh * 2) * W + ow * 2) * C + c;
        float a[V], t[V];
        VecIO<T, V>::ld(b, a);
        VecIO<T, V>::ld(b + C, t); for (int r = 0; r < V; ++r) a[r] += t[r];
        VecIO<T, V>::ld(b + (long)W * C, t); for (int r = 0; r < V; ++r) a[r] += t[r];
        VecIO<T, V>::ld(b + (long)W * C + C, t); for (int r = 0; r < V; ++r) a[r] += t[r];
        for (int r = 0; r < V; ++r) a[r] *= 0.25f;
        const long o = (((long)n * OH + oh) * OW + ow) * C + c;
        VecIO<T, V>::st(y + o, a);
        if (y2) { apply_act_vec<V>(a, act2); VecIO<T, V>::st(y2 + o, a); }
    }
}

template <typename T, int V>
__global__ void avgpool2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C) {
    const int OH = H / 2, OW = W / 2;
    const long total = (long)N * H * W * (C / V);
    SP_FOR_VEC(total) {
        SP_DECODE(i, H, W, C)
        float a[V];
        for (int r = 0; r < V; ++r) a[r] = 0.f;
        if (oh / 2 < OH && ow / 2 < OW) {
            VecIO<T, V>::ld(dy + (((long)n * OH + oh / 2) * OW + ow / 2) * C + c, a);
            for (int r = 0; r < V; ++r) a[r] *= 0.25f;
        }
        VecIO<T, V>::st(dx + (((long)n * H + oh) * W + ow) * C + c, a);
    }
}

// LeakyReLU / ReLU and 2x2 average pooling of the SAME tensor in one pass (the input of a discriminator block feeds
// LeakyReLU -> conv and AvgPool -> 1x1 conv, models.py:452-462 after the commutation in models.py of this package): x is read
// once; the backward pass forms dx = act'(x) * d_act + expand(d_pool) / 4 in one kernel instead of an activation backward,
// a pooling backward and the autograd sum of the two branches.
template <typename T, int V>
__global__ void act_avgpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ ya, T* __restrict__ yp, int N, int H, int W, int C,
                                        int act) {
    const int OH = H / 2, OW = W / 2;
    const long total = (long)N * OH * OW * (C / V);
    SP_FOR_VEC(total) {
        SP_DECODE(i, OH, OW, C)
        const long base = (((long)n * H + oh * 2) * W + ow * 2) * C + c;
        const long offs[4] = {0, C, (long)W * C, (long)W * C + C};
        float a[V], t[V];
        for (int r = 0; r < V; ++r) a[r] = 0.f;
        for (int k = 0; k < 4; ++k) {
            VecIO<T, V>::ld(x + base + offs[k], t);
            for (int r = 0; r < V; ++r) a[r] += t[r];
            apply_act_vec<V>(t, act);
            VecIO<T, V>::st(ya + base + offs[k], t);
        }
        for (int r = 0; r < V; ++r) a[r] *= 0.25f;
        VecIO<T, V>::st(yp + (((long)n * OH + oh) * OW + ow) * C + c, a);
    }
}

template <typename T, int V>
__global__ void act_avgpool2_bwd_kernel(const T* __restrict__ ga, const T* __restrict__ gp, const T* __restrict__ x, T* __restrict__ dx,
                                        int N, int H, int W, int C, int act) {
    const int OH = H / 2, OW = W / 2;
    const long total = (long)N * OH * OW * (C / V);
    const float slope = act == SP_ACT_LRELU ? 0.2f : 0.f;
    SP_FOR_VEC(total) {
        SP_DECODE(i, OH, OW, C)
        const long base = (((long)n * H + oh * 2) * W + ow * 2) * C + c;
        const long offs[4] = {0, C, (long)W * C, (long)W * C + C};
        // all nine loads of an item are requested before the first is used (round 5: one window position at a time - load, mask,
        // store - made four dependent round trips per item)
        float p[V], g[4][V], t[4][V];
        for (int r = 0; r < V; ++r) p[r] = 0.f;
        if (gp) VecIO<T, V>::ld(gp + (((long)n * OH + oh) * OW + ow) * C + c, p);
        const bool masked = ga != nullptr && act != SP_ACT_NONE;
        if (ga) {
#pragma unroll
            for (int k = 0; k < 4; ++k) VecIO<T, V>::ld(ga + base + offs[k], g[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                for (int r = 0; r < V; ++r) g[k][r] = 0.f;
        }
        if (masked) {
#pragma unroll
            for (int k = 0; k < 4; ++k) VecIO<T, V>::ld(x + base + offs[k], t[k]);
        }
        for (int r = 0; r < V; ++r) p[r] *= 0.25f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (masked)
                for (int r = 0; r < V; ++r) g[k][r] *= (t[k][r] > 0.f ? 1.f : slope);
            for (int r = 0; r < V; ++r) g[k][r] += p[r];
            VecIO<T, V>::st(dx + base + offs[k], g[k]);
        }
    }
}

// max-pool; `relu` applies max(.,0) to the pooled value (VGG taps are post-ReLU; relu and max commute)
template <typename T, int V>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int relu) {
    const int OH = H / 2, OW = W / 2;
    const long total = (long)N * OH * OW * (C / V);
    SP_FOR_VEC(total) {
        SP_DECODE(i, OH, OW, C)
        const T* b = x + (((long)n * H + oh * 2) * W + ow * 2) * C + c;
        float a[V], t[V];
        VecIO<T, V>::ld(b, a);
        VecIO<T, V>::ld(b + C, t); for (int r = 0; r < V; ++r) a[r] = fmaxf(a[r], t[r]);
        VecIO<T, V>::ld(b + (long)W * C, t); for (int r = 0; r < V; ++r) a[r] = fmaxf(a[r], t[r]);
        VecIO<T, V>::ld(b + (long)W * C + C, t); for (int r = 0; r < V; ++r) a[r] = fmaxf(a[r], t[r]);
        if (relu) for (int r = 0; r < V; ++r) a[r] = fmaxf(a[r], 0.f);
        VecIO<T, V>::st(y + (((long)n * OH + oh) * OW + ow) * C + c, a);
    }
}

// dx[h,w] = dy[h/2,w/2] if (h,w) is the FIRST maximum of its window in scan order (torch picks the first
// element that compares greater), else 0.  With `relu`, additionally zero where the pooled max <= 0.
template <typename T, int V>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx, int N, int H, int W,
                                    int C, int relu) {
    const int OH = H / 2, OW = W / 2;
    const long total = (long)N * OH * OW * (C / V);
    SP_FOR_VEC(total) {
        SP_DECODE(i, OH, OW, C)
        const long base = (((long)n * H + oh * 2) * W + ow * 2) * C + c;
        const long offs[4] = {0, C, (long)W * C, (long)W * C + C};
        float v[4][V], g[V];
        for (int k = 0; k < 4; ++k) VecIO<T, V>::ld(x + base + offs[k], v[k]);
        VecIO<T, V>::ld(dy + (((long)n * OH + oh) * OW + ow) * C + c, g);
        float o[4][V];
        for (int r = 0; r < V; ++r) {
            int best = 0;
            float m = v[0][r];
            for (int k = 1; k < 4; ++k) if (v[k][r] > m) { m = v[k][r]; best = k; }
            const float gv = (relu && !(m > 0.f)) ? 0.f : g[r];
            for (int k = 0; k < 4; ++k) o[k][r] = (k == best) ? gv : 0.f;
        }
        for (int k = 0; k < 4; ++k) VecIO<T, V>::st(dx + base + offs[k], o[k]);
    }
}

// The same gradient routing from what the convolution's fused ReLU + MaxPool epilogue recorded (sp_conv_params.pool_idx: 2 bits per
// pooled element = 2 * row + column of the first maximum): the unpooled tensor is not read - it was never written.
template <typename T, int V>
__global__ void maxpool2_bwd_idx_kernel(const T* __restrict__ dy, const T* __restrict__ y, const uint32_t* __restrict__ idx,
                                        T* __restrict__ dx, int N, int H, int W, int C) {
    const int OH = H / 2, OW = W / 2;
    const long total = (long)N * OH * OW * (C / V);
    SP_FOR_VEC(total) {
        SP_DECODE(i, OH, OW, C)
        const long ppix = ((long)n * OH + oh) * OW + ow;
        const long base = (((long)n * H + oh * 2) * W + ow * 2) * C + c;
        const long offs[4] = {0, C, (long)W * C, (long)W * C + C};
        float g[V], m[V];
        VecIO<T, V>::ld(dy + ppix * C + c, g);
        VecIO<T, V>::ld(y + ppix * C + c, m);
        const uint32_t word = idx[ppix * (C >> 4) + (c >> 4)] >> (2 * (c & 15));
        float o[4][V];
        for (int r = 0; r < V; ++r) {
            const int best = (int)((word >> (2 * r)) & 3u);
            const float gv = m[r] > 0.f ? g[r] : 0.f;
            for (int k = 0; k < 4; ++k) o[k][r] = (k == best) ? gv : 0.f;
        }
        for (int k = 0; k < 4; ++k) VecIO<T, V>::st(dx + base + offs[k], o[k]);
    }
}

__device__ __forceinline__ int ad_start(int o, int in, int out) { return (int)(((long)o * in) / out); }
__device__ __forceinline__ int ad_end(int o, int in, int out) { return (int)((((long)o + 1) * in + out - 1) / out); }

template <typename T>
__global__ void adaptive_avg_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int OH, int OW,
                                        int act_in) {
    constexpr int V = 4;
    const long total = (long)N * OH * OW * (C / 4);
    SP_FOR_VEC(total) {
        SP_DECODE(i, OH, OW, C)
        const int h0 = ad_start(oh, H, OH), h1 = ad_end(oh, H, OH), w0 = ad_start(ow, W, OW), w1 = ad_end(ow, W, OW);
        float a[4] = {0.f, 0.f, 0.f, 0.f}, t[4];
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) {
                Elem<T>::ld4(x + (((long)n * H + h) * W + w) * C + c, t);
                apply_act_vec<4>(t, act_in);
                for (int r = 0; r < 4; ++r) a[r] += t[r];
            }
        const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
        for (int r = 0; r < 4; ++r) a[r] *= inv;
        Elem<T>::st4(y + (((long)n * OH + oh) * OW + ow) * C + c, a);
    }
}

// gather over the (few) output windows that contain input (h,w); optional act'(x) of a fused input activation
template <typename T>
__global__ void adaptive_avg_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx, int N, int H,
                                        int W, int C, int OH, int OW, int act_in) {
    constexpr int V = 4;
    const long total = (long)N * H * W * (C / 4);
    SP_FOR_VEC(total) {
        SP_DECODE(i, H, W, C)      // here (oh, ow) index the INPUT pixel
        float a[4] = {0.f, 0.f, 0.f, 0.f}, t[4];
        for (int o = 0; o < OH; ++o) {
            const int h0 = ad_start(o, H, OH), h1 = ad_end(o, H, OH);
            if (oh < h0 || oh >= h1) continue;
            for (int p = 0; p < OW; ++p) {
                const int w0 = ad_start(p, W, OW), w1 = ad_end(p, W, OW);
                if (ow < w0 || ow >= w1) continue;
                Elem<T>::ld4(dy + (((long)n * OH + o) * OW + p) * C + c, t);
                const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
                for (int r = 0; r < 4; ++r) a[r] += t[r] * inv;
            }
        }
        const long off = (((long)n * H + oh) * W + ow) * C + c;
        if (act_in != SP_ACT_NONE) {
            Elem<T>::ld4(x + off, t);
            for (int r = 0; r < 4; ++r) {
                if (act_in == SP_ACT_LRELU) a[r] = t[r] > 0.f ? a[r] : 0.2f * a[r];
                else if (act_in == SP_ACT_RELU) a[r] = t[r] > 0.f ? a[r] : 0.f;
            }
        }
        Elem<T>::st4(dx + off, a);
    }
}

// bilinear x2, align_corners=True: src = dst * (in-1)/(out-1) (float), taps floor/floor+1.
// grid (column slabs, N * OH): a block owns (part of) one output row - sample, source rows and row weight are block-uniform, a
// thread keeps one channel group and walks columns (no per-vector index decoding: three integer divisions per 16 bytes held the
// flat-index form at 3.1 TB/s)
template <typename T, int V>
__global__ __launch_bounds__(256) void upsample2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
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
    const int lanes_per_pix = ngroups < 256 ? ngroups : 256, pix_par = 256 / lanes_per_pix;
    const int cg = threadIdx.x % lanes_per_pix, pl = threadIdx.x / lanes_per_pix;
    for (int cbase = 0; cbase < ngroups; cbase += lanes_per_pix) {
        const int c = (cbase + cg) * V;
        if (c >= C || pl >= pix_par) continue;
        const T* r0 = x + ((long)n * H + h0) * W * C + c;
        const T* r1 = x + ((long)n * H + h1) * W * C + c;
        T* yr = y + ((long)n * OH + oh) * OW * C + c;
        // output columns 2k + 1 and 2k + 2 both interpolate between source columns k and k + 1 (align_corners: column ow sits at
        // ow (W - 1) / (2W - 1)): a thread takes the four outputs 4q - 3 ... 4q from the THREE source columns 2q - 2, 2q - 1, 2q with a
        // fixed pattern - 6 loads per 4 vectors instead of 16, no index arithmetic per output (round 4; rows first, then columns)
        for (int q = blockIdx.x * pix_par + pl; 4 * q - 3 < OW; q += gridDim.x * pix_par) {
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
                const int w0 = 2 * q - 2 + (j >> 1);                           // = floor(ow (W - 1) / (2W - 1)) for ow >= 1
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

// dx[h,w] = sum over the output pixels whose taps include (h,w), with the forward's weights.  With align_corners output row o >= 1
// interpolates between input rows (o - 1) / 2 and (o - 1) / 2 + 1 (row 0 is input row 0), so input row h hears from exactly the
// four output rows 2h - 1 ... 2h + 2 - likewise for columns.  grid (column slabs, N * H): a block owns (part of) one INPUT row,
// whose four row weights are block-uniform; a thread keeps one channel group and produces TWO adjacent input columns from the six
// output columns 2k - 1 ... 2k + 4 (column sums over the four rows first: 24 loads per two results).  Round 4: the first form walked
// a bounding range of candidate rows / columns per pixel and tested each one's taps (~49 candidates with floor / ceil / divisions for
// 16 hits: 2.3 TB/s on the 256 x 256 layer).
template <typename T, int V>
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C) {
    const int OH = 2 * H, OW = 2 * W;
    const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const int ngroups = C / V;
    const int lanes_per_pix = ngroups < 256 ? ngroups : 256, pix_par = 256 / lanes_per_pix;
    const int cg = threadIdx.x % lanes_per_pix, pl = threadIdx.x / lanes_per_pix;
    // weight of output index o (extent O = 2 I, scale s) on input index i
    auto tap_weight = [](int o, int O, int I, float s, int i) {
        if (o < 0 || o >= O) return 0.f;
        const int i0 = o == 0 ? 0 : (o - 1) >> 1;
        const float l = o == 0 ? 0.f : s * o - (float)i0;
        const int i1 = i0 + 1 < I ? i0 + 1 : I - 1;
        return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
    };
    for (int row = blockIdx.y; row < N * H; row += gridDim.y) {
        const int n = row / H, h = row - n * H;
        float wh[4];
        long roff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = 2 * h - 1 + i;
            wh[i] = tap_weight(o, OH, H, sh, h);
            roff[i] = (long)(o < 0 ? 0 : (o < OH ? o : OH - 1)) * OW * C;
        }
        for (int cbase = 0; cbase < ngroups; cbase += lanes_per_pix) {
            const int c = (cbase + cg) * V;
            if (c >= C || pl >= pix_par) continue;
            const T* dyn = dy + (long)n * OH * OW * C + c;
            T* dxr = dx + ((long)n * H + h) * W * C + c;
            for (int kp = blockIdx.x * pix_par + pl; 2 * kp < W; kp += gridDim.x * pix_par) {
                const int k = 2 * kp;
                float vs[6][V];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int pcol = 2 * k - 1 + j;
                    const long coff = (long)(pcol < 0 ? 0 : (pcol < OW ? pcol : OW - 1)) * C;
#pragma unroll
                    for (int r = 0; r < V; ++r) vs[j][r] = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t[V];
                        VecIO<T, V>::ld(dyn + roff[i] + coff, t);
#pragma unroll
                        for (int r = 0; r < V; ++r) vs[j][r] += wh[i] * t[r];
                    }
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int kk = k + m;
                    if (kk >= W) break;
                    float a[V];
#pragma unroll
                    for (int r = 0; r < V; ++r) a[r] = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float ww = tap_weight(2 * kk - 1 + j, OW, W, sw, kk);
#pragma unroll
                        for (int r = 0; r < V; ++r) a[r] += ww * vs[2 * m + j][r];
                    }
                    VecIO<T, V>::st(dxr + (long)kk * C, a);
                }
            }
        }
    }
}

}  // namespace

#define SP_POOL_ARGS_OK(x, y, c) ((x) && (y) && (c) % 4 == 0 && (c) > 0)
#define SP_DT2(dtype, K, grid, ...) do { if ((dtype) == SP_F32) hipLaunchKernelGGL(K<float>, dim3(grid), dim3(256), 0, s, __VA_ARGS__); \
    else hipLaunchKernelGGL(K<bf16>, dim3(grid), dim3(256), 0, s, __VA_ARGS__); } while (0)

extern "C" int sp_avgpool2_fwd(const void* x, void* y, void* y_act, int32_t act, int32_t n, int32_t h, int32_t w_,
                               int32_t c, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(SP_POOL_ARGS_OK(x, y, c) && h % 2 == 0 && w_ % 2 == 0, "sp_avgpool2_fwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // bf16 tensors with C % 8 == 0 move 16 bytes per lane, everything else 4 elements
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    const int g = ew_grid((long)n * (h / 2) * (w_ / 2) * c / v);
    SP_CHECK_ARG(g > 0, "tensor too large for the 32-bit item index of the resampling kernels");
    if (dtype == SP_F32) hipLaunchKernelGGL((avgpool2_fwd_kernel<float, 4>), dim3(g), dim3(256), 0, s, (const float*)x, (float*)y, n, h, w_, c, act, (float*)y_act);
    else if (v == 8) hipLaunchKernelGGL((avgpool2_fwd_kernel<bf16, 8>), dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w_, c, act, (bf16*)y_act);
    else hipLaunchKernelGGL((avgpool2_fwd_kernel<bf16, 4>), dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w_, c, act, (bf16*)y_act);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_avgpool2_bwd(const void* dy, void* dx, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t dtype,
                               sp_stream_t stream) {
    SP_CHECK_ARG(SP_POOL_ARGS_OK(dy, dx, c) && h % 2 == 0 && w_ % 2 == 0, "sp_avgpool2_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // bf16 tensors with C % 8 == 0 move 16 bytes per lane, everything else 4 elements
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    const int g = ew_grid((long)n * h * w_ * c / v);
    SP_CHECK_ARG(g > 0, "tensor too large for the 32-bit item index of the resampling kernels");
    if (dtype == SP_F32) hipLaunchKernelGGL((avgpool2_bwd_kernel<float, 4>), dim3(g), dim3(256), 0, s, (const float*)dy, (float*)dx, n, h, w_, c);
    else if (v == 8) hipLaunchKernelGGL((avgpool2_bwd_kernel<bf16, 8>), dim3(g), dim3(256), 0, s, (const bf16*)dy, (bf16*)dx, n, h, w_, c);
    else hipLaunchKernelGGL((avgpool2_bwd_kernel<bf16, 4>), dim3(g), dim3(256), 0, s, (const bf16*)dy, (bf16*)dx, n, h, w_, c);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_act_avgpool2_fwd(const void* x, void* y_act, void* y_pool, int32_t act, int32_t n, int32_t h, int32_t w_,
                                   int32_t c, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(SP_POOL_ARGS_OK(x, y_act, c) && y_pool && h % 2 == 0 && w_ % 2 == 0, "sp_act_avgpool2_fwd: bad args");
    SP_CHECK_ARG(act == SP_ACT_NONE || act == SP_ACT_LRELU || act == SP_ACT_RELU, "sp_act_avgpool2_fwd: act %d unsupported", act);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    const int g = ew_grid((long)n * (h / 2) * (w_ / 2) * c / v);
    SP_CHECK_ARG(g > 0, "tensor too large for the 32-bit item index of the resampling kernels");
    if (dtype == SP_F32) hipLaunchKernelGGL((act_avgpool2_fwd_kernel<float, 4>), dim3(g), dim3(256), 0, s, (const float*)x, (float*)y_act, (float*)y_pool, n, h, w_, c, act);
    else if (v == 8) hipLaunchKernelGGL((act_avgpool2_fwd_kernel<bf16, 8>), dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)y_act, (bf16*)y_pool, n, h, w_, c, act);
    else hipLaunchKernelGGL((act_avgpool2_fwd_kernel<bf16, 4>), dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)y_act, (bf16*)y_pool, n, h, w_, c, act);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_act_avgpool2_bwd(const void* d_act, const void* d_pool, const void* x, void* dx, int32_t act, int32_t n, int32_t h,
                                   int32_t w_, int32_t c, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(SP_POOL_ARGS_OK(x, dx, c) && (d_act || d_pool) && h % 2 == 0 && w_ % 2 == 0, "sp_act_avgpool2_bwd: bad args");
    SP_CHECK_ARG(act == SP_ACT_NONE || act == SP_ACT_LRELU || act == SP_ACT_RELU, "sp_act_avgpool2_bwd: act %d unsupported", act);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    const int g = ew_grid((long)n * (h / 2) * (w_ / 2) * c / v);
    SP_CHECK_ARG(g > 0, "tensor too large for the 32-bit item index of the resampling kernels");
    if (dtype == SP_F32) hipLaunchKernelGGL((act_avgpool2_bwd_kernel<float, 4>), dim3(g), dim3(256), 0, s, (const float*)d_act, (const float*)d_pool, (const float*)x, (float*)dx, n, h, w_, c, act);
    else if (v == 8) hipLaunchKernelGGL((act_avgpool2_bwd_kernel<bf16, 8>), dim3(g), dim3(256), 0, s, (const bf16*)d_act, (const bf16*)d_pool, (const bf16*)x, (bf16*)dx, n, h, w_, c, act);
    else hipLaunchKernelGGL((act_avgpool2_bwd_kernel<bf16, 4>), dim3(g), dim3(256), 0, s, (const bf16*)d_act, (const bf16*)d_pool, (const bf16*)x, (bf16*)dx, n, h, w_, c, act);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_maxpool2_fwd(const void* x, void* y, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t relu,
                               int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(SP_POOL_ARGS_OK(x, y, c) && h % 2 == 0 && w_ % 2 == 0, "sp_maxpool2_fwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // bf16 tensors with C % 8 == 0 move 16 bytes per lane, everything else 4 elements
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    const int g = ew_grid((long)n * (h / 2) * (w_ / 2) * c / v);
    SP_CHECK_ARG(g > 0, "tensor too large for the 32-bit item index of the resampling kernels");
    if (dtype == SP_F32) hipLaunchKernelGGL((maxpool2_fwd_kernel<float, 4>), dim3(g), dim3(256), 0, s, (const float*)x, (float*)y, n, h, w_, c, relu);
    else if (v == 8) hipLaunchKernelGGL((maxpool2_fwd_kernel<bf16, 8>), dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w_, c, relu);
    else hipLaunchKernelGGL((maxpool2_fwd_kernel<bf16, 4>), dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w_, c, relu);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_maxpool2_bwd(const void* dy, const void* x, void* dx, int32_t n, int32_t h, int32_t w_, int32_t c,
                               int32_t relu, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(SP_POOL_ARGS_OK(dy, dx, c) && x && h % 2 == 0 && w_ % 2 == 0, "sp_maxpool2_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // bf16 tensors with C % 8 == 0 move 16 bytes per lane, everything else 4 elements
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    const int g = ew_grid((long)n * (h / 2) * (w_ / 2) * c / v);
    SP_CHECK_ARG(g > 0, "tensor too large for the 32-bit item index of the resampling kernels");
    if (dtype == SP_F32) hipLaunchKernelGGL((maxpool2_bwd_kernel<float, 4>), dim3(g), dim3(256), 0, s, (const float*)dy, (const float*)x, (float*)dx, n, h, w_, c, relu);
    else if (v == 8) hipLaunchKernelGGL((maxpool2_bwd_kernel<bf16, 8>), dim3(g), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (bf16*)dx, n, h, w_, c, relu);
    else hipLaunchKernelGGL((maxpool2_bwd_kernel<bf16, 4>), dim3(g), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (bf16*)dx, n, h, w_, c, relu);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_maxpool2_bwd_idx(const void* dy, const void* y, const uint32_t* idx, void* dx, int32_t n, int32_t h, int32_t w_, int32_t c,
                                   int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(SP_POOL_ARGS_OK(dy, dx, c) && y && idx && h % 2 == 0 && w_ % 2 == 0 && c % 16 == 0, "sp_maxpool2_bwd_idx: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int v = dtype == SP_BF16 ? 8 : 4;
    const int g = ew_grid((long)n * (h / 2) * (w_ / 2) * c / v);
    SP_CHECK_ARG(g > 0, "tensor too large for the 32-bit item index of the resampling kernels");
    if (dtype == SP_F32) hipLaunchKernelGGL((maxpool2_bwd_idx_kernel<float, 4>), dim3(g), dim3(256), 0, s, (const float*)dy, (const float*)y, idx, (float*)dx, n, h, w_, c);
    else hipLaunchKernelGGL((maxpool2_bwd_idx_kernel<bf16, 8>), dim3(g), dim3(256), 0, s, (const bf16*)dy, (const bf16*)y, idx, (bf16*)dx, n, h, w_, c);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_adaptive_avgpool_fwd(const void* x, void* y, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t oh,
                                       int32_t ow, int32_t act_in, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(SP_POOL_ARGS_OK(x, y, c) && oh > 0 && ow > 0, "sp_adaptive_avgpool_fwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int g = ew_grid((long)n * oh * ow * (c / 4));
    SP_CHECK_ARG(g > 0, "tensor too large for the 32-bit item index of the resampling kernels");
    if (dtype == SP_F32) hipLaunchKernelGGL(adaptive_avg_fwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, (float*)y, n, h, w_, c, oh, ow, act_in);
    else hipLaunchKernelGGL(adaptive_avg_fwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w_, c, oh, ow, act_in);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_adaptive_avgpool_bwd(const void* dy, const void* x, void* dx, int32_t n, int32_t h, int32_t w_,
                                       int32_t c, int32_t oh, int32_t ow, int32_t act_in, int32_t dtype,
                                       sp_stream_t stream) {
    SP_CHECK_ARG(SP_POOL_ARGS_OK(dy, dx, c) && oh > 0 && ow > 0 && (act_in == SP_ACT_NONE || x), "sp_adaptive_avgpool_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int g = ew_grid((long)n * h * w_ * (c / 4));
    SP_CHECK_ARG(g > 0, "tensor too large for the 32-bit item index of the resampling kernels");
    if (dtype == SP_F32) hipLaunchKernelGGL(adaptive_avg_bwd_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)dy, (const float*)x, (float*)dx, n, h, w_, c, oh, ow, act_in);
    else hipLaunchKernelGGL(adaptive_avg_bwd_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (bf16*)dx, n, h, w_, c, oh, ow, act_in);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_upsample2_fwd(const void* x, void* y, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t dtype,
                                sp_stream_t stream) {
    SP_CHECK_ARG(SP_POOL_ARGS_OK(x, y, c), "sp_upsample2_fwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // bf16 tensors with C % 8 == 0 move 16 bytes per lane, everything else 4 elements
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    const int groups = c / v, lanes = groups < 256 ? groups : 256, pix_par = 256 / lanes;
    int bx = ((2 * w_ + 2) / 4 + 1 + pix_par - 1) / pix_par;        // a thread: output columns 4q - 3 ... 4q
    if (bx < 1) bx = 1;
    const dim3 g(bx, (long)n * 2 * h < 65535 ? n * 2 * h : 65535);
    if (dtype == SP_F32) hipLaunchKernelGGL((upsample2_fwd_kernel<float, 4>), g, dim3(256), 0, s, (const float*)x, (float*)y, n, h, w_, c);
    else if (v == 8) hipLaunchKernelGGL((upsample2_fwd_kernel<bf16, 8>), g, dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w_, c);
    else hipLaunchKernelGGL((upsample2_fwd_kernel<bf16, 4>), g, dim3(256), 0, s, (const bf16*)x, (bf16*)y, n, h, w_, c);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_upsample2_bwd(const void* dy, void* dx, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t dtype,
                                sp_stream_t stream) {
    SP_CHECK_ARG(SP_POOL_ARGS_OK(dy, dx, c), "sp_upsample2_bwd: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // bf16 tensors with C % 8 == 0 move 16 bytes per lane, everything else 4 elements
    const int v = (dtype == SP_BF16 && c % 8 == 0) ? 8 : 4;
    const int groups = c / v, lanes = groups < 256 ? groups : 256, pix_par = 256 / lanes;
    int bx = ((w_ + 1) / 2 + pix_par - 1) / pix_par;                  // two adjacent input columns per thread
    if (bx < 1) bx = 1;
    const dim3 g(bx, (long)n * h < 65535 ? n * h : 65535);
    if (dtype == SP_F32) hipLaunchKernelGGL((upsample2_bwd_kernel<float, 4>), g, dim3(256), 0, s, (const float*)dy, (float*)dx, n, h, w_, c);
    else if (v == 8) hipLaunchKernelGGL((upsample2_bwd_kernel<bf16, 8>), g, dim3(256), 0, s, (const bf16*)dy, (bf16*)dx, n, h, w_, c);
    else hipLaunchKernelGGL((upsample2_bwd_kernel<bf16, 4>), g, dim3(256), 0, s, (const bf16*)dy, (bf16*)dx, n, h, w_, c);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
