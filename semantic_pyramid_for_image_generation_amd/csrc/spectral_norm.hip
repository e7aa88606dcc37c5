// Spectral normalisation for all layers of one network in three launches (forward) - replaces the
// per-layer forward-pre-hook of torch.nn.utils.spectral_norm (60 call sites):
//   v <- normalize(W^T u);  u <- normalize(W v);  sigma = u . (W v);  W_sn = W / sigma
// plus the re-layout of W_sn into the MFMA-friendly packings the convolution / linear kernels read.
// Layers are described by a device-resident table (sp_sn_layer); persistent blocks walk the work units of all layers
// (sn_for_each_unit below).
//
// Per-call scratch (fp32, offsets from the table): t[cols] (becomes the v snapshot), s[rows] (= W v),
// usnap[rows], scal[4] = {sigma, 1/sigma, -, -}; at part_off: ceil(rows/128) x cols partial sums of W^T u.  The snapshots are what the backward of THIS forward
// needs: the discriminator runs 2-3 forwards (each with its own power iteration) before a backward.
#include <type_traits>
#include "common.h"

namespace {

constexpr float SN_EPS = 1e-12f;

// Ragged batch over the layers of a network.  The first version used grid.y = layer with grid.x / grid.z sized for the LARGEST
// layer and let the blocks past a layer's extent exit: 12 k - 57 k mostly empty workgroups per launch, whose dispatch alone
// took longer than the useful work (rocprof: 27 + 27 + 41 + 66 us for ~10 us of memory traffic each).  Now a fixed grid of
// persistent blocks walks the work units of all layers: every block builds the prefix of units per layer in LDS (one table
// read per layer, in parallel) and finds the layer of a unit by bisection.
constexpr int SN_MAX_LAYERS = 256, SN_GRID = 1024;
template <typename Tab, typename UnitsOf, typename Body>
__device__ __forceinline__ void sn_for_each_unit(const Tab* __restrict__ table, int n_layers, UnitsOf units_of, Body body) {
    __shared__ int pre[SN_MAX_LAYERS + 1];
    for (int l = threadIdx.x; l < n_layers; l += 256) pre[l + 1] = units_of(table[l]);
    if (threadIdx.x == 0) pre[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int l = 1; l <= n_layers; ++l) pre[l] += pre[l - 1];
    __syncthreads();
    const int total = pre[n_layers];
    for (int u = blockIdx.x; u < total; u += gridDim.x) {
        int lo = 0, hi = n_layers - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pre[mid] <= u) lo = mid; else hi = mid - 1;
        }
        body(table[lo], lo, u - pre[lo]);
        __syncthreads();                                   // the bodies use block-wide LDS scratch
    }
}

// phase 1a: part[z][c] = sum_{r in row slab z} W[r][c] * u[r]   (one 128-row slab per blockIdx.z; plain stores)
constexpr int SN_SLAB = 128;
__global__ __launch_bounds__(256) void sn_wtu_kernel(const sp_sn_layer* __restrict__ table, int n_layers, float* __restrict__ scratch) {
    __shared__ float us[SN_SLAB];
    __shared__ float4 red[4][64];
    sn_for_each_unit(table, n_layers,
        [](const sp_sn_layer& L) { return ((L.cols + 255) / 256) * ((L.rows + SN_SLAB - 1) / SN_SLAB); },
        [&](const sp_sn_layer& L, int, int local) {
            const int ncb = (L.cols + 255) / 256;
            const int z = local / ncb, bx = local - z * ncb;
            const int r0 = z * SN_SLAB, r1 = min(r0 + SN_SLAB, L.rows);
            if (threadIdx.x < r1 - r0) us[threadIdx.x] = L.u[r0 + threadIdx.x];
            __syncthreads();
            if ((L.cols & 3) == 0) {
                // four columns per lane (16-byte loads, a wave reads 1 KB of a row), the slab's rows dealt to the four waves in
                // groups of 32: eight independent loads in flight per lane instead of one dependent chain of 128
                const int q = threadIdx.x & 63, rg = threadIdx.x >> 6;
                const int c4 = bx * 256 + q * 4;
                float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c4 < L.cols) {
                    const int ra = r0 + rg * 32, rb = min(ra + 32, r1);
                    const float* w = L.w + (long)ra * L.cols + c4;
#pragma unroll 8
                    for (int r = 0; r < rb - ra; ++r) {
                        const float4 v = *reinterpret_cast<const float4*>(w + (long)r * L.cols);
                        const float uu = us[rg * 32 + r];
                        a4.x += v.x * uu; a4.y += v.y * uu; a4.z += v.z * uu; a4.w += v.w * uu;
                    }
                }
                red[rg][q] = a4;
                __syncthreads();
                if (rg == 0 && c4 < L.cols) {
                    float4 t = red[0][q];
#pragma unroll
                    for (int k = 1; k < 4; ++k) { const float4 v = red[k][q]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
                    *reinterpret_cast<float4*>(scratch + L.part_off + (long)z * L.cols + c4) = t;
                }
                return;
            }
            const int c = bx * 256 + threadIdx.x;
            if (c >= L.cols) return;
            float acc = 0.f;
            const float* w = L.w + (long)r0 * L.cols + c;
#pragma unroll 8
            for (int r = 0; r < r1 - r0; ++r) acc += w[(long)r * L.cols] * us[r];
            scratch[L.part_off + (long)z * L.cols + c] = acc;
        });
}
// phase 1b: t[c] = sum_z part[z][c] in slab order (the atomics this replaces made even the forward pass vary run to run)
__global__ __launch_bounds__(256) void sn_tsum_kernel(const sp_sn_layer* __restrict__ table, float* __restrict__ scratch) {
    const sp_sn_layer L = table[blockIdx.y];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= L.cols) return;
    const int nz = (L.rows + SN_SLAB - 1) / SN_SLAB;
    const float* part = scratch + L.part_off + c;
    float t = 0.f;
    for (int z = 0; z < nz; ++z) t += part[(long)z * L.cols];
    scratch[L.scratch_off + c] = t;
}

// phase 2: one wave per row: s[r] = W[r] . v  with v = t / max(||t||, eps) (power iteration) or the
// stored v (eval mode).  The wave accumulates ||t||^2 on the same pass.  Row 0's wave also writes v.
__global__ __launch_bounds__(256) void sn_wv_kernel(const sp_sn_layer* __restrict__ table, int n_layers, float* __restrict__ scratch,
                                                    int power_iter) {
    sn_for_each_unit(table, n_layers, [](const sp_sn_layer& L) { return (L.rows + 3) / 4; },
        [&](const sp_sn_layer& L, int, int local) {
            const int r = local * 4 + (threadIdx.x >> 6);
            const int lane = threadIdx.x & 63;
            if (r >= L.rows) return;
            float* t = scratch + L.scratch_off;
            float* s = t + L.cols;
            const float* vin = power_iter ? t : L.v;
            const float* w = L.w + (long)r * L.cols;
            float dot = 0.f, nn = 0.f;
            if ((L.cols & 3) == 0) {
                // 16-byte loads, four of them in flight per lane (the scalar loop was one dependent round trip per 64 columns)
                const float4* w4 = reinterpret_cast<const float4*>(w);
                const float4* v4 = reinterpret_cast<const float4*>(vin);
                const int n4 = L.cols >> 2;
#pragma unroll 4
                for (int c = lane; c < n4; c += 64) {
                    const float4 a = w4[c], tv = v4[c];
                    dot += a.x * tv.x + a.y * tv.y + a.z * tv.z + a.w * tv.w;
                    nn += tv.x * tv.x + tv.y * tv.y + tv.z * tv.z + tv.w * tv.w;
                }
            } else {
#pragma unroll 4
                for (int c = lane; c < L.cols; c += 64) {
                    const float tv = vin[c];
                    dot += w[c] * tv;
                    nn += tv * tv;
                }
            }
            dot = wave_sum(dot);
            nn = wave_sum(nn);
            float inv = 1.f;
            if (power_iter) inv = 1.f / fmaxf(sqrtf(nn), SN_EPS);
            if (lane == 0) s[r] = dot * inv;
            if (power_iter && r == 0) {
                // v <- t/||t|| : persistent buffer now, snapshot (in place over t) by the pack kernel later would
                // race with other rows still reading t, so only the persistent copy is written here.
                for (int c = lane; c < L.cols; c += 64) L.v[c] = vin[c] * inv;
            }
        });
}

// phase 3: one block per layer finalizes u / sigma and takes the (u, v) snapshots the backward of this forward needs.
__global__ __launch_bounds__(256) void sn_finalize_kernel(const sp_sn_layer* __restrict__ table, float* __restrict__ scratch,
                                                          int power_iter) {
    const sp_sn_layer L = table[blockIdx.x];
    __shared__ float red[4];
    float* t = scratch + L.scratch_off;
    float* s = t + L.cols;
    float* usnap = s + L.rows;
    float* scal = usnap + L.rows;
    float part = 0.f;
    for (int r = threadIdx.x; r < L.rows; r += 256) {
        const float sv = s[r];
        part += power_iter ? sv * sv : sv * L.u[r];
    }
    const float tot = block_sum_256(part, red);
    float sigma, inv_norm = 1.f;
    if (power_iter) {
        const float nrm = sqrtf(tot);
        inv_norm = 1.f / fmaxf(nrm, SN_EPS);
        sigma = tot * inv_norm;                         // u . (W v) with u = s / ||s||
    } else {
        sigma = tot;
    }
    for (int r = threadIdx.x; r < L.rows; r += 256) {
        const float uv = power_iter ? s[r] * inv_norm : L.u[r];
        if (power_iter) L.u[r] = uv;
        usnap[r] = uv;
    }
    for (int c = threadIdx.x; c < L.cols; c += 256) t[c] = L.v[c];   // v snapshot (phase 2 finished)
    if (threadIdx.x == 0) { scal[0] = sigma; scal[1] = 1.f / sigma; }
}

// phase 4: W / sigma into the forward packing [rows][taps][cin_p] and the input-gradient packing
// [cin][flipped tap][cout_p].  A block moves a 32 (co) x 32 (ci) x taps tile through LDS: the source rows are read
// as contiguous runs of 32*taps floats, both packings are written as 32 consecutive elements per wave-half
// (row pitch 32*taps + 1 floats: both transposed reads are bank-conflict free).
constexpr int SN_TILE = 32, SN_MAX_TAPS = 9;
template <typename T>
__global__ __launch_bounds__(256) void sn_pack_kernel(const sp_sn_layer* __restrict__ table, const float* __restrict__ scratch,
                                                      char* __restrict__ pack, int n_layers, int flat) {
    // flat: 1-D grid, layer i owns blocks [pack_block0[i], pack_block0[i+1]) - found by bisection (wave-uniform loads)
    int layer = blockIdx.y, bx = blockIdx.x;
    if (flat) {
        int lo = 0, hi = n_layers - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[mid].pack_block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        layer = lo;
        bx = (int)blockIdx.x - table[lo].pack_block0;
    }
    const sp_sn_layer L = table[layer];
    const float inv_sigma = scratch[L.scratch_off + L.cols + 2 * L.rows + 1];
    if (L.kind == 1) {   // plain fp32 copy [rows][cols] (spectral-normalised nn.Embedding, models.py:135)
        const long chunk0 = (long)bx * 1024;
        float* out = reinterpret_cast<float*>(pack + L.fwd_off);
        for (long e = chunk0 + threadIdx.x; e < chunk0 + 1024 && e < (long)L.rows * L.cols; e += 256)
            out[e] = L.w[e] * inv_sigma;
        return;
    }
    const int cit = (L.cin_p + SN_TILE - 1) / SN_TILE, cot = (L.cout_p + SN_TILE - 1) / SN_TILE;
    if (bx >= cit * cot) return;
    const int co0 = (bx / cit) * SN_TILE, ci0 = (bx % cit) * SN_TILE;
    const int taps = L.taps;
    const int pitch = SN_TILE * taps + 1;
    __shared__ float tile[SN_TILE * (SN_TILE * SN_MAX_TAPS + 1)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int run = SN_TILE * taps;                     // floats per source row of this tile
    // source tile -> LDS.  The 3x3 and 1x1 cases are unrolled completely (36 / 4 independent loads per thread in flight; the
    // generic loop below waited for every load before the next: most of the kernel's 68 us)
    const float* src0 = L.w + (long)co0 * L.cols + (long)ci0 * taps;
    const int vrun = min(run, L.cols - ci0 * taps);
    auto stage = [&](auto taps_c) {
        constexpr int TAPS = decltype(taps_c)::value, RUN = SN_TILE * TAPS, PITCH = RUN + 1, N = SN_TILE * RUN / 256;
        float v[N];
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const int e = threadIdx.x + q * 256;
            const int r = e / RUN, k = e - r * RUN;
            v[q] = (co0 + r < L.rows && k < vrun) ? src0[(long)r * L.cols + k] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const int e = threadIdx.x + q * 256;
            const int r = e / RUN, k = e - r * RUN;
            tile[r * PITCH + k] = v[q] * inv_sigma;
        }
    };
    if (taps == 9) stage(std::integral_constant<int, 9>{});
    else if (taps == 1) stage(std::integral_constant<int, 1>{});
    else {
        for (int r = wave; r < SN_TILE; r += 4) {
            const int co = co0 + r;
            const float* src = L.w + (long)co * L.cols + (long)ci0 * taps;
            const int valid = co < L.rows ? vrun : 0;
            for (int k = lane; k < run; k += 64) tile[r * pitch + k] = k < valid ? src[k] * inv_sigma : 0.f;
        }
    }
    __syncthreads();
    // a lane writes TWO consecutive elements (4 bytes for bf16): 16 groups of 16 lanes, each group one 32-element row segment
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    if (L.fwd_off >= 0) {
        T* fwd = reinterpret_cast<T*>(pack + L.fwd_off);
        const int ci = ci0 + 2 * l16;
        if (ci < L.cin_p)
            for (int q = grp; q < SN_TILE * taps; q += 16) {       // q = (co_l, tap)
                const int r = q / taps, tap = q - r * taps;
                const int co = co0 + r;
                if (co < L.rows)
                    Elem<T>::st2(fwd + ((long)co * taps + tap) * L.cin_p + ci, tile[r * pitch + (2 * l16) * taps + tap],
                                 tile[r * pitch + (2 * l16 + 1) * taps + tap]);
            }
    }
    if (L.dgrad_off >= 0) {
        T* dg = reinterpret_cast<T*>(pack + L.dgrad_off);
        const int co = co0 + 2 * l16;
        if (co < L.cout_p)
            for (int q = grp; q < SN_TILE * taps; q += 16) {       // q = (ci_l, tap)
                const int c = q / taps, tap = q - c * taps;
                const int ci = ci0 + c;
                if (ci < L.cin)
                    Elem<T>::st2(dg + ((long)ci * taps + (taps - 1 - tap)) * L.cout_p + co, tile[(2 * l16) * pitch + c * taps + tap],
                                 tile[(2 * l16 + 1) * pitch + c * taps + tap]);
            }
    }
}

// ---- backward (per layer): dW = (dWsn - <dWsn, Wsn> u v^T) / sigma, <dWsn,Wsn> = <dWsn,W>/sigma
// dWsn comes in the forward packing [rows][taps][cin_p] (kind 0) or plain [rows][cols] (kind 1).
__global__ __launch_bounds__(256) void sn_bwd_dot_kernel(const float* __restrict__ dwsn, const float* __restrict__ w,
                                                         int rows, int cols, int cin, int taps, int cin_p, int plain,
                                                         float* __restrict__ dot_out) {
    __shared__ float red[4];
    const long total = (long)rows * cols;
    float part = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        long src = e;
        if (!plain) {
            const int r = (int)(e / cols), c = (int)(e % cols);
            const int ci = c / taps, tap = c - ci * taps;
            src = ((long)r * taps + tap) * cin_p + ci;
        }
        part += dwsn[src] * w[e];
    }
    const float tot = block_sum_256(part, red);
    if (threadIdx.x == 0) atomicAdd(dot_out, tot);
}

__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(const float* __restrict__ dwsn, const float* __restrict__ usnap,
                                                           const float* __restrict__ vsnap, const float* __restrict__ scal,
                                                           const float* __restrict__ dot, int dot_normalised, int rows,
                                                           int cols, int cin, int taps, int cin_p, int plain,
                                                           float* __restrict__ grad) {
    const long total = (long)rows * cols;
    const float inv_sigma = scal[1];
    const float coef = dot_normalised ? dot[0] : dot[0] * inv_sigma;     // <dwsn, W/sigma>
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int r = (int)(e / cols), c = (int)(e % cols);
        long src = e;
        if (!plain) {
            const int ci = c / taps, tap = c - ci * taps;
            src = ((long)r * taps + tap) * cin_p + ci;
        }
        grad[e] = (dwsn[src] - coef * usnap[r] * vsnap[c]) * inv_sigma;
    }
}


// ---- backward, batched over the layers of one network (one launch pair per backward pass instead of one per layer):
// grid.y = layer, blocks past a layer's extent exit.  dots are zero-filled by the caller (they live in the same arena as
// the dW slots, which the caller zero-fills once per backward pass).
constexpr int SN_DOT_BLOCKS = 512;      // blocks (= partial sums) per layer of the batched dot kernel
// 3x3 layers: dWsn arrives in the forward packing [row][tap][cin_p], W and the gradient are [row][cin][tap].  Read element by element
// in the destination's order, a wave gathers seven floats from each of nine tap planes per load (the apply kernel ran at 2.3 TB/s).
// Here a work item is (row, chunk of 256 input channels): the nine tap rows of the chunk are read as they lie (thread = channel),
// turned in LDS ([channel][tap], pitch 9 floats: odd, conflict-free), and everything in the destination's order - W, v, the
// gradient - is then one contiguous run of up to 2 304 floats.
constexpr int SN_TR_CI = 256, SN_TR_MAXTAPS = 9;
__device__ __forceinline__ int sn_tr_chunks(const sp_sn_bwd_layer& L) { return (L.cin + SN_TR_CI - 1) / SN_TR_CI; }
// stages chunk `ch` of row r into tile[]; returns the number of floats staged (channels x taps); callers __syncthreads() around it
__device__ __forceinline__ int sn_tr_stage(const sp_sn_bwd_layer& L, const float* __restrict__ dwsn, int r, int ch, float* tile) {
    const int ci0 = ch * SN_TR_CI, nci = min(SN_TR_CI, L.cin - ci0);
    const int t = threadIdx.x;
    if (t < nci) {
        float v[SN_TR_MAXTAPS];
#pragma unroll
        for (int tap = 0; tap < SN_TR_MAXTAPS; ++tap)
            v[tap] = tap < L.taps ? dwsn[((long)r * L.taps + tap) * L.cin_p + ci0 + t] : 0.f;
#pragma unroll
        for (int tap = 0; tap < SN_TR_MAXTAPS; ++tap)
            if (tap < L.taps) tile[t * L.taps + tap] = v[tap];
    }
    return nci * L.taps;
}

__global__ __launch_bounds__(256) void sn_bwd_dot_batched_kernel(const sp_sn_bwd_layer* __restrict__ table, const float* __restrict__ arena,
                                                                 float* __restrict__ dot_partials) {
    const sp_sn_bwd_layer L = table[blockIdx.y];
    const long total = (long)L.rows * L.cols;
    const long nb = min((long)gridDim.x, (total + 1023) / 1024);      // blocks working on this layer
    if ((long)blockIdx.x >= nb) return;
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) float tile[SN_TR_CI * SN_TR_MAXTAPS];
    const float* dwsn = arena + L.dw_off;
    float part = 0.f;
    if (!L.plain && L.taps > 1 && L.taps <= SN_TR_MAXTAPS) {
        const int chunks = sn_tr_chunks(L);
        const long items = (long)L.rows * chunks;
        for (long it = blockIdx.x; it < items; it += nb) {
            const int r = (int)(it / chunks), ch = (int)(it - (long)r * chunks);
            __syncthreads();
            const int cnt = sn_tr_stage(L, dwsn, r, ch, tile);
            __syncthreads();
            const float* wrow = L.w + (long)r * L.cols + (long)ch * SN_TR_CI * L.taps;
            if (((L.cols | cnt) & 3) == 0 && (reinterpret_cast<uintptr_t>(L.w) & 15) == 0) {      // 16-byte runs (every layer but the RGB one)
                for (int q = threadIdx.x * 4; q < cnt; q += 1024) {
                    const float4 a = *reinterpret_cast<const float4*>(tile + q), wv = *reinterpret_cast<const float4*>(wrow + q);
                    part += (a.x * wv.x + a.y * wv.y) + (a.z * wv.z + a.w * wv.w);
                }
            } else {
                for (int q = threadIdx.x; q < cnt; q += 256) part += tile[q] * wrow[q];
            }
        }
    } else {
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += nb * 256) {
            long src = e;
            if (!L.plain) {
                const int r = (int)(e / L.cols), c = (int)(e % L.cols);
                const int ci = c / L.taps, tap = c - ci * L.taps;
                src = ((long)r * L.taps + tap) * L.cin_p + ci;
            }
            part += dwsn[src] * L.w[e];
        }
    }
    const float tot = block_sum_256(part, red);
    if (threadIdx.x == 0) dot_partials[(long)blockIdx.y * SN_DOT_BLOCKS + blockIdx.x] = tot;     // summed in block order by the apply kernel
}

__global__ __launch_bounds__(256) void sn_bwd_apply_batched_kernel(const sp_sn_bwd_layer* __restrict__ table, const float* __restrict__ arena,
                                                                   const float* __restrict__ scratch, float* grads, const float* prev,
                                                                   float* bias_grads, const float* __restrict__ dot_partials, float grad_scale,
                                                                   const float* __restrict__ grad_scale_dev) {
    if (grad_scale_dev != nullptr) grad_scale = *grad_scale_dev;      // sp_sn_backward_batched_dscaled: the fp16 mode's dynamic loss scale
    const sp_sn_bwd_layer L = table[blockIdx.y];
    if (bias_grads != nullptr && blockIdx.x == 0) {            // bias gradients: plain sums, copied / added as they are
        for (int r = threadIdx.x; r < L.rows; r += 256) {
            const float b = arena[L.db_off + r] * grad_scale;
            bias_grads[L.bias_off + r] = prev ? bias_grads[L.bias_off + r] + b : b;
        }
    }
    const long total = (long)L.rows * L.cols;
    const long nb = min((long)gridDim.x, (total + 1023) / 1024);
    if ((long)blockIdx.x >= nb) return;
    // <dwsn, W>: the layer's per-block partial sums, added in the same order by every block (no atomics, no zero fill)
    __shared__ float red[4];
    float dpart = 0.f;
    for (long k = threadIdx.x; k < nb; k += 256) dpart += dot_partials[(long)blockIdx.y * SN_DOT_BLOCKS + k];
    const float dot = block_sum_256(dpart, red);
    const float* dwsn = arena + L.dw_off;
    const float* vsnap = scratch + L.scratch_off;
    const float* usnap = vsnap + L.cols + L.rows;
    const float inv_sigma = usnap[L.rows + 1];
    const float coef = dot * inv_sigma;                    // <dwsn, W/sigma>
    float* grad = grads + L.grad_off;
    const float* acc = prev ? prev + L.grad_off : nullptr;    // may alias grad: every element is read, then written, by one thread
    if (!L.plain && L.taps > 1 && L.taps <= SN_TR_MAXTAPS) {
        __shared__ __attribute__((aligned(16))) float tile[SN_TR_CI * SN_TR_MAXTAPS];
        const int chunks = sn_tr_chunks(L);
        const long items = (long)L.rows * chunks;
        for (long it = blockIdx.x; it < items; it += nb) {
            const int r = (int)(it / chunks), ch = (int)(it - (long)r * chunks);
            __syncthreads();
            const int cnt = sn_tr_stage(L, dwsn, r, ch, tile);
            __syncthreads();
            const int c0 = ch * SN_TR_CI * L.taps;
            const long e0 = (long)r * L.cols + c0;
            const float cu = coef * usnap[r];
            if (((L.cols | cnt | (int)(L.grad_off & 3) | (int)(L.scratch_off & 3)) & 3) == 0 && (reinterpret_cast<uintptr_t>(grads) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(scratch) & 15) == 0 && (acc == nullptr || (reinterpret_cast<uintptr_t>(prev) & 15) == 0)) {
                for (int q = threadIdx.x * 4; q < cnt; q += 1024) {
                    const float4 a = *reinterpret_cast<const float4*>(tile + q), vv = *reinterpret_cast<const float4*>(vsnap + c0 + q);
                    float4 g4;
                    g4.x = (a.x - cu * vv.x) * inv_sigma * grad_scale;
                    g4.y = (a.y - cu * vv.y) * inv_sigma * grad_scale;
                    g4.z = (a.z - cu * vv.z) * inv_sigma * grad_scale;
                    g4.w = (a.w - cu * vv.w) * inv_sigma * grad_scale;
                    if (acc) { const float4 p4 = *reinterpret_cast<const float4*>(acc + e0 + q); g4.x += p4.x; g4.y += p4.y; g4.z += p4.z; g4.w += p4.w; }
                    *reinterpret_cast<float4*>(grad + e0 + q) = g4;
                }
            } else {
                for (int q = threadIdx.x; q < cnt; q += 256) {
                    const float gv = (tile[q] - cu * vsnap[c0 + q]) * inv_sigma * grad_scale;
                    grad[e0 + q] = acc ? acc[e0 + q] + gv : gv;
                }
            }
        }
        return;
    }
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += nb * 256) {
        const int r = (int)(e / L.cols), c = (int)(e % L.cols);
        long src = e;
        if (!L.plain) {
            const int ci = c / L.taps, tap = c - ci * L.taps;
            src = ((long)r * L.taps + tap) * L.cin_p + ci;
        }
        const float gv = (dwsn[src] - coef * usnap[r] * vsnap[c]) * inv_sigma * grad_scale;      // (x 1.0f is exact: the plain entry's results are unchanged)
        grad[e] = acc ? acc[e] + gv : gv;
    }
}

// One-off packing of a frozen fp32 weight (no spectral norm): the VGG-16 pyramid (models.py:176-181).
// chw_c > 0 permutes the input-feature index from NCHW-flatten order (c*hw + s) to NHWC order (s*C + c), which
// lets the classifier consume the NHWC avg-pool output directly (models.py:208 flattens NCHW).
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, int rows, int cols, int cin, int taps, int cin_p, int cout_p,
                                   int chw_c, int chw_hw, T* __restrict__ fwd, T* __restrict__ dg) {
    const long fwd_n = fwd ? (long)rows * taps * cin_p : 0;
    const long dg_n = dg ? (long)cin * taps * cout_p : 0;
    const long total = fwd_n > dg_n ? fwd_n : dg_n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        if (e < fwd_n) {
            const int ci = (int)(e % cin_p);
            const long q = e / cin_p;
            const int tap = (int)(q % taps);
            const int r = (int)(q / taps);
            float v = 0.f;
            if (ci < cin) {
                int src = ci;
                if (chw_c > 0) { const int sp = ci / chw_c, c = ci - sp * chw_c; src = c * chw_hw + sp; }
                v = w[(long)r * cols + (long)src * taps + tap];
            }
            Elem<T>::st(fwd + e, v);
        }
        if (e < dg_n) {
            const int co = (int)(e % cout_p);
            const long q = e / cout_p;
            const int tapf = (int)(q % taps);
            const int ci = (int)(q / taps);
            const int tap = taps - 1 - tapf;
            float v = 0.f;
            if (co < rows) {
                int src = ci;
                if (chw_c > 0) { const int sp = ci / chw_c, c = ci - sp * chw_c; src = c * chw_hw + sp; }
                v = w[(long)co * cols + (long)src * taps + tap];
            }
            Elem<T>::st(dg + e, v);
        }
    }
}

}  // namespace

extern "C" int sp_sn_forward(const sp_sn_layer* table_dev, int32_t n_layers, int32_t max_rows, int32_t max_cols,
                             int64_t max_pack_elems, float* scratch, int64_t scratch_floats, void* pack_arena,
                             int32_t power_iter, int32_t dtype, int32_t pack_blocks, sp_stream_t stream) {
    SP_CHECK_ARG(table_dev && scratch && pack_arena && pack_blocks >= 0, "sp_sn_forward: null pointer / negative pack_blocks");
    SP_CHECK_ARG(n_layers > 0 && n_layers <= SN_MAX_LAYERS && max_rows > 0 && max_cols > 0 && max_pack_elems > 0, "sp_sn_forward: bad extents");
    SP_CHECK_ARG(dtype == SP_F32 || dtype == SP_BF16, "sp_sn_forward: bad dtype %d", dtype);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (power_iter) {
        hipLaunchKernelGGL(sn_wtu_kernel, dim3(SN_GRID), dim3(256), 0, s, table_dev, n_layers, scratch);
        hipLaunchKernelGGL(sn_tsum_kernel, dim3(sp_div_up(max_cols, 256), n_layers), dim3(256), 0, s, table_dev, scratch);
        SP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sn_wv_kernel, dim3(SN_GRID), dim3(256), 0, s, table_dev, n_layers, scratch, power_iter);
    SP_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_finalize_kernel, dim3(n_layers), dim3(256), 0, s, table_dev, scratch, power_iter);
    SP_LAUNCH_CHECK();
    dim3 pgrid(sp_div_up(max_pack_elems, 1024), n_layers);     // >= tiles of the largest layer (a tile holds >= 1024 packed elements)
    if (pack_blocks > 0) pgrid = dim3((unsigned)pack_blocks);
    const int flat = pack_blocks > 0 ? 1 : 0;
    if (dtype == SP_F32)
        hipLaunchKernelGGL(sn_pack_kernel<float>, pgrid, dim3(256), 0, s, table_dev, scratch, (char*)pack_arena, n_layers, flat);
    else
        hipLaunchKernelGGL(sn_pack_kernel<bf16>, pgrid, dim3(256), 0, s, table_dev, scratch, (char*)pack_arena, n_layers, flat);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// two forwards of one network inside one step (sp_conv_params.img_scale): per layer {1, sigma_a / sigma_b}
static __global__ void sn_pair_scales_kernel(const sp_sn_layer* __restrict__ table, int n_layers, const float* __restrict__ scratch_a,
                                      const float* __restrict__ scratch_b, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_layers) return;
    const long so = table[i].scratch_off + table[i].cols + 2L * table[i].rows;     // {sigma, 1 / sigma}
    out[2 * i] = 1.f;
    out[2 * i + 1] = scratch_a[so] * scratch_b[so + 1];
}

extern "C" int sp_sn_pair_scales(const sp_sn_layer* table_dev, int32_t n_layers, const float* scratch_a, const float* scratch_b,
                                 float* out, sp_stream_t stream) {
    SP_CHECK_ARG(table_dev && scratch_a && scratch_b && out && n_layers > 0, "sp_sn_pair_scales: bad args");
    hipLaunchKernelGGL(sn_pair_scales_kernel, dim3(sp_div_up(n_layers, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table_dev,
                       n_layers, scratch_a, scratch_b, out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_sn_backward(const float* dwsn, const float* w_orig, const float* layer_scratch, int32_t rows,
                              int32_t cols, int32_t cin, int32_t taps, int32_t cin_p, int32_t plain, float* dot_tmp,
                              int32_t dot_ready, float* grad, sp_stream_t stream) {
    SP_CHECK_ARG(dwsn && w_orig && layer_scratch && dot_tmp && grad, "sp_sn_backward: null pointer");
    SP_CHECK_ARG(rows > 0 && cols > 0 && (plain || cin * taps == cols), "sp_sn_backward: bad dims");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long total = (long)rows * cols;
    int blocks = sp_div_up(total, 1024);
    if (blocks > 1024) blocks = 1024;
    // dot_ready: 0 = compute <dwsn, W> here; 3 = same, but dot_tmp was already zero-filled by the caller
    // (sp_conv2d_wgrad_fused's single fill); 2 = sp_conv2d_wgrad_fused delivered <dwsn, W/sigma>; 1 = <dwsn, W> given
    if (dot_ready == 0 || dot_ready == 3) {
        if (dot_ready == 0) {
            hipError_t e = hipMemsetAsync(dot_tmp, 0, sizeof(float), s);
            if (e != hipSuccess) { sp_set_error("sp_sn_backward: memset failed: %s", hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        }
        hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(blocks), dim3(256), 0, s, dwsn, w_orig, rows, cols, cin, taps, cin_p, plain, dot_tmp);
        SP_LAUNCH_CHECK();
    }
    const float* vsnap = layer_scratch;
    const float* usnap = layer_scratch + cols + rows;
    const float* scal = usnap + rows;
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, s, dwsn, usnap, vsnap, scal, dot_tmp, dot_ready == 2 ? 1 : 0, rows,
                       cols, cin, taps, cin_p, plain, grad);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_sn_backward_batched(const sp_sn_bwd_layer* table_dev, int32_t n_layers, int64_t max_elems, float* arena,
                                      const float* scratch, float* grads, const float* accumulate_from, float* bias_grads,
                                      float* dot_partials, sp_stream_t stream) {
    return sp_sn_backward_batched_scaled(table_dev, n_layers, max_elems, arena, scratch, grads, accumulate_from, bias_grads, dot_partials, 1.0f, stream);
}

static int sn_backward_batched(const sp_sn_bwd_layer* table_dev, int32_t n_layers, int64_t max_elems, float* arena, const float* scratch, float* grads,
                               const float* accumulate_from, float* bias_grads, float* dot_partials, float grad_scale, const float* grad_scale_dev,
                               sp_stream_t stream);

extern "C" int sp_sn_backward_batched_scaled(const sp_sn_bwd_layer* table_dev, int32_t n_layers, int64_t max_elems, float* arena,
                                             const float* scratch, float* grads, const float* accumulate_from, float* bias_grads,
                                             float* dot_partials, float grad_scale, sp_stream_t stream) {
    return sn_backward_batched(table_dev, n_layers, max_elems, arena, scratch, grads, accumulate_from, bias_grads, dot_partials, grad_scale, nullptr, stream);
}

extern "C" int sp_sn_backward_batched_dscaled(const sp_sn_bwd_layer* table_dev, int32_t n_layers, int64_t max_elems, float* arena,
                                              const float* scratch, float* grads, const float* accumulate_from, float* bias_grads,
                                              float* dot_partials, const float* grad_scale_dev, sp_stream_t stream) {
    SP_CHECK_ARG(grad_scale_dev != nullptr, "sp_sn_backward_batched_dscaled: null scale pointer");
    return sn_backward_batched(table_dev, n_layers, max_elems, arena, scratch, grads, accumulate_from, bias_grads, dot_partials, 1.0f, grad_scale_dev, stream);
}

static int sn_backward_batched(const sp_sn_bwd_layer* table_dev, int32_t n_layers, int64_t max_elems, float* arena, const float* scratch, float* grads,
                               const float* accumulate_from, float* bias_grads, float* dot_partials, float grad_scale, const float* grad_scale_dev,
                               sp_stream_t stream) {
    SP_CHECK_ARG(table_dev && arena && scratch && grads && dot_partials && n_layers > 0 && max_elems > 0, "sp_sn_backward_batched: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int bx = (int)((max_elems + 1023) / 1024);
    if (bx > SN_DOT_BLOCKS) bx = SN_DOT_BLOCKS;
    dim3 grid(bx, n_layers);
    hipLaunchKernelGGL(sn_bwd_dot_batched_kernel, grid, dim3(256), 0, s, table_dev, arena, dot_partials);
    hipLaunchKernelGGL(sn_bwd_apply_batched_kernel, grid, dim3(256), 0, s, table_dev, arena, scratch, grads, accumulate_from, bias_grads,
                       dot_partials, grad_scale, grad_scale_dev);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_pack_weight(const float* w, int32_t rows, int32_t cols, int32_t cin, int32_t taps, int32_t cin_p,
                              int32_t cout_p, int32_t chw_c, int32_t chw_hw, void* fwd, void* dgrad, int32_t dtype,
                              sp_stream_t stream) {
    SP_CHECK_ARG(w && (fwd || dgrad) && rows > 0 && cin * taps == cols && cin_p >= cin && cout_p >= rows, "sp_pack_weight: bad args");
    SP_CHECK_ARG(chw_c == 0 || chw_c * chw_hw == cin, "sp_pack_weight: chw permutation does not match cin");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long a = fwd ? (long)rows * taps * cin_p : 0, b = dgrad ? (long)cin * taps * cout_p : 0;
    long blocks = ((a > b ? a : b) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == SP_F32) hipLaunchKernelGGL(pack_weight_kernel<float>, dim3((int)blocks), dim3(256), 0, s, w, rows, cols, cin, taps, cin_p, cout_p, chw_c, chw_hw, (float*)fwd, (float*)dgrad);
    else hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3((int)blocks), dim3(256), 0, s, w, rows, cols, cin, taps, cin_p, cout_p, chw_c, chw_hw, (bf16*)fwd, (bf16*)dgrad);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
