// Weight gradient of the NHWC 1x1 / 3x3 convolution on gfx950 MFMA.
//
// GEMM view per tap: dW[co][ci] = sum_p dY[p][co] * X[p + shift(tap)][ci] - the reduction runs over
// PIXELS, which is the slow (row) index of both NHWC operands, so both MFMA operands need a
// transposed read.  Tiles are staged exactly as they lie in HBM ([pixel][channel], coalesced 16-byte
// loads, zero-filled halo) and the transpose happens on the LDS read:
//   bf16: ds_read_b64_tr_b16 (gfx950 transpose read): a 16-lane group reads a [4 pixels][16 channels]
//         block and lane i receives the 4 pixels of channel i; two reads = the 8 k-values of one
//         v_mfma_f32_16x16x32_bf16 operand.  Lane group g takes pixels {g*4..g*4+3, 16+g*4..}, the
//         same permutation of the reduction index for A and B, so each 32-lane half touches 8
//         consecutive LDS rows; with a row pitch of (tile bytes + 32) they fall on distinct banks.
//   fp32: v_mfma_f32_16x16x4_f32 takes one k per lane: plain ds_read_b32 down a column. Exact fp32.
// Split-K over pixel ranges (grid.z = taps * nsplit).  The splits meet either through fp32 atomics on dW (throughput mode) or,
// in the deterministic mode (SP_TUNE_DETERMINISTIC; default for fp32 storage), through one partial slab per split that an
// ordered reduce pass sums (fp64) and adds to dW: bit-identical results run to run.
#include <cstdlib>
#include "common.h"

namespace {

template <typename T> struct WgTraits;
template <> struct WgTraits<bf16> { static constexpr int PK = 64, PAD = 32; };
template <> struct WgTraits<float> { static constexpr int PK = 32, PAD = 64; };

template <typename T, int FCO, int FCI>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         float* __restrict__ dw, int N, int H, int W, int CIN,
                                                         int COUT, int LD_DY, int ksize, int nsplit,
                                                         long px_per_split, float* __restrict__ dbias,
                                                         const T* __restrict__ w_packed, float* __restrict__ dot,
                                                         float* __restrict__ slabs, long n_dw, float* __restrict__ bias_slabs,
                                                         int bias_ld) {
    constexpr int CO_T = 2 * FCO * 16, CI_T = 2 * FCI * 16;
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int PK = WgTraits<T>::PK;
    constexpr int PA = CO_T * (int)sizeof(T) + WgTraits<T>::PAD;   // LDS row pitch of the dY tile
    constexpr int PB = CI_T * (int)sizeof(T) + WgTraits<T>::PAD;   // LDS row pitch of the X tile
    constexpr int A_CPR = CO_T / E, B_CPR = CI_T / E;              // 16-byte chunks per row
    constexpr int A_CH = PK * A_CPR, B_CH = PK * B_CPR;
    constexpr int A_PER = (A_CH + 255) / 256, B_PER = (B_CH + 255) / 256;
    constexpr int STAGE = PK * (PA + PB);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int ci0 = blockIdx.x * CI_T, co0 = blockIdx.y * CO_T;
    const int tap = blockIdx.z / nsplit, split = blockIdx.z - tap * nsplit;
    const int taps = ksize * ksize;
    int dr = 0, ds = 0;
    if (ksize == 3) { dr = tap / 3 - 1; ds = tap - (tap / 3) * 3 - 1; }
    const long M = (long)N * H * W;
    const bool pow2 = ((H & (H - 1)) == 0) && ((W & (W - 1)) == 0);
    const int logw = pow2 ? __builtin_ctz(W) : -1;
    const long hw_mask = (long)H * W - 1;
    const long p_begin = split * px_per_split;
    const long p_end = (p_begin + px_per_split < M) ? p_begin + px_per_split : M;
    const int nk = p_begin < p_end ? (int)((p_end - p_begin + PK - 1) / PK) : 0;

    // fused by-products (optional): bias gradient = column sums of dY (taken by the blocks of the centre tap and the
    // first ci tile, which see every dY row exactly once across the pixel splits) and <dW, W/sigma> for the
    // spectral-norm backward (linear in dW, so per-block partial tiles can be dotted before the atomic merge; the
    // normalised weights are read from the forward packing, which has dW's own layout -> coalesced)
    const bool do_bias = dbias != nullptr && blockIdx.x == 0 && tap == taps / 2;
    float bsum[E];
#pragma unroll
    for (int q = 0; q < E; ++q) bsum[q] = 0.f;
    // ---- staging descriptors, fixed across the K loop: byte offsets relative to the step's first pixel, LDS offsets ----
    int a_src[A_PER], a_dst[A_PER], a_row[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int ch = tid + 256 * i;
        const int row = ch / A_CPR, cc = ch - row * A_CPR;
        const int c = cc * E + co0;
        a_row[i] = row;
        a_src[i] = (ch < A_CH && c < LD_DY) ? (row * LD_DY + c) * (int)sizeof(T) : -1;
        a_dst[i] = ch < A_CH ? row * PA + cc * 16 : -1;
    }
    int b_src[B_PER], b_dst[B_PER], b_row[B_PER];
    bool b_cok[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int ch = tid + 256 * i;
        const int row = ch / B_CPR, cc = ch - row * B_CPR;
        const int c = cc * E + ci0;
        b_row[i] = row;
        b_cok[i] = ch < B_CH && c < CIN;
        b_src[i] = ((row + dr * W + ds) * CIN + c) * (int)sizeof(T);      // may be negative (halo above the first pixel)
        b_dst[i] = ch < B_CH ? PK * PA + row * PB + cc * 16 : -1;
    }
    // ---- fragment read offsets (bytes) for the first k-step; later k-steps / the second transpose read add constants ----
    const int i16 = lane & 15, g = lane >> 4;
    int fa[FCO], fb[FCI];
    if constexpr (sizeof(T) == 2) {
        const int row1 = g * 4 + (i16 >> 2);
#pragma unroll
        for (int i = 0; i < FCO; ++i) fa[i] = row1 * PA + ((wa * FCO + i) * 16 + (i16 & 3) * 4) * 2;
#pragma unroll
        for (int j = 0; j < FCI; ++j) fb[j] = PK * PA + row1 * PB + ((wb * FCI + j) * 16 + (i16 & 3) * 4) * 2;
    } else {
#pragma unroll
        for (int i = 0; i < FCO; ++i) fa[i] = g * PA + ((wa * FCO + i) * 16 + i16) * 4;
#pragma unroll
        for (int j = 0; j < FCI; ++j) fb[j] = PK * PA + g * PB + ((wb * FCI + j) * 16 + i16) * 4;
    }

    // Every request of a step is UNCONDITIONAL (round 5): a chunk that does not exist - rows past the pixel range, channels past the
    // tensor, the zero halo of a tap - re-reads the step's first pixel and is replaced by zeros when it is written to LDS, its validity
    // kept as one bit per chunk.  With the loads under their per-lane conditions hipcc wrapped each one in an exec branch and waited
    // `vmcnt(0)` behind it (the bias sums read the value at once): the requests of step k + 1 did not fly over the MFMAs of step k but
    // one after the other in front of them.  The bias sums are taken where the registers are stored.
    uint4 ar[A_PER], br[B_PER];
    unsigned amask = 0, bmask = 0;
    auto load_global = [&](int ks) {
        const long pb = p_begin + (long)ks * PK;                           // wave-uniform
        const char* dyb = reinterpret_cast<const char*>(dy + pb * LD_DY);
        const char* xb = reinterpret_cast<const char*>(x + pb * CIN);
        const int left = (int)(p_end - pb);                                // rows [0, left) of this step exist (left >= 1)
        const int rem0 = logw >= 0 ? (int)(pb & hw_mask) : (int)(pb % ((long)H * W));
        amask = 0;
        bmask = 0;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const bool ok = a_src[i] >= 0 && a_row[i] < left;
            amask |= (ok ? 1u : 0u) << i;
            ar[i] = *reinterpret_cast<const uint4*>(dyb + (ok ? a_src[i] : 0));
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            bool ok = b_cok[i] && b_row[i] < left;
            int hh, ww;
            if (logw >= 0) {                     // power-of-two H and W (every layer of this model): no integer division
                const int rem = (rem0 + b_row[i]) & (int)hw_mask;
                hh = (rem >> logw) + dr;
                ww = (rem & (W - 1)) + ds;
            } else {
                const int rem = (int)((pb + b_row[i]) % ((long)H * W));
                hh = rem / W + dr;
                ww = rem % W + ds;
            }
            ok = ok && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            bmask |= (ok ? 1u : 0u) << i;
            br[i] = *reinterpret_cast<const uint4*>(xb + (ok ? b_src[i] : 0));
        }
    };
    auto store_lds = [&](int buf) {
        char* sb = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const uint4 v = ((amask >> i) & 1u) ? ar[i] : make_uint4(0, 0, 0, 0);
            if (do_bias) {
                if constexpr (sizeof(T) == 2) {
                    bsum[0] += h16_lo_to_f32(v.x); bsum[1] += h16_hi_to_f32(v.x);
                    bsum[2] += h16_lo_to_f32(v.y); bsum[3] += h16_hi_to_f32(v.y);
                    bsum[4] += h16_lo_to_f32(v.z); bsum[5] += h16_hi_to_f32(v.z);
                    bsum[6] += h16_lo_to_f32(v.w); bsum[7] += h16_hi_to_f32(v.w);
                } else {
                    bsum[0] += __uint_as_float(v.x); bsum[1] += __uint_as_float(v.y);
                    bsum[2] += __uint_as_float(v.z); bsum[3] += __uint_as_float(v.w);
                }
            }
            if (a_dst[i] >= 0) *reinterpret_cast<uint4*>(sb + a_dst[i]) = v;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            if (b_dst[i] >= 0) *reinterpret_cast<uint4*>(sb + b_dst[i]) = ((bmask >> i) & 1u) ? br[i] : make_uint4(0, 0, 0, 0);
    };

    f32x4_t acc[FCO][FCI];
#pragma unroll
    for (int i = 0; i < FCO; ++i)
#pragma unroll
        for (int j = 0; j < FCI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) {
        load_global(0);
        store_lds(0);
    }
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_global(ks + 1);
        const char* sb = smem + buf * STAGE;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < PK / 32; ++kk) {
                uint4 a[FCO], b[FCI];
#pragma unroll
                for (int i = 0; i < FCO; ++i) {
                    const char* base = sb + fa[i] + kk * 32 * PA;
                    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base));
                    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + 16 * PA));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    a[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int j = 0; j < FCI; ++j) {
                    const char* base = sb + fb[j] + kk * 32 * PB;
                    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base));
                    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + 16 * PB));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    b[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int i = 0; i < FCO; ++i)
#pragma unroll
                    for (int j = 0; j < FCI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i]),
                                                                            __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll 2
            for (int k4 = 0; k4 < PK / 4; ++k4) {
                float a[FCO], b[FCI];
#pragma unroll
                for (int i = 0; i < FCO; ++i) a[i] = *reinterpret_cast<const float*>(sb + fa[i] + k4 * 4 * PA);
#pragma unroll
                for (int j = 0; j < FCI; ++j) b[j] = *reinterpret_cast<const float*>(sb + fb[j] + k4 * 4 * PB);
#pragma unroll
                for (int i = 0; i < FCO; ++i)
#pragma unroll
                    for (int j = 0; j < FCI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (ks + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }

    if (nk == 0) return;
    // ---- merge.  Split layers (nsplit > 1) meet through fp32 atomics straight from the MFMA layout (64-byte segments; the
    // same atomics issued 16 bytes per lane on contiguous kilobytes measured 2x SLOWER: more lanes per cache line).  A block
    // that is the only writer of its dW rows (nsplit == 1: the 4x4 layers) needs no atomics: the tile goes through LDS (the
    // staging buffers are idle, the loop ended with a barrier) and every thread read-modify-writes 16 bytes (43 -> 26 us).
    float dpart = 0.f;
    if (nsplit == 1 && slabs == nullptr) {
        constexpr int TP = CI_T + 4;
        float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < FCO; ++i)
#pragma unroll
            for (int j = 0; j < FCI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    tile[((wa * FCO + i) * 16 + (lane >> 4) * 4 + r) * TP + (wb * FCI + j) * 16 + (lane & 15)] = acc[i][j][r];
        __syncthreads();
        constexpr int C4 = CI_T / 4;
        for (int e = tid; e < CO_T * C4; e += 256) {
            const int row = e / C4, c4 = e - row * C4;
            const int co = co0 + row, ci = ci0 + c4 * 4;
            if (co >= COUT || ci >= CIN) continue;             // CIN % 4 == 0 (checked by the entry points)
            const float4 v = *reinterpret_cast<const float4*>(tile + row * TP + c4 * 4);
            const long o = ((long)co * taps + tap) * CIN + ci;
            float4 d = *reinterpret_cast<const float4*>(dw + o);
            d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w;
            *reinterpret_cast<float4*>(dw + o) = d;
            if (w_packed != nullptr)
                dpart += v.x * Elem<T>::ld(w_packed + o) + v.y * Elem<T>::ld(w_packed + o + 1) + v.z * Elem<T>::ld(w_packed + o + 2) +
                         v.w * Elem<T>::ld(w_packed + o + 3);
        }
        __syncthreads();                                       // the tile is dead: `red` below reuses the buffer
    } else {
#pragma unroll
        for (int i = 0; i < FCO; ++i) {
#pragma unroll
            for (int j = 0; j < FCI; ++j) {
                const int ci = ci0 + (wb * FCI + j) * 16 + (lane & 15);
                if (ci >= CIN) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + (wa * FCO + i) * 16 + (lane >> 4) * 4 + r;
                    if (co < COUT) {
                        const long o = ((long)co * taps + tap) * CIN + ci;
                        if (slabs != nullptr) {
                            slabs[(long)split * n_dw + o] = acc[i][j][r];       // plain store; wgrad_reduce_kernel sums the splits
                        } else {
                            atomicAdd(dw + o, acc[i][j][r]);
                            if (w_packed != nullptr) dpart += acc[i][j][r] * Elem<T>::ld(w_packed + o);
                        }
                    }
                }
            }
        }
    }
    float* red = reinterpret_cast<float*>(smem);          // the staging buffers are idle now (loop ended with a barrier)
    if (w_packed != nullptr && slabs == nullptr) {
        const float tot = block_sum_256(dpart, red);
        if (tid == 0) atomicAdd(dot, tot);
    }
    if (do_bias) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < E; ++q) red[tid * E + q] = bsum[q];
        __syncthreads();
        if (tid < CO_T) {
            const int col = tid / E, e = tid - col * E;
            float t = 0.f;
            for (int rr = 0; rr < 256 / A_CPR; ++rr) t += red[(rr * A_CPR + col) * E + e];
            if (co0 + tid < COUT) {
                if (bias_slabs != nullptr) bias_slabs[(long)split * bias_ld + co0 + tid] = t;   // one writer per (split, co)
                else atomicAdd(dbias + co0 + tid, t);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// 3x3 weight gradient, ALL NINE TAPS per block ("wgrad9"), for images at least one segment wide.
// A block owns a 64(co) x 64(ci) tile of dW for every tap and walks 64-pixel (fp32: 32-pixel) segments of image
// rows.  Per segment it stages dY[seg][64 co] and the X halo (rows h-1..h+1, columns w0-1..w0+PXS) x 64 ci ONCE and
// issues the MFMAs of all nine taps from it: the per-tap kernel above moves the same bytes nine times.  The dY
// fragments (MFMA A operand) are shared by the nine taps; the X fragments are the same LDS tile read at nine
// pixel offsets.  9 x (2x2) accumulator fragments per wave = 144 registers.  Same transpose-on-read scheme
// (ds_read_b64_tr_b16 / ds_read_b32) and the same fused by-products as the kernel above.
// ------------------------------------------------------------------------------------------------------------
template <typename T> struct Wg9Traits;
template <> struct Wg9Traits<bf16> { static constexpr int PXS = 64, PITCH = 64 * 2 + 32; };
template <> struct Wg9Traits<float> { static constexpr int PXS = 32, PITCH = 64 * 4 + 64; };

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_wgrad9_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw,
                                                          int N, int H, int W, int CIN, int COUT, int LD_DY, int segs_per_split,
                                                          float* __restrict__ dbias, const T* __restrict__ w_packed,
                                                          float* __restrict__ dot, float* __restrict__ slabs, long n_dw,
                                                          float* __restrict__ bias_slabs, int bias_ld) {
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int PXS = Wg9Traits<T>::PXS, PITCH = Wg9Traits<T>::PITCH;
    constexpr int CPR = 64 / E;                           // 16-byte chunks per 64-channel row
    constexpr int HW_COLS = PXS + 2;
    constexpr int A_CH = PXS * CPR, B_CH = 3 * HW_COLS * CPR;
    constexpr int A_PER = (A_CH + 255) / 256, B_PER = (B_CH + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* abuf = smem;                                    // [PXS][PITCH]          dY segment
    char* bbuf = smem + PXS * PITCH;                      // [3*HW_COLS][PITCH]    X halo

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
    const int segs_per_row = W / PXS;
    const long nseg = (long)N * H * segs_per_row;
    const long s_begin = (long)blockIdx.z * segs_per_split;
    const long s_end = s_begin + segs_per_split < nseg ? s_begin + segs_per_split : nseg;
    if (s_begin >= s_end) return;

    // fixed per-thread chunk descriptors
    int a_row[A_PER], a_dst[A_PER], b_r[B_PER], b_c[B_PER], b_dst[B_PER];
    const int cslot = tid % CPR;                          // 256 % CPR == 0 -> same channel slot for every chunk of a thread
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int ch = tid + 256 * i;
        a_row[i] = ch / CPR;
        a_dst[i] = ch < A_CH ? a_row[i] * PITCH + cslot * 16 : -1;
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int ch = tid + 256 * i;
        const int hp = ch / CPR;
        b_r[i] = hp / HW_COLS;
        b_c[i] = hp - b_r[i] * HW_COLS;
        b_dst[i] = ch < B_CH ? hp * PITCH + cslot * 16 : -1;
    }
    const bool a_cok = co0 + cslot * E < LD_DY;
    const bool b_cok = ci0 + cslot * E < CIN;
    const bool do_bias = dbias != nullptr && blockIdx.x == 0;
    float bsum[E];
#pragma unroll
    for (int q = 0; q < E; ++q) bsum[q] = 0.f;

    uint4 ar[A_PER], br[B_PER];
    auto load_seg = [&](long seg) {
        const int sx = (int)(seg % segs_per_row);
        const long rowid = seg / segs_per_row;            // n*H + h
        const int h = (int)(rowid % H);
        const int w0 = sx * PXS;
        const long pix0 = rowid * W + w0;                 // first pixel of the segment
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a_dst[i] >= 0 && a_cok) v = *reinterpret_cast<const uint4*>(dy + (pix0 + a_row[i]) * LD_DY + co0 + cslot * E);
            ar[i] = v;
            if (do_bias) {
                if constexpr (sizeof(T) == 2) {
                    bsum[0] += bf16_bits_to_f32(v.x & 0xffffu); bsum[1] += bf16_bits_to_f32(v.x >> 16);
                    bsum[2] += bf16_bits_to_f32(v.y & 0xffffu); bsum[3] += bf16_bits_to_f32(v.y >> 16);
                    bsum[4] += bf16_bits_to_f32(v.z & 0xffffu); bsum[5] += bf16_bits_to_f32(v.z >> 16);
                    bsum[6] += bf16_bits_to_f32(v.w & 0xffffu); bsum[7] += bf16_bits_to_f32(v.w >> 16);
                } else {
                    bsum[0] += __uint_as_float(v.x); bsum[1] += __uint_as_float(v.y);
                    bsum[2] += __uint_as_float(v.z); bsum[3] += __uint_as_float(v.w);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const int hh = h + b_r[i] - 1, ww = w0 + b_c[i] - 1;
            if (b_dst[i] >= 0 && b_cok && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
                v = *reinterpret_cast<const uint4*>(x + (pix0 + (long)(b_r[i] - 1) * W + b_c[i] - 1) * CIN + ci0 + cslot * E);
            br[i] = v;
        }
    };
    auto store_seg = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
            if (a_dst[i] >= 0) *reinterpret_cast<uint4*>(abuf + a_dst[i]) = ar[i];
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            if (b_dst[i] >= 0) *reinterpret_cast<uint4*>(bbuf + b_dst[i]) = br[i];
    };

    f32x4_t acc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int i16 = lane & 15, g = lane >> 4;
    load_seg(s_begin);
    store_seg();
    __syncthreads();
    for (long seg = s_begin; seg < s_end; ++seg) {
        const bool more = seg + 1 < s_end;
        if (more) load_seg(seg + 1);
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < PXS / 32; ++kk) {
                const int prow = kk * 32 + g * 4 + (i16 >> 2);          // pixel row of the first tr-read (second: +16)
                uint4 a[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const char* base = abuf + prow * PITCH + ((wa * 2 + i) * 16 + (i16 & 3) * 4) * 2;
                    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base));
                    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + 16 * PITCH));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    a[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int hoff = ((t / 3) * HW_COLS + (t % 3)) * PITCH;   // halo pixel (dr+1)*cols + (1+ds) + px, with -1 folded
                    uint4 b[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const char* base = bbuf + hoff + prow * PITCH + ((wb * 2 + j) * 16 + (i16 & 3) * 4) * 2;
                        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base));
                        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + 16 * PITCH));
                        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                        b[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[j]),
                                                                                   acc[t][i][j], 0, 0, 0);
                }
            }
        } else {
#pragma unroll 2
            for (int k4 = 0; k4 < PXS / 4; ++k4) {
                const int prow = k4 * 4 + g;
                float a[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const float*>(abuf + prow * PITCH + ((wa * 2 + i) * 16 + i16) * 4);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int hoff = ((t / 3) * HW_COLS + (t % 3)) * PITCH;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float bv = *reinterpret_cast<const float*>(bbuf + hoff + prow * PITCH + ((wb * 2 + j) * 16 + i16) * 4);
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv, acc[t][i][j], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        if (more) {
            store_seg();
            __syncthreads();
        }
    }

    float dpart = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ci = ci0 + (wb * 2 + j) * 16 + (lane & 15);
                if (ci >= CIN) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + (wa * 2 + i) * 16 + (lane >> 4) * 4 + r;
                    if (co < COUT) {
                        const long o = ((long)co * 9 + t) * CIN + ci;
                        if (slabs != nullptr) {
                            slabs[(long)blockIdx.z * n_dw + o] = acc[t][i][j][r];
                        } else {
                            atomicAdd(dw + o, acc[t][i][j][r]);
                            if (w_packed != nullptr) dpart += acc[t][i][j][r] * Elem<T>::ld(w_packed + o);
                        }
                    }
                }
            }
    float* red = reinterpret_cast<float*>(smem);
    if (w_packed != nullptr && slabs == nullptr) {
        const float tot = block_sum_256(dpart, red);
        if (tid == 0) atomicAdd(dot, tot);
    }
    if (do_bias) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < E; ++q) red[tid * E + q] = bsum[q];
        __syncthreads();
        if (tid < 64) {
            const int col = tid / E, e = tid - col * E;
            float t = 0.f;
            for (int rr = 0; rr < 256 / CPR; ++rr) t += red[(rr * CPR + col) * E + e];
            if (co0 + tid < COUT) {
                if (bias_slabs != nullptr) bias_slabs[(long)blockIdx.z * bias_ld + co0 + tid] = t;
                else atomicAdd(dbias + co0 + tid, t);
            }
        }
    }
}


// Adds the per-split partial tiles (slab mode), summed in split order in fp64, to dW (and the per-split bias sums to dbias);
// forms <dW, W/sigma> of the legacy fused entry in the same pass.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int nsplit, long n_dw, float* __restrict__ dw,
                                                           const T* __restrict__ w_packed, float* __restrict__ dot,
                                                           const float* __restrict__ bias_slabs, int bias_ld, int cout,
                                                           float* __restrict__ dbias) {
    __shared__ float red[4];
    float dpart = 0.f;
    for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < n_dw; e += (long)gridDim.x * 1024) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 8
        for (int sidx = 0; sidx < nsplit; ++sidx) {
            const float4 b = *reinterpret_cast<const float4*>(slabs + (long)sidx * n_dw + e);
            a0 += b.x; a1 += b.y; a2 += b.z; a3 += b.w;
        }
        float4 d = *reinterpret_cast<const float4*>(dw + e);
        const float4 a = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
        d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
        *reinterpret_cast<float4*>(dw + e) = d;
        if (w_packed != nullptr) {
            float wv[4];
            Elem<T>::ld4(w_packed + e, wv);
            dpart += a.x * wv[0] + a.y * wv[1] + a.z * wv[2] + a.w * wv[3];
        }
    }
    if (w_packed != nullptr) {
        const float tot = block_sum_256(dpart, red);
        if (threadIdx.x == 0) atomicAdd(dot, tot);
    }
    if (bias_slabs != nullptr && blockIdx.x == 0) {
        for (int c = threadIdx.x; c < cout; c += 256) {
            double t = 0.0;
#pragma unroll 8
            for (int sidx = 0; sidx < nsplit; ++sidx) t += bias_slabs[(long)sidx * bias_ld + c];
            dbias[c] += (float)t;
        }
    }
}

struct WgPlan { int nine; int nsplit; long per_split; };

template <typename T>
WgPlan plan_wgrad(int n, int h, int w, int cin, int cout, int ksize, int co_t, int ci_t) {
    WgPlan pl;
    const long M = (long)n * h * w;
    pl.nine = (ksize == 3 && w % Wg9Traits<T>::PXS == 0 && cin >= 16 && cin <= 64 && cout <= 64) ? 1 : 0;
    if (pl.nine) {
        const long nseg = (long)n * h * (w / Wg9Traits<T>::PXS);
        const int tiles = sp_div_up(cin, 64) * sp_div_up(cout, 64);
        const int target9 = sp_tune(SP_TUNE_WGRAD9_BLOCKS, 512);
        long nsplit = (target9 + tiles - 1) / tiles;
        if (nsplit > nseg / 8) nsplit = nseg / 8;
        if (nsplit < 1) nsplit = 1;
        const long sps = (nseg + nsplit - 1) / nsplit;
        pl.nsplit = (int)((nseg + sps - 1) / sps);
        pl.per_split = sps;
    } else {
        constexpr int PK = WgTraits<T>::PK;
        const int taps = ksize * ksize;
        const int tiles = sp_div_up(cin, ci_t) * sp_div_up(cout, co_t) * taps;
        const long steps = (M + PK - 1) / PK;
        // split-K trades parallelism against reduction traffic: measured optimum ~1024 blocks for large images, ~256 for
        // <= 8192 pixels (profiles/README.md); SP_TUNE_WGRAD_BLOCKS overrides
        const int env_blocks = sp_tune(SP_TUNE_WGRAD_BLOCKS, 0);
        const int target_blocks = env_blocks > 0 ? env_blocks : (M <= 8192 ? 256 : 1024);
        int nsplit = (target_blocks + tiles - 1) / tiles;
        int min_steps = sp_tune(SP_TUNE_WGRAD_MINSTEPS, 4);
        if (min_steps < 1) min_steps = 1;
        if (nsplit > steps / min_steps) nsplit = (int)(steps / min_steps);
        if (nsplit < 1) nsplit = 1;
        const long pps = ((steps + nsplit - 1) / nsplit) * PK;
        pl.nsplit = (int)((M + pps - 1) / pps);
        pl.per_split = pps;
    }
    return pl;
}

inline void wgrad_tile(int cin, int cout, int& co_t, int& ci_t) {
    co_t = cout <= 64 ? 64 : 128;
    ci_t = cin <= 64 ? 64 : 128;
}

template <typename T>
int launch_wgrad9(const T* x, const T* dy, float* dw, int n, int h, int w, int cin, int cout, int ld_dy, float* dbias,
                  const T* w_packed, float* dot, float* slabs, float* bias_slabs, int bias_ld, const WgPlan& pl, hipStream_t s) {
    constexpr int PXS = Wg9Traits<T>::PXS, PITCH = Wg9Traits<T>::PITCH;
    constexpr int LDS = (PXS + 3 * (PXS + 2)) * PITCH;
    static bool attr_set = false;
    auto kern = conv_wgrad9_kernel<T>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        attr_set = true;
    }
    const long n_dw = (long)cout * 9 * cin;
    dim3 grid(sp_div_up(cin, 64), sp_div_up(cout, 64), (unsigned)pl.nsplit);
    sp_note_route("conv_wgrad9 (per-tap, <= 64 channels)");
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, x, dy, dw, n, h, w, cin, cout, ld_dy, (int)pl.per_split, dbias, w_packed, dot, slabs, n_dw,
                       bias_slabs, bias_ld);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

template <typename T, int FCO, int FCI>
int launch_wgrad(const T* x, const T* dy, float* dw, int n, int h, int w, int cin, int cout, int ld_dy, int ksize,
                 float* dbias, const T* w_packed, float* dot, float* slabs, float* bias_slabs, int bias_ld, const WgPlan& pl,
                 hipStream_t s) {
    constexpr int CO_T = 2 * FCO * 16, CI_T = 2 * FCI * 16, PK = WgTraits<T>::PK;
    constexpr int LDS = 2 * PK * ((CO_T + CI_T) * (int)sizeof(T) + 2 * WgTraits<T>::PAD);
    static bool attr_set = false;
    auto kern = conv_wgrad_kernel<T, FCO, FCI>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        attr_set = true;
    }
    const int taps = ksize * ksize;
    const long n_dw = (long)cout * taps * cin;
    dim3 grid(sp_div_up(cin, CI_T), sp_div_up(cout, CO_T), taps * pl.nsplit);
    sp_note_route(sizeof(T) == 4 ? "conv_wgrad<f32> (per-tap)" : "conv_wgrad<16bit> (per-tap)");
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, x, dy, dw, n, h, w, cin, cout, ld_dy, ksize, pl.nsplit, pl.per_split, dbias, w_packed, dot,
                       slabs, n_dw, bias_slabs, bias_ld);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// tile shape + split plan of the per-tap kernels for one layer (shared by the dispatcher and the workspace query)
template <typename T>
WgPlan pertap_plan(int n, int h, int w, int cin, int cout, int ksize, int& co_t, int& ci_t) {
    wgrad_tile(cin, cout, co_t, ci_t);
    // tiny pixel counts (8x8, 4x4 maps): the operands live in L2, so 64 x 64 tiles cost nothing extra and give enough blocks
    // without a K split - no merge at all (SP_TUNE_WGRAD_SMALL_M, default 2048 pixels)
    const long small_m = sp_tune(SP_TUNE_WGRAD_SMALL_M, 2048);
    // 1x1 layers: dW is at most 256 x 256, so 64 x 64 tiles quadruple the tile count and shorten every merge
    const int k1_small = sp_tune(SP_TUNE_WGRAD_K1_TILE64, 1);
    if ((long)n * h * w <= small_m || (ksize == 1 && k1_small)) co_t = ci_t = 64;
    return plan_wgrad<T>(n, h, w, cin, cout, ksize, co_t, ci_t);
}

// fp32 scratch (floats) the deterministic merge of this layer needs: one dW slab + one bias row per split
template <typename T>
long pertap_slab_floats(int n, int h, int w, int cin, int cout, int ksize) {
    int co_t, ci_t;
    const WgPlan pl = pertap_plan<T>(n, h, w, cin, cout, ksize, co_t, ci_t);
    if (pl.nsplit <= 1) return 0;
    return (long)pl.nsplit * ((long)cout * ksize * ksize * cin + ((cout + 3) & ~3));
}

template <typename T>
int dispatch_wgrad(const void* x, const void* dy, float* dw, int n, int h, int w, int cin, int cout, int ld_dy,
                   int ksize, float* dbias, const void* w_packed, float* dot, float* ws, long ws_floats, hipStream_t s) {
    const T* xt = reinterpret_cast<const T*>(x);
    const T* dt = reinterpret_cast<const T*>(dy);
    const T* wpk = reinterpret_cast<const T*>(w_packed);
    const bool det = sp_deterministic(sizeof(T) == 2 ? SP_BF16 : SP_F32);
    if (sizeof(T) == 2 && ksize == 3 && cin == 8 && w_packed == nullptr && dot == nullptr) {
        // the padded RGB images: streaming kernel with an im2col X tile (conv_wgrad_1x1.hip)
        const int rc = sp_wgrad3x3_cin8_launch(x, dy, dw, dbias, n, h, w, cout, ld_dy, ws, ws_floats, s);
        if (rc != 1) return rc;
    }
    if (sizeof(T) == 2 && ksize == 3 && w_packed == nullptr && dot == nullptr) {
        // row-walker kernel (conv_wgrad_rows.hip): all nine taps per block, 4.4x fewer L2 bytes per flop
        if (sp_tune(SP_TUNE_WGRAD_ROWS, 1)) {
            const int rc = sp_wgrad_rows_launch(x, dy, dw, dbias, n, h, w, cin, cout, ld_dy, ws, ws_floats, 0, s);
            if (rc != 1) return rc;
        }
    }
    if (sizeof(T) == 2 && ksize == 1 && w_packed == nullptr && dot == nullptr) {
        // streaming kernel of the 1x1 layers (conv_wgrad_1x1.hip): long-lived blocks, LDS-DMA ring, slabs + ordered reduce
        const int rc = sp_wgrad1x1_launch(x, dy, dw, dbias, n, h, w, cin, cout, ld_dy, ws, ws_floats, s);
        if (rc != 1) return rc;
    }
    // throughput mode: the per-tap kernels keep their atomics (their slab mode measured slower); deterministic mode: slabs
    if (!det) { ws = nullptr; ws_floats = 0; }
    int co_t, ci_t;
    WgPlan pl = pertap_plan<T>(n, h, w, cin, cout, ksize, co_t, ci_t);
    const long n_dw = (long)cout * ksize * ksize * cin;
    const int bias_ld = (cout + 3) & ~3;
    // slab mode: every split stores its partial tile (and bias row) with plain stores and a second pass sums them in split
    // order; without it the splits meet through fp32 atomics.  A single split owns its dW rows: no merge either way.
    float* slabs = (ws != nullptr && pl.nsplit > 1 && ws_floats >= (long)pl.nsplit * (n_dw + bias_ld) && (n_dw & 3) == 0) ? ws : nullptr;
    if (det && pl.nsplit > 1 && slabs == nullptr) {
        // deterministic mode without (enough) scratch - sp_conv2d_wgrad has no workspace argument, the others document it as
        // optional: ONE split per tile, which then owns its dW rows (ordered, no merge; slower than the split plan)
        const long M = (long)n * h * w;
        pl.nsplit = 1;
        pl.per_split = pl.nine ? (long)n * h * (w / Wg9Traits<T>::PXS) : ((M + WgTraits<T>::PK - 1) / WgTraits<T>::PK) * WgTraits<T>::PK;
    }
    float* bias_slabs = (slabs != nullptr && dbias != nullptr) ? slabs + (long)pl.nsplit * n_dw : nullptr;
    int rc;
    if (pl.nine) rc = launch_wgrad9<T>(xt, dt, dw, n, h, w, cin, cout, ld_dy, dbias, wpk, dot, slabs, bias_slabs, bias_ld, pl, s);
    else if (co_t == 64 && ci_t == 64) rc = launch_wgrad<T, 2, 2>(xt, dt, dw, n, h, w, cin, cout, ld_dy, ksize, dbias, wpk, dot, slabs, bias_slabs, bias_ld, pl, s);
    else if (co_t == 64) rc = launch_wgrad<T, 2, 4>(xt, dt, dw, n, h, w, cin, cout, ld_dy, ksize, dbias, wpk, dot, slabs, bias_slabs, bias_ld, pl, s);
    else if (ci_t == 64) rc = launch_wgrad<T, 4, 2>(xt, dt, dw, n, h, w, cin, cout, ld_dy, ksize, dbias, wpk, dot, slabs, bias_slabs, bias_ld, pl, s);
    else rc = launch_wgrad<T, 4, 4>(xt, dt, dw, n, h, w, cin, cout, ld_dy, ksize, dbias, wpk, dot, slabs, bias_slabs, bias_ld, pl, s);
    if (rc != SP_OK || slabs == nullptr) return rc;
    long blocks = (n_dw / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, slabs, pl.nsplit, n_dw, dw, wpk, dot, bias_slabs, bias_ld,
                       cout, dbias);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

}  // namespace

extern "C" int sp_conv2d_wgrad_accum_pair(const void* x, const void* dy, float* dw_a, float* dbias_a, float* dw_b, float* dbias_b,
                                          float* workspace, int64_t workspace_floats, int32_t n, int32_t split, int32_t h, int32_t w_,
                                          int32_t cin_p, int32_t cout, int32_t ld_dy, int32_t ksize, int32_t dy_pooled, int32_t dtype,
                                          sp_stream_t stream) {
    SP_CHECK_ARG(x && dy && dw_a && dw_b && split > 0 && split < n, "sp_conv2d_wgrad_accum_pair: bad args");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SP_BF16 && ksize == 3 && cin_p == 8 && !dy_pooled) {
        // the padded RGB images: the streaming kernel, one launch over both groups + a reduce pass per group
        const int rc = sp_wgrad3x3_cin8_launch_pair(x, dy, dw_a, dbias_a, dw_b, dbias_b, n, split, h, w_, cout, ld_dy, workspace, workspace_floats, s);
        if (rc != 1) return rc;
    }
    if (dtype == SP_BF16 && ksize == 3 && cin_p != 8 && sp_tune(SP_TUNE_WGRAD_ROWS, 1)) {
        // one launch of the row walker over both groups where the group boundary falls between two of its blocks
        const int rc = sp_wgrad_rows_launch_pair(x, dy, dw_a, dbias_a, dw_b, dbias_b, n, split, h, w_, cin_p, cout, ld_dy, workspace, workspace_floats,
                                                 dy_pooled ? 1 : 0, s);
        if (rc != 1) return rc;
    }
    if (dtype == SP_BF16 && ksize == 1 && !dy_pooled) {
        const int rc = sp_wgrad1x1_launch_pair(x, dy, dw_a, dbias_a, dw_b, dbias_b, n, split, h, w_, cin_p, cout, ld_dy, workspace, workspace_floats, s);
        if (rc != 1) return rc;
    }
    // everything else: the two groups one after the other (contiguous image ranges)
    const long esz = dtype == SP_F32 ? 4 : 2;
    const long xs = (long)h * w_ * cin_p * esz;
    const long ds = (dy_pooled ? (long)(h / 2) * (w_ / 2) : (long)h * w_) * ld_dy * esz;
    for (int g = 0; g < 2; ++g) {
        const int n0 = g ? split : 0, ng = g ? n - split : split;
        const char* xg = reinterpret_cast<const char*>(x) + n0 * xs;
        const char* dg = reinterpret_cast<const char*>(dy) + n0 * ds;
        const int rc = dy_pooled ? sp_conv2d_wgrad_accum_pooled(xg, dg, g ? dw_b : dw_a, g ? dbias_b : dbias_a, workspace, workspace_floats, ng, h, w_,
                                                                cin_p, cout, ld_dy, ksize, dtype, stream)
                                 : sp_conv2d_wgrad_accum(xg, dg, g ? dw_b : dw_a, g ? dbias_b : dbias_a, workspace, workspace_floats, ng, h, w_, cin_p, cout,
                                                         ld_dy, ksize, dtype, stream);
        if (rc != SP_OK) return rc;
    }
    return SP_OK;
}

extern "C" int sp_conv2d_wgrad_workspace(int32_t n, int32_t h, int32_t w_, int32_t cin_p, int32_t cout, int32_t ksize,
                                         int32_t dtype, int64_t* floats_out) {
    SP_CHECK_ARG(floats_out && n > 0 && h > 0 && w_ > 0 && cin_p > 0 && cout > 0 && (ksize == 1 || ksize == 3), "sp_conv2d_wgrad_workspace: bad args");
    // scratch the row-walker kernel wants for its per-block partial tiles (bf16, 3x3, W % 32 == 0) ...
    int64_t need = (dtype == SP_BF16 && ksize == 3 && sp_tune(SP_TUNE_WGRAD_ROWS, 1)) ? (int64_t)sp_wgrad_rows_workspace(n, h, w_, cin_p, cout) : 0;
    if (dtype == SP_BF16 && ksize == 3 && cin_p == 8) {
        const int64_t s8 = sp_wgrad3x3_cin8_workspace(n, h, w_, cout, (cout + 7) & ~7);
        if (s8 > need) need = s8;
    }
    // ... the per-split slabs of the streaming 1x1 kernel ...
    if (dtype == SP_BF16 && ksize == 1) {
        const int64_t s1 = sp_wgrad1x1_workspace(n, h, w_, cin_p, cout, (cout + 7) & ~7);
        if (s1 > need) need = s1;
    }
    // ... and, in the deterministic mode, the per-split slabs of the per-tap kernels for every other shape
    if (sp_deterministic(dtype)) {
        const int64_t slab = dtype == SP_F32 ? pertap_slab_floats<float>(n, h, w_, cin_p, cout, ksize) : pertap_slab_floats<bf16>(n, h, w_, cin_p, cout, ksize);
        if (slab > need) need = slab;
    }
    *floats_out = need;
    return SP_OK;
}

extern "C" int sp_conv2d_wgrad_fused(const void* x, const void* dy, float* dw, float* dbias, const void* w_packed,
                                     float* dot, float* workspace, int64_t workspace_floats, int32_t n, int32_t h,
                                     int32_t w_, int32_t cin_p, int32_t cout, int32_t ld_dy, int32_t ksize,
                                     int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && dy && dw, "sp_conv2d_wgrad: null pointer");
    SP_CHECK_ARG(ksize == 1 || ksize == 3, "sp_conv2d_wgrad: ksize %d unsupported", ksize);
    SP_CHECK_ARG(dtype == SP_F32 || dtype == SP_BF16, "sp_conv2d_wgrad: bad dtype %d", dtype);
    const int e = dtype == SP_F32 ? 4 : 8;
    SP_CHECK_ARG(cin_p % e == 0 && ld_dy % e == 0, "sp_conv2d_wgrad: cin_p=%d and ld_dy=%d must be multiples of %d", cin_p, ld_dy, e);
    SP_CHECK_ARG(n > 0 && h > 0 && w_ > 0 && cout > 0 && cout <= ld_dy, "sp_conv2d_wgrad: bad dims");
    SP_CHECK_ARG(w_packed == nullptr || dot != nullptr, "sp_conv2d_wgrad: w_packed needs dot");
    // dot without w_packed: the slot is only zero-filled here (one fill for [dW | dot | dbias]) and sp_sn_backward
    // (dot_ready = 3) accumulates into it.
    if (!w_packed) { /* the kernels see dot == nullptr */ }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t n_dw = (size_t)cout * ksize * ksize * cin_p;
    hipError_t err;
    if (dot == dw + n_dw && dbias == dot + 1) {
        // caller laid the three outputs out back to back: one fill instead of three
        err = hipMemsetAsync(dw, 0, sizeof(float) * (n_dw + 1 + (size_t)cout), s);
    } else {
        err = hipMemsetAsync(dw, 0, sizeof(float) * n_dw, s);
        if (err == hipSuccess && dbias) err = hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)cout, s);
        if (err == hipSuccess && dot) err = hipMemsetAsync(dot, 0, sizeof(float), s);
    }
    if (err != hipSuccess) { sp_set_error("sp_conv2d_wgrad: memset failed: %s", hipGetErrorString(err)); return SP_ERR_LAUNCH; }
    return dtype == SP_F32 ? dispatch_wgrad<float>(x, dy, dw, n, h, w_, cin_p, cout, ld_dy, ksize, dbias, w_packed, w_packed ? dot : nullptr, workspace, workspace_floats, s)
                           : dispatch_wgrad<bf16>(x, dy, dw, n, h, w_, cin_p, cout, ld_dy, ksize, dbias, w_packed, w_packed ? dot : nullptr, workspace, workspace_floats, s);
}

extern "C" int sp_conv2d_wgrad_accum(const void* x, const void* dy, float* dw, float* dbias, float* workspace,
                                     int64_t workspace_floats, int32_t n, int32_t h, int32_t w_, int32_t cin_p, int32_t cout,
                                     int32_t ld_dy, int32_t ksize, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && dy && dw, "sp_conv2d_wgrad_accum: null pointer");
    SP_CHECK_ARG(ksize == 1 || ksize == 3, "sp_conv2d_wgrad_accum: ksize %d unsupported", ksize);
    SP_CHECK_ARG(dtype == SP_F32 || dtype == SP_BF16, "sp_conv2d_wgrad_accum: bad dtype %d", dtype);
    const int e = dtype == SP_F32 ? 4 : 8;
    SP_CHECK_ARG(cin_p % e == 0 && ld_dy % e == 0, "sp_conv2d_wgrad_accum: cin_p=%d and ld_dy=%d must be multiples of %d", cin_p, ld_dy, e);
    SP_CHECK_ARG(n > 0 && h > 0 && w_ > 0 && cout > 0 && cout <= ld_dy, "sp_conv2d_wgrad_accum: bad dims");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == SP_F32 ? dispatch_wgrad<float>(x, dy, dw, n, h, w_, cin_p, cout, ld_dy, ksize, dbias, nullptr, nullptr, workspace, workspace_floats, s)
                           : dispatch_wgrad<bf16>(x, dy, dw, n, h, w_, cin_p, cout, ld_dy, ksize, dbias, nullptr, nullptr, workspace, workspace_floats, s);
}

extern "C" int sp_conv2d_wgrad_accum_pooled(const void* x, const void* dy, float* dw, float* dbias, float* workspace,
                                            int64_t workspace_floats, int32_t n, int32_t h, int32_t w_, int32_t cin_p, int32_t cout,
                                            int32_t ld_dy, int32_t ksize, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(x && dy && dw, "sp_conv2d_wgrad_accum_pooled: null pointer");
    SP_CHECK_ARG(ksize == 3 && dtype == SP_BF16, "sp_conv2d_wgrad_accum_pooled: bf16 3x3 layers only");
    SP_CHECK_ARG(cin_p % 8 == 0 && ld_dy % 8 == 0, "sp_conv2d_wgrad_accum_pooled: cin_p=%d and ld_dy=%d must be multiples of 8", cin_p, ld_dy);
    SP_CHECK_ARG(n > 0 && h > 0 && w_ > 0 && cout > 0 && cout <= ld_dy && h % 2 == 0 && w_ % 2 == 0, "sp_conv2d_wgrad_accum_pooled: bad dims");
    const int rc = sp_wgrad_rows_launch(x, dy, dw, dbias, n, h, w_, cin_p, cout, ld_dy, workspace, workspace_floats, 1,
                                        reinterpret_cast<hipStream_t>(stream));
    SP_CHECK_ARG(rc != 1, "sp_conv2d_wgrad_accum_pooled: shape not covered by the row-walking kernel (w %% 32, h %% 2)");
    return rc;
}

extern "C" int sp_conv2d_wgrad(const void* x, const void* dy, float* dw, int32_t n, int32_t h, int32_t w_,
                               int32_t cin_p, int32_t cout, int32_t ld_dy, int32_t ksize, int32_t dtype,
                               sp_stream_t stream) {
    return sp_conv2d_wgrad_fused(x, dy, dw, nullptr, nullptr, nullptr, nullptr, 0, n, h, w_, cin_p, cout, ld_dy, ksize, dtype, stream);
}
