// Weight gradient of the 1x1 convolutions (bf16): dW[co][ci] = sum_p dY[p][co] * X[p][ci], a GEMM whose reduction runs over
// ALL pixels (20 k .. 1.3 M at batch 20) into at most 768 x 768 outputs - the operands are read once and the arithmetic is
// negligible, so the layer is a streaming reduction bound by HBM (64 -> 128 @64^2: 31 MB, ~4 us at 8 TB/s).  The per-tap
// kernel (conv_wgrad.hip) runs these layers as a thousand short-lived blocks (3-5 register-staged steps each) that meet through
// fp32 atomics: 17 .. 59 us.  Here
//   * a block owns a 64 co x 64 ci tile and a LONG pixel range, and streams it through a 4-slot LDS ring filled by LDS-DMA
//     (buffer_load ... lds, 16 B per lane, rows exactly as they lie in HBM; lanes past the tensor / the channel count read an
//     out-of-range offset = zeros): three 16 KB stages in flight per block, two blocks per CU;
//   * the transpose the MFMA operands need (k = pixels) happens on the LDS read (ds_read_b64_tr_b16; 128-byte rows, the 32-byte
//     column swizzled by (row >> 1) & 3 on the DMA's source side, as in the row-walking 3x3 kernel);
//   * workgroups are numbered so that the tiles sharing a pixel range run on the same XCD (one L2 fetch of the rows);
//   * the bias gradient (column sums of dY) rides on the MFMA pipe: one extra product per fragment row with an all-ones B;
//   * pixel splits store their partial tile into their own slab with plain stores and a second kernel sums the slabs in a
//     fixed order (16 slab groups per column block, LDS tree): no atomics, bit-identical run to run in every mode.
#include <cstdint>
#include <utility>
#include "common.h"

namespace {

constexpr int W1_TILE = 64 * 128;          // 64 pixel rows x 64 channels x 2 B
constexpr int W1_STAGE = 2 * W1_TILE;      // dY tile + X tile
constexpr int W1_NST = 4;
constexpr int W1_LDS = W1_NST * W1_STAGE;  // 64 KB

struct W1Args {
    const bf16* x;
    const bf16* dy;
    float* dw;
    float* dbias;
    float* slabs;          // [nsplit][n_dw] partial dW (nullptr: a single split adds to dW itself)
    float* bias_slabs;     // [nsplit][bias_ld]
    long M, n_dw;
    int CIN, COUT, LD_DY, tiles, ci_tiles, stages_per_split, nsplit, bias_ld;
};

template <int N> __device__ __forceinline__ void w1_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF> __device__ __forceinline__ void w1_tr(uint2& d, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ bf16x8_t w1_frag(const uint2& lo, const uint2& hi) {
    return __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
}

__global__ __launch_bounds__(256, 2) void wgrad1x1_stream_kernel(W1Args a) {
    extern __shared__ __attribute__((aligned(16))) char w1_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;
    // consecutive workgroups go to consecutive XCDs (8, each with its own L2): the tiles of one pixel split - which read the
    // same dY / X rows - are dealt to the SAME XCD, the splits round-robin over the XCDs
    // (layers with fewer than 8 splits keep the plain numbering: their tiles are what fills the chip)
    int split, tile;
    if (a.nsplit >= 8) {
        const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
        split = (q / a.tiles) * 8 + xcd;
        tile = q % a.tiles;
        if (split >= a.nsplit) return;
    } else {
        split = (int)blockIdx.x / a.tiles;
        tile = (int)blockIdx.x % a.tiles;
    }
    const int co0 = (tile / a.ci_tiles) * 64, ci0 = (tile % a.ci_tiles) * 64;
    const long total_stages = (a.M + 63) / 64;
    const long st0 = (long)split * a.stages_per_split;
    const int nk = (int)(total_stages - st0 < a.stages_per_split ? total_stages - st0 : a.stages_per_split);
    if (nk <= 0) return;                                                   // (the plan gives every split at least one stage)
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)w1_smem);

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, (int)(a.M * a.CIN * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.dy), 0, (int)(a.M * a.LD_DY * 2), 0x00020000);
    // ---- DMA side: lane l of a wave instruction writes pixel row (l >> 3), physical 16-byte slot (l & 7) of an 8-row group;
    // the 32-byte column is swizzled by key = (row >> 1) & 3 = (l >> 4) & 3 (8 | first row of the group)
    const int dpx = lane >> 3;
    const int dls = ((((lane & 7) >> 1) ^ ((dpx >> 1) & 3)) << 1) | (lane & 1);      // logical 16-byte slot this lane fetches
    const bool y_ok = co0 + dls * 8 < a.LD_DY;
    const bool x_ok = ci0 + dls * 8 < a.CIN;
    const unsigned y_lane = (unsigned)((dpx * a.LD_DY + co0 + dls * 8) * 2);
    const unsigned x_lane = (unsigned)((dpx * a.CIN + ci0 + dls * 8) * 2);
    auto issue = [&](int ks) {
        const long pb = (st0 + ks) * 64;
        char* sb = w1_smem + (ks & (W1_NST - 1)) * W1_STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int grp = wave + 4 * q;                                  // 8-row group of the 64-pixel stage
            const long row0 = pb + grp * 8;
            const bool in = row0 + dpx < a.M;
            const unsigned yo = (in && y_ok) ? (unsigned)(row0 * a.LD_DY * 2) + y_lane : OOB;
            const unsigned xo = (in && x_ok) ? (unsigned)(row0 * a.CIN * 2) + x_lane : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, (__attribute__((address_space(3))) void*)(sb + grp * 1024), 16, (int)yo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(sb + W1_TILE + grp * 1024), 16, (int)xo, 0, 0, 0);
        }
    };
    // ---- fragment reads: a 16-lane group reads [4 pixels][16 channels]; lane group g takes pixels {g*4 .. g*4+3} (+16 for the
    // second read), the same permutation of the reduction index for both operands
    const int i16 = lane & 15, g = lane >> 4;
    const int row1 = g * 4 + (i16 >> 2);
    const int key = (row1 >> 1) & 3;
    unsigned fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fa[i] = lds_base + (unsigned)(row1 * 128 + (((wa * 2 + i) ^ key) * 32) + (i16 & 3) * 8);
        fb[i] = lds_base + (unsigned)(W1_TILE + row1 * 128 + (((wb * 2 + i) ^ key) * 32) + (i16 & 3) * 8);
    }
    const bool do_bias = a.dbias != nullptr && ci0 == 0 && wb == 0;       // wave-uniform
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(SP_H16_ONE_PAIR, SP_H16_ONE_PAIR, SP_H16_ONE_PAIR, SP_H16_ONE_PAIR));
    f32x4_t acc[2][2], accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);
    for (int ks = 0; ks < nk; ++ks) {
        // own loads of stage ks landed (4 instructions per stage; up to two later stages stay in flight) ...
        if (ks + 2 < nk) w1_wait_vmcnt<8>();
        else if (ks + 1 < nk) w1_wait_vmcnt<4>();
        else w1_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                                      // ... everybody's did, and everybody left stage ks - 1
        if (ks + 3 < nk) issue(ks + 3);                                    // into the slot stage ks - 1 occupied
        const unsigned sb = (unsigned)((ks & (W1_NST - 1)) * W1_STAGE);
        uint2 al[2][2], ah[2][2], bl[2][2], bh[2][2];                      // [32-pixel half][fragment]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            w1_tr<0>(al[0][i], sb + fa[i]);     w1_tr<2048>(ah[0][i], sb + fa[i]);
            w1_tr<4096>(al[1][i], sb + fa[i]);  w1_tr<6144>(ah[1][i], sb + fa[i]);
            w1_tr<0>(bl[0][i], sb + fb[i]);     w1_tr<2048>(bh[0][i], sb + fb[i]);
            w1_tr<4096>(bl[1][i], sb + fb[i]);  w1_tr<6144>(bh[1][i], sb + fb[i]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8_t av = w1_frag(al[kk][i], ah[kk][i]);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w1_frag(bl[kk][j], bh[kk][j]), acc[i][j], 0, 0, 0);
                if (do_bias) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, ones, accb[i], 0, 0, 0);
            }
        }
    }

    // ---- partial tile -> this split's slab (plain stores), or straight onto dW when the layer has a single split
    const bool direct = a.slabs == nullptr;
    float* out = direct ? a.dw : a.slabs + (long)split * a.n_dw;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + (wa * 2 + i) * 16 + g * 4 + r;
            if (co >= a.COUT) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ci = ci0 + (wb * 2 + j) * 16 + i16;
                if (ci >= a.CIN) continue;
                const long o = (long)co * a.CIN + ci;
                if (direct) out[o] += acc[i][j][r]; else out[o] = acc[i][j][r];
            }
            if (do_bias && i16 == 0) {
                if (direct) a.dbias[co] += accb[i][r]; else a.bias_slabs[(long)split * a.bias_ld + co] = accb[i][r];
            }
        }
}


// ------------------------------------------------------------------------------------------------------------
// 3x3 weight gradient of the layers that read the 8-channel (padded RGB) images: dW[co][tap][8] over 1.3 M pixels, 168 MB of dY
// and 21 MB of X - a streaming reduction like the 1x1 layers (the row-walking kernel moves 64 pixels per barrier and spends its
// time in stage overhead: 99 us).  Same ring, same slabs; the X tile is built as an IM2COL row per pixel: the LDS-DMA lane that
// owns (pixel, tap) fetches the 16 bytes of the tap's neighbour pixel (out-of-image -> zeros), so a row holds 9 x 8 channels +
// 16 bytes of padding = 160 bytes, and two taps form one 16-column MFMA fragment: dW is a [Cout][72] matrix, its column index
// the fragment column.  160-byte rows put the 8 rows a transposed read touches on distinct banks without a swizzle.
// ------------------------------------------------------------------------------------------------------------
constexpr int C8_BPITCH = 160, C8_BTILE = 64 * C8_BPITCH, C8_STAGE = W1_TILE + C8_BTILE;      // 8 KB dY + 10 KB X rows
constexpr int C8_LDS = W1_NST * C8_STAGE;                                                     // 72 KB: two blocks per CU

__global__ __launch_bounds__(256, 2) void wgrad3x3_cin8_stream_kernel(W1Args a, int H, int W, int logw, int logh) {
    extern __shared__ __attribute__((aligned(16))) char w1_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;
    int split, tile;
    if (a.nsplit >= 8) {
        const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
        split = (q / a.tiles) * 8 + xcd;
        tile = q % a.tiles;
        if (split >= a.nsplit) return;
    } else {
        split = (int)blockIdx.x / a.tiles;
        tile = (int)blockIdx.x % a.tiles;
    }
    const int co0 = tile * 64;
    const long total_stages = (a.M + 63) / 64;
    const long st0 = (long)split * a.stages_per_split;
    const int nk = (int)(total_stages - st0 < a.stages_per_split ? total_stages - st0 : a.stages_per_split);
    if (nk <= 0) return;
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)w1_smem);

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, (int)(a.M * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.dy), 0, (int)(a.M * a.LD_DY * 2), 0x00020000);
    const int dpx = lane >> 3;
    const int dls = ((((lane & 7) >> 1) ^ ((dpx >> 1) & 3)) << 1) | (lane & 1);
    const bool y_ok = co0 + dls * 8 < a.LD_DY;
    const unsigned y_lane = (unsigned)((dpx * a.LD_DY + co0 + dls * 8) * 2);
    // im2col side: chunk q = instruction * 64 + lane of the stage's X tile <-> (pixel q / 10, tap q % 10; tap 9 = padding)
    const int nxi = wave < 2 ? 3 : 2;                                     // 10 instructions per stage: waves 0, 1 issue three
    int xq_px[3], xq_dr[3], xq_ds[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = (wave + 4 * j) * 64 + lane;
        const int px = q / 10, tap = q - px * 10;
        xq_px[j] = px;
        xq_dr[j] = tap < 9 ? tap / 3 - 1 : 99;                            // 99: padding chunk, never in the image
        xq_ds[j] = tap < 9 ? tap % 3 - 1 : 0;
    }
    auto issue = [&](int ks) {
        const long pb = (st0 + ks) * 64;
        char* sb = w1_smem + (ks & (W1_NST - 1)) * C8_STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int grp = wave + 4 * q;
            const long row0 = pb + grp * 8;
            const unsigned yo = (row0 + dpx < a.M && y_ok) ? (unsigned)(row0 * a.LD_DY * 2) + y_lane : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, (__attribute__((address_space(3))) void*)(sb + grp * 1024), 16, (int)yo, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j < nxi) {                                                 // wave-uniform
                const long P = pb + xq_px[j];
                const int xx = (int)(P & (W - 1)) + xq_ds[j];
                const int yy = (int)((P >> logw) & (H - 1)) + xq_dr[j];
                const bool ok = P < a.M && (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H;
                const unsigned xo = ok ? (unsigned)((P + xq_dr[j] * W + xq_ds[j]) * 16) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(sb + W1_TILE + (wave + 4 * j) * 1024), 16,
                                                         (int)xo, 0, 0, 0);
            }
        }
    };
    const int i16 = lane & 15, g = lane >> 4;
    const int row1 = g * 4 + (i16 >> 2);
    const int key = (row1 >> 1) & 3;
    unsigned fa[2], fb[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = lds_base + (unsigned)(row1 * 128 + (((wa * 2 + i) ^ key) * 32) + (i16 & 3) * 8);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int f = min(wb * 3 + j, 4);                                  // five 16-column fragments; wave half 1 repeats the last
        fb[j] = lds_base + (unsigned)(W1_TILE + row1 * C8_BPITCH + f * 32 + (i16 & 3) * 8);
    }
    const bool do_bias = a.dbias != nullptr && wb == 0;
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(SP_H16_ONE_PAIR, SP_H16_ONE_PAIR, SP_H16_ONE_PAIR, SP_H16_ONE_PAIR));
    f32x4_t acc[2][3], accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    // own requests of one stage: 2 (dY) + 3 or 2 (X)
    auto wait_landed = [&](int later) {                                    // `later` younger stages may stay in flight
        if (wave < 2) { if (later == 2) w1_wait_vmcnt<10>(); else if (later == 1) w1_wait_vmcnt<5>(); else w1_wait_vmcnt<0>(); }
        else { if (later == 2) w1_wait_vmcnt<8>(); else if (later == 1) w1_wait_vmcnt<4>(); else w1_wait_vmcnt<0>(); }
    };

    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);
    for (int ks = 0; ks < nk; ++ks) {
        wait_landed(ks + 2 < nk ? 2 : (ks + 1 < nk ? 1 : 0));
        __builtin_amdgcn_s_barrier();
        if (ks + 3 < nk) issue(ks + 3);
        const unsigned sb = (unsigned)((ks & (W1_NST - 1)) * C8_STAGE);
        uint2 al[2][2], ah[2][2], bl[2][3], bh[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            w1_tr<0>(al[0][i], sb + fa[i]);     w1_tr<2048>(ah[0][i], sb + fa[i]);
            w1_tr<4096>(al[1][i], sb + fa[i]);  w1_tr<6144>(ah[1][i], sb + fa[i]);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            w1_tr<0>(bl[0][j], sb + fb[j]);                  w1_tr<16 * C8_BPITCH>(bh[0][j], sb + fb[j]);
            w1_tr<32 * C8_BPITCH>(bl[1][j], sb + fb[j]);     w1_tr<48 * C8_BPITCH>(bh[1][j], sb + fb[j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8_t av = w1_frag(al[kk][i], ah[kk][i]);
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, w1_frag(bl[kk][j], bh[kk][j]), acc[i][j], 0, 0, 0);
                if (do_bias) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, ones, accb[i], 0, 0, 0);
            }
        }
    }

    const bool direct = a.slabs == nullptr;
    float* out = direct ? a.dw : a.slabs + (long)split * a.n_dw;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + (wa * 2 + i) * 16 + g * 4 + r;
            if (co >= a.COUT) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int f = wb * 3 + j;
                const int n = f * 16 + i16;                                // = tap * 8 + ci
                if (f > 4 || n >= 72) continue;
                const long o = (long)co * 72 + n;
                if (direct) out[o] += acc[i][j][r]; else out[o] = acc[i][j][r];
            }
            if (do_bias && i16 == 0) {
                if (direct) a.dbias[co] += accb[i][r]; else a.bias_slabs[(long)split * a.bias_ld + co] = accb[i][r];
            }
        }
}

// out[e] += sum_s slab[s][e] for the dW columns and then the bias columns.  A block owns 16 float4 columns; its 16 thread rows
// take the slabs s = g, g + 16, ... (independent loads) and meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void wgrad1x1_reduce_kernel(const float* __restrict__ slabs, int nsplit, long n_dw, float* __restrict__ dw,
                                                              const float* __restrict__ bias_slabs, int bias_ld, int cout,
                                                              float* __restrict__ dbias) {
    __shared__ float4 red[16][16];
    const int c16 = threadIdx.x & 15, g = threadIdx.x >> 4;
    const long cols4 = n_dw / 4;
    const long bias4 = bias_slabs != nullptr ? bias_ld / 4 : 0;
    const long c = (long)blockIdx.x * 16 + c16;
    const float* src = nullptr;
    long stride = 0;
    if (c < cols4) { src = slabs + c * 4; stride = n_dw; }
    else if (c < cols4 + bias4) { src = bias_slabs + (c - cols4) * 4; stride = bias_ld; }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (src != nullptr) {
#pragma unroll 4
        for (int k = g; k < nsplit; k += 16) {
            const float4 v = *reinterpret_cast<const float4*>(src + (long)k * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[g][c16] = s;
    __syncthreads();
    if (g != 0 || src == nullptr) return;
    float4 t = red[0][c16];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 v = red[k][c16]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    if (c < cols4) {
        float4 d = *reinterpret_cast<float4*>(dw + c * 4);
        d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
        *reinterpret_cast<float4*>(dw + c * 4) = d;
    } else {
        const int co = (int)(c - cols4) * 4;
        const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (co + r < cout) dbias[co + r] += v[r];
    }
}

struct W1Plan { int tiles, ci_tiles, nsplit, stages_per_split; bool ok; };

W1Plan w1_plan(int n, int h, int w, int cin, int cout, int ld_dy) {
    W1Plan p;
    p.ok = false;
    const int target = sp_tune(SP_TUNE_WGRAD1X1, 256);
    const long M = (long)n * h * w;
    p.ci_tiles = (cin + 63) / 64;
    p.tiles = p.ci_tiles * ((cout + 63) / 64);
    p.nsplit = 1;
    p.stages_per_split = 1;
    if (target <= 0 || cin % 8 != 0 || ld_dy % 8 != 0) return p;
    if (M * cin * 2 >= (1L << 31) || M * ld_dy * 2 >= (1L << 31)) return p;           // 32-bit buffer offsets
    const long stages = (M + 63) / 64;
    // ~target blocks (one per CU measured best: 128 / 192 / 256 / 384 / 512 -> 0.250 / 0.211 / 0.189 / 0.201 / 0.197 ms over the step's
    // fourteen shapes - more splits mean more slab traffic), at least four 64-pixel stages per split
    long nsplit = (target + p.tiles - 1) / p.tiles;
    if (nsplit > stages / 4) nsplit = stages / 4;
    if (nsplit < 1) nsplit = 1;
    const long sps = (stages + nsplit - 1) / nsplit;
    p.stages_per_split = (int)sps;
    p.nsplit = (int)((stages + sps - 1) / sps);
    p.ok = true;
    return p;
}

}  // namespace

long sp_wgrad1x1_workspace(int n, int h, int w, int cin, int cout, int ld_dy) {
    const W1Plan p = w1_plan(n, h, w, cin, cout, ld_dy);
    if (!p.ok || p.nsplit <= 1) return 0;
    return (long)p.nsplit * ((long)cout * cin + ((cout + 3) & ~3));
}

// SP_OK after launching, 1 if the layer is not covered (the caller falls back to the per-tap kernel)
// Two reduce passes over the slab sub-ranges of the two groups of a two-group batch (sp_conv2d_wgrad_accum_pair): split k of the
// streaming kernels covers the pixels [k, k + 1) * 64 * stages_per_split, so when the group boundary is a multiple of that every
// slab belongs to one group.
static void w1_reduce_groups(const W1Args& a, int nsplit, int kb, float* dw_a, float* dbias_a, float* dw_b, float* dbias_b, int cout, hipStream_t s) {
    const long cols = a.n_dw / 4 + (a.bias_slabs != nullptr ? a.bias_ld / 4 : 0);
    for (int g = 0; g < 2; ++g) {
        const int k0 = g ? kb : 0, k1 = g ? nsplit : kb;
        if (spq_push_reduce(a.slabs + (long)k0 * a.n_dw, k1 - k0, a.n_dw, g ? dw_b : dw_a, a.bias_slabs != nullptr ? a.bias_slabs + (long)k0 * a.bias_ld : nullptr,
                            a.bias_ld, cout, g ? dbias_b : dbias_a)) continue;          // queued (reduce_queue.hip)
        hipLaunchKernelGGL(wgrad1x1_reduce_kernel, dim3((unsigned)((cols + 15) / 16)), dim3(256), 0, s, a.slabs + (long)k0 * a.n_dw, k1 - k0, a.n_dw,
                           g ? dw_b : dw_a, a.bias_slabs != nullptr ? a.bias_slabs + (long)k0 * a.bias_ld : nullptr, a.bias_ld, cout, g ? dbias_b : dbias_a);
    }
}

static int w1_launch_impl(const void* x, const void* dy, float* dw, float* dbias, float* dw_b, float* dbias_b, long split_pixels, int n, int h, int w,
                          int cin, int cout, int ld_dy, float* ws, long ws_floats, hipStream_t s);

int sp_wgrad1x1_launch(const void* x, const void* dy, float* dw, float* dbias, int n, int h, int w, int cin, int cout, int ld_dy,
                       float* ws, long ws_floats, hipStream_t s) {
    return w1_launch_impl(x, dy, dw, dbias, nullptr, nullptr, 0, n, h, w, cin, cout, ld_dy, ws, ws_floats, s);
}

// the two-group form: 1 unless the layer runs with slabs and the group boundary falls between two splits
int sp_wgrad1x1_launch_pair(const void* x, const void* dy, float* dw_a, float* dbias_a, float* dw_b, float* dbias_b, int n, int split, int h, int w,
                            int cin, int cout, int ld_dy, float* ws, long ws_floats, hipStream_t s) {
    if (split <= 0 || split >= n || (dbias_a == nullptr) != (dbias_b == nullptr)) return 1;
    return w1_launch_impl(x, dy, dw_a, dbias_a, dw_b, dbias_b, (long)split * h * w, n, h, w, cin, cout, ld_dy, ws, ws_floats, s);
}

static int w1_launch_impl(const void* x, const void* dy, float* dw, float* dbias, float* dw_b, float* dbias_b, long split_pixels, int n, int h, int w,
                          int cin, int cout, int ld_dy, float* ws, long ws_floats, hipStream_t s) {
    const W1Plan p = w1_plan(n, h, w, cin, cout, ld_dy);
    if (!p.ok) return 1;
    const long px_per_split = 64L * p.stages_per_split;
    if (dw_b != nullptr && (p.nsplit <= 1 || split_pixels % px_per_split != 0)) return 1;
    W1Args a;
    a.x = reinterpret_cast<const bf16*>(x);
    a.dy = reinterpret_cast<const bf16*>(dy);
    a.dw = dw;
    a.dbias = dbias;
    a.M = (long)n * h * w;
    a.n_dw = (long)cout * cin;
    a.CIN = cin; a.COUT = cout; a.LD_DY = ld_dy;
    a.ci_tiles = p.ci_tiles;
    a.tiles = p.tiles;
    a.stages_per_split = p.stages_per_split;
    a.nsplit = p.nsplit;
    a.bias_ld = (cout + 3) & ~3;
    a.slabs = nullptr;
    a.bias_slabs = nullptr;
    if (p.nsplit > 1) {
        if (ws == nullptr || ws_floats < (long)p.nsplit * (a.n_dw + a.bias_ld)) return 1;
        a.slabs = ws;
        a.bias_slabs = dbias != nullptr ? ws + (long)p.nsplit * a.n_dw : nullptr;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad1x1_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W1_LDS);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", W1_LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        attr_set = true;
    }
    sp_note_route(dw_b != nullptr ? "wgrad1x1_stream (two groups) + 2 x reduce" : "wgrad1x1_stream + reduce");
    hipLaunchKernelGGL(wgrad1x1_stream_kernel, dim3((unsigned)(p.tiles * (p.nsplit >= 8 ? ((p.nsplit + 7) / 8) * 8 : p.nsplit))), dim3(256), W1_LDS, s, a);
    SP_LAUNCH_CHECK();
    if (dw_b != nullptr) {
        w1_reduce_groups(a, p.nsplit, (int)(split_pixels / px_per_split), dw, dbias, dw_b, dbias_b, cout, s);
        SP_LAUNCH_CHECK();
        return SP_OK;
    }
    if (p.nsplit > 1 && !spq_push_reduce(a.slabs, p.nsplit, a.n_dw, dw, a.bias_slabs, a.bias_ld, cout, dbias)) {
        const long cols = a.n_dw / 4 + (a.bias_slabs != nullptr ? a.bias_ld / 4 : 0);
        hipLaunchKernelGGL(wgrad1x1_reduce_kernel, dim3((unsigned)((cols + 15) / 16)), dim3(256), 0, s, a.slabs, p.nsplit, a.n_dw, dw, a.bias_slabs,
                           a.bias_ld, cout, dbias);
        SP_LAUNCH_CHECK();
    }
    return SP_OK;
}

// ---- 3x3, 8-channel input (bf16): same contract
static bool c8_plan(int n, int h, int w, int cout, int ld_dy, W1Plan& p) {
    const int target = sp_tune(SP_TUNE_WGRAD1X1, 256);
    const long M = (long)n * h * w;
    if (target <= 0 || (h & (h - 1)) != 0 || (w & (w - 1)) != 0 || ld_dy % 8 != 0) return false;
    if (M * 16 >= (1L << 31) || M * ld_dy * 2 >= (1L << 31) || M < 16384) return false;     // small maps: the row-walker is fine
    p.ci_tiles = 1;
    p.tiles = (cout + 63) / 64;
    const long stages = (M + 63) / 64;
    long nsplit = (target + p.tiles - 1) / p.tiles;
    if (nsplit > stages / 4) nsplit = stages / 4;
    if (nsplit < 1) nsplit = 1;
    const long sps = (stages + nsplit - 1) / nsplit;
    p.stages_per_split = (int)sps;
    p.nsplit = (int)((stages + sps - 1) / sps);
    p.ok = true;
    return true;
}

long sp_wgrad3x3_cin8_workspace(int n, int h, int w, int cout, int ld_dy) {
    W1Plan p;
    if (!c8_plan(n, h, w, cout, ld_dy, p) || p.nsplit <= 1) return 0;
    return (long)p.nsplit * ((long)cout * 72 + ((cout + 3) & ~3));
}

static int c8_launch_impl(const void* x, const void* dy, float* dw, float* dbias, float* dw_b, float* dbias_b, long split_pixels, int n, int h, int w,
                          int cout, int ld_dy, float* ws, long ws_floats, hipStream_t s);

int sp_wgrad3x3_cin8_launch(const void* x, const void* dy, float* dw, float* dbias, int n, int h, int w, int cout, int ld_dy, float* ws,
                            long ws_floats, hipStream_t s) {
    return c8_launch_impl(x, dy, dw, dbias, nullptr, nullptr, 0, n, h, w, cout, ld_dy, ws, ws_floats, s);
}

int sp_wgrad3x3_cin8_launch_pair(const void* x, const void* dy, float* dw_a, float* dbias_a, float* dw_b, float* dbias_b, int n, int split, int h,
                                 int w, int cout, int ld_dy, float* ws, long ws_floats, hipStream_t s) {
    if (split <= 0 || split >= n || (dbias_a == nullptr) != (dbias_b == nullptr)) return 1;
    return c8_launch_impl(x, dy, dw_a, dbias_a, dw_b, dbias_b, (long)split * h * w, n, h, w, cout, ld_dy, ws, ws_floats, s);
}

static int c8_launch_impl(const void* x, const void* dy, float* dw, float* dbias, float* dw_b, float* dbias_b, long split_pixels, int n, int h, int w,
                          int cout, int ld_dy, float* ws, long ws_floats, hipStream_t s) {
    W1Plan p;
    if (!c8_plan(n, h, w, cout, ld_dy, p)) return 1;
    const long px_per_split = 64L * p.stages_per_split;
    if (dw_b != nullptr && (p.nsplit <= 1 || split_pixels % px_per_split != 0)) return 1;
    W1Args a;
    a.x = reinterpret_cast<const bf16*>(x);
    a.dy = reinterpret_cast<const bf16*>(dy);
    a.dw = dw;
    a.dbias = dbias;
    a.M = (long)n * h * w;
    a.n_dw = (long)cout * 72;
    a.CIN = 8; a.COUT = cout; a.LD_DY = ld_dy;
    a.ci_tiles = 1;
    a.tiles = p.tiles;
    a.stages_per_split = p.stages_per_split;
    a.nsplit = p.nsplit;
    a.bias_ld = (cout + 3) & ~3;
    a.slabs = nullptr;
    a.bias_slabs = nullptr;
    if (p.nsplit > 1) {
        if (ws == nullptr || ws_floats < (long)p.nsplit * (a.n_dw + a.bias_ld)) return 1;
        a.slabs = ws;
        a.bias_slabs = dbias != nullptr ? ws + (long)p.nsplit * a.n_dw : nullptr;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_cin8_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C8_LDS);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", C8_LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        attr_set = true;
    }
    int logw = 0, logh = 0;
    while ((1 << logw) < w) ++logw;
    while ((1 << logh) < h) ++logh;
    sp_note_route(dw_b != nullptr ? "wgrad3x3_cin8_stream (two groups) + 2 x reduce" : "wgrad3x3_cin8_stream + reduce");
    hipLaunchKernelGGL(wgrad3x3_cin8_stream_kernel, dim3((unsigned)(p.tiles * (p.nsplit >= 8 ? ((p.nsplit + 7) / 8) * 8 : p.nsplit))), dim3(256), C8_LDS, s, a,
                       h, w, logw, logh);
    SP_LAUNCH_CHECK();
    if (dw_b != nullptr) {
        w1_reduce_groups(a, p.nsplit, (int)(split_pixels / px_per_split), dw, dbias, dw_b, dbias_b, cout, s);
        SP_LAUNCH_CHECK();
        return SP_OK;
    }
    if (p.nsplit > 1 && !spq_push_reduce(a.slabs, p.nsplit, a.n_dw, dw, a.bias_slabs, a.bias_ld, cout, dbias)) {
        const long cols = a.n_dw / 4 + (a.bias_slabs != nullptr ? a.bias_ld / 4 : 0);
        hipLaunchKernelGGL(wgrad1x1_reduce_kernel, dim3((unsigned)((cols + 15) / 16)), dim3(256), 0, s, a.slabs, p.nsplit, a.n_dw, dw, a.bias_slabs,
                           a.bias_ld, cout, dbias);
        SP_LAUNCH_CHECK();
    }
    return SP_OK;
}
