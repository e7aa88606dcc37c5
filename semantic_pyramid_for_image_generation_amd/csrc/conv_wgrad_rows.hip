// 3x3 weight gradient, "row walker" (bf16): dW[co][tap][ci] = sum_p dY[p][co] * X[p + shift(tap)][ci].
//
// The per-tap kernel in conv_wgrad.hip re-reads both operand tiles for each of the nine taps (64 flop per L2 byte).
// Here a block owns a 64 co x 64 ci x 9 taps tile of dW in registers (a wave: 64 co x 16 ci x 9 = 36 accumulator
// fragments) and WALKS DOWN a 32-pixel-wide column strip of an image: per image row it needs one new dY row segment
// (32 px x 64 co) and one new X row segment (34 px x 64 ci, the halo columns included); the X rows y-1, y, y+1 of the
// three tap rows come from a rolling ring in LDS, and the tap columns are just the same LDS row read at a pixel
// offset of 0 / 1 / 2.  8.3 KB of L2 traffic per 2.4 MFLOP (284 flop/B).
//   * staging: LDS-DMA (buffer_load ... lds), zero fill through out-of-range offsets; dY ring of 6 and X ring of 10
//     row slots, requested TWO stages (of two image rows each) ahead, counted vmcnt + one barrier per stage;
//   * the reduction index of the GEMM is the pixel, the slow index of both NHWC operands: fragments are read with
//     ds_read_b64_tr_b16 straight from the [pixel][channel] rows.  Rows are 128 B, so the four pixel rows a lane
//     group reads would hit the same banks: the 32-byte channel columns are XOR-swizzled by (pixel & 3), applied on
//     the DMA source side;
//   * split-K: a block loops over (image, strip, row chunk) units accumulating in registers and merges into dW with
//     fp32 atomics ONCE at the end; the bias gradient rides along as one extra MFMA per dY fragment against a
//     fragment of ones (blocks of the first ci tile only).
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

constexpr int WR_R = 2, WR_PD = 2;
constexpr int WR_XPX = 40;                       // pixels per X row slot (34 used: x0-1 .. x0+32)
constexpr int WR_XSLOT = WR_XPX * 128;
constexpr int WR_YSLOT = 32 * 128;
constexpr int WR_NSX = 10, WR_NSY = 6;
constexpr int WR_XBYTES = WR_NSX * WR_XSLOT;
constexpr int WR_LDS = WR_XBYTES + WR_NSY * WR_YSLOT;

struct WrArgs {
    const bf16* x;
    const bf16* dy;
    float* dw;
    float* dbias;
    float* slabs;          // per-block partial tiles [9][64][64] (nullptr: merge with atomics)
    float* bias_part;      // per-block partial bias sums [co tile][block][64] (nullptr: atomics on dbias - hundreds of blocks on
                           // the same 64 addresses serialise: ~50 us of a 64 -> 64 @256^2 launch)
    int N, H, W, CIN, COUT, LD_DY;
    int rows_per_unit, units, ci_tiles;
    int thin_mode;         // 0: off, 1: thin ci tiles split the rows over the waves (diagnostic 2: merge only, 3: split only)
    int dy_up2;            // dy is stored at half resolution and stands for 1/4 x its nearest-neighbour x2 expansion
};

template <int... I, typename F>
__device__ __forceinline__ void wr_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void wr_static_for(F&& f) { wr_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

template <int N> __device__ __forceinline__ void wr_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wr_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int OFF> __device__ __forceinline__ void wr_tr(uint2& d, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ bf16x8_t wr_frag(const uint2& lo, const uint2& hi) {
    return __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
}

// NW = 0: maps at least 32 pixels wide (column strips of an image).  NW = 16 / 8: NARROW maps - a "strip" is 32 / NW images side
// by side: the dY row slot holds their rows back to back, the X row slot their halo rows (NW + 2 pixels each), so pixel p of
// the strip finds its X neighbourhood at LDS row p + 2 * (p / NW) + tap column: the fragment reads only get a lane-dependent
// base and their own address for the second half (the swizzle key of row + 16 + 2 * (16 / NW) differs from that of row).
template <int NW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_rows_kernel(WrArgs a) {
    constexpr int G = NW ? 32 / NW : 1;
    extern __shared__ __attribute__((aligned(16))) char wr_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, CIN = a.CIN;
    const int co0 = (blockIdx.y / a.ci_tiles) * 64, ci0 = (blockIdx.y % a.ci_tiles) * 64;
    const int RU = a.rows_per_unit, SPU = RU / WR_R;
    const int strips = NW ? 1 : W / 32, chunks = H / RU;
    const int my_units = (a.units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_stages = my_units * SPU;
    if (total_stages <= 0) return;
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)wr_smem);

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, a.N * H * W * CIN * 2, 0x00020000);
    const int up = a.dy_up2 ? 1 : 0;
    const int HY = H >> up, WY = W >> up;
    const float oscale = up ? 0.25f : 1.f;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.dy), 0, a.N * HY * WY * a.LD_DY * 2, 0x00020000);

    // ---- DMA side: lane l of a wave-instruction writes LDS pixel row (l >> 3), physical 16-byte slot (l & 7); the 32-byte
    // column index is swizzled by key = (pixel >> 1) & 3, so the lane fetches logical slot (((l & 7) >> 1) ^ key) * 2 + (l & 1).
    // One instruction = 8 pixels, and 8 | pixel base, so key = (l >> 4) & 3.  (ds_read_b64_tr_b16 serves 32 lanes = 8 pixel
    // rows per pass over 64 banks: rows r and r + 2 share a 128-byte half and must differ in the column; the first
    // version keyed on pixel & 3 and measured 2 conflict cycles per read, SQ_LDS_BANK_CONFLICT.)
    const int dpx = lane >> 3;
    const int dls = ((((lane & 7) >> 1) ^ ((dpx >> 1) & 3)) << 1) | (lane & 1);   // logical 16-byte slot
    const bool x_ch_ok = ci0 + dls * 8 < CIN;
    const bool y_ch_ok = co0 + dls * 8 < a.LD_DY;
    auto unit_coords = [&](int ui, int& n, int& x0, int& y0) {
        const int u = (int)blockIdx.x + ui * (int)gridDim.x;
        const int ch = u % chunks;
        const int t = u / chunks;
        x0 = (t % strips) * 32;
        n = (t / strips) * G;                                                    // narrow: first image of the group
        y0 = ch * RU;
    };
    // X row `urow` (0 .. RU+1 <-> image row y0 - 1 + urow) of unit ui -> ring slot `slot`; returns instructions issued
    auto issue_x_row = [&](int n, int x0, int y, int slot) {
        int cnt = 0;
        char* dst = wr_smem + slot * WR_XSLOT;
        const bool row_ok = (unsigned)y < (unsigned)H;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if ((i & 3) == wave) {                                               // wave-uniform
                const int px = i * 8 + dpx;
                int img = n, xx = x0 - 1 + px;
                bool ok = row_ok && x_ch_ok && px < 34;
                if constexpr (NW != 0) {
                    const int j = px / (NW + 2);                                 // image of the group, column inside its halo row
                    img = n + j;
                    xx = px - j * (NW + 2) - 1;
                    ok = row_ok && x_ch_ok && j < G && img < a.N;
                }
                ok = ok && (unsigned)xx < (unsigned)W;
                const unsigned off = ok ? (unsigned)((((img * H + y) * W + xx) * CIN + ci0 + dls * 8) * 2) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, (int)off, 0, 0, 0);
                ++cnt;
            }
        }
        return cnt;
    };
    auto issue_y_row = [&](int n, int x0, int y, int slot) {
        char* dst = wr_smem + WR_XBYTES + slot * WR_YSLOT;
        int img = n, xx = x0 + wave * 8 + dpx;
        bool ok = y_ch_ok;
        if constexpr (NW != 0) {
            img = n + xx / NW;
            xx &= NW - 1;
            ok = ok && img < a.N;
        }
        const unsigned off = ok ? (unsigned)((((img * HY + (y >> up)) * WY + (xx >> up)) * a.LD_DY + co0 + dls * 8) * 2) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, (__attribute__((address_space(3))) void*)(dst + wave * 1024), 16, (int)off, 0, 0, 0);
        return 1;
    };
    // requests everything the NEXT not-yet-requested stage needs beyond what earlier stages already requested.  All ring /
    // unit bookkeeping is incremental (wave-uniform counters with a compare-and-wrap): the first version recomputed it with
    // integer divisions per stage and spent 23 % of its issue slots on SALU.
    int l_ui = 0, l_k = 0, l_n = 0, l_x0 = 0, l_y0 = 0, l_xslot = 0, l_yslot = 0;
    auto issue_stage = [&]() {
        if (l_k == 0) unit_coords(l_ui, l_n, l_x0, l_y0);
        int cnt = 0;
        const int first = l_k == 0 ? 0 : l_k * WR_R + 2, nrows = l_k == 0 ? WR_R + 2 : WR_R;
        for (int j = 0; j < nrows; ++j) {
            cnt += issue_x_row(l_n, l_x0, l_y0 - 1 + first + j, l_xslot);
            l_xslot = l_xslot + 1 == WR_NSX ? 0 : l_xslot + 1;
        }
#pragma unroll
        for (int r = 0; r < WR_R; ++r) {
            // pooled dY (up): the two rows of a stage are the same pooled row - fetched once, read twice (slot of r = 0)
            if (!(up && (r & 1))) cnt += issue_y_row(l_n, l_x0, l_y0 + l_k * WR_R + r, l_yslot);
            l_yslot = l_yslot + 1 == WR_NSY ? 0 : l_yslot + 1;
        }
        if (++l_k == SPU) { l_k = 0; ++l_ui; }
        return cnt;
    };
    auto wait_dyn = [&](int n) {
        switch (n) {
            case 0: wr_wait_vmcnt<0>(); break;
            case 1: wr_wait_vmcnt<1>(); break;
            case 2: wr_wait_vmcnt<2>(); break;
            case 3: wr_wait_vmcnt<3>(); break;
            case 4: wr_wait_vmcnt<4>(); break;
            case 5: wr_wait_vmcnt<5>(); break;
            case 6: wr_wait_vmcnt<6>(); break;
            case 7: wr_wait_vmcnt<7>(); break;
            case 8: wr_wait_vmcnt<8>(); break;
            case 9: wr_wait_vmcnt<9>(); break;
            default: wr_wait_vmcnt<10>(); break;
        }
    };

    // ---- fragment read addresses (bytes inside a row slot).  Lane (g, i16) reads pixel row g*4 + (i16 >> 2) [+16 for the
    // second half of the k-step], 8 bytes at (i16 & 3) * 8 inside the 32-byte channel column; the column is XOR-ed with
    // (pixel & 3).
    const int i16 = lane & 15, g = lane >> 4;
    const int prow = g * 4 + (i16 >> 2);
    unsigned a_off[4], b_off[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[i] = (unsigned)(prow * 128 + ((i ^ ((prow >> 1) & 3)) << 5) + (i16 & 3) * 8);
    // THIN ci tile (at most 16 input channels left: the 3 -> 8 padded first layer, the +8 channels of the concatenated
    // pyramid masks): three of the four 16-channel wave slices would multiply zeros.  The waves then share slice 0 and
    // split the pixel ROWS instead (row j of a stage pair goes to wave j & 3); their partial tiles meet in LDS at the end.
    const bool thin = a.thin_mode != 0 && CIN - ci0 <= 16;
    const int bcol = thin ? 0 : wave;
    const int prow_x = NW ? prow + 2 * (prow / (NW ? NW : 1)) : prow;
    [[maybe_unused]] unsigned b_offh[3];
#pragma unroll
    for (int ds = 0; ds < 3; ++ds) {
        b_off[ds] = (unsigned)((prow_x + ds) * 128 + ((bcol ^ (((prow_x + ds) >> 1) & 3)) << 5) + (i16 & 3) * 8);
        const int rh = prow_x + ds + 16 + (NW ? 2 * (16 / (NW ? NW : 1)) : 0);
        b_offh[ds] = (unsigned)(rh * 128 + ((bcol ^ ((rh >> 1) & 3)) << 5) + (i16 & 3) * 8);
    }
    auto tr_bhi = [&](uint2& d, unsigned base, int ds) {
        if constexpr (NW != 0) wr_tr<0>(d, base + b_offh[ds]); else wr_tr<2048>(d, base + b_off[ds]);
    };

    f32x4_t acc[9][4], accb[4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = a.dbias != nullptr && ci0 == 0 && (thin || wave == 0);    // thin: every wave sums the rows it owns
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(SP_H16_ONE_PAIR, SP_H16_ONE_PAIR, SP_H16_ONE_PAIR, SP_H16_ONE_PAIR));

    int n_last = 0;
    issue_stage();
    if (total_stages > 1) n_last = issue_stage();
    int c_k = 0, c_xbase = 0, c_yslot = 0;                 // compute side: stage inside the unit, ring slots of its first X / dY row
    for (int gs = 0; gs < total_stages; ++gs) {
        wait_dyn(gs + 1 < total_stages ? n_last : 0);
        __builtin_amdgcn_s_barrier();                      // stage gs landed for everyone; everyone left stage gs - 1
        n_last = gs + WR_PD < total_stages ? issue_stage() : 0;
#pragma unroll
        for (int r = 0; r < WR_R; ++r) {
            if (thin && a.thin_mode != 2 && ((((gs & 1) * WR_R + r) & 3) != wave)) continue;      // wave-uniform
            int ys = c_yslot + (up ? (r & ~1) : r);
            if (ys >= WR_NSY) ys -= WR_NSY;
            const unsigned ab = lds_base + WR_XBYTES + (unsigned)(ys * WR_YSLOT);
            unsigned bb[3];
#pragma unroll
            for (int dr = 0; dr < 3; ++dr) {
                int xs = c_xbase + r + dr;
                if (xs >= WR_NSX) xs -= WR_NSX;
                bb[dr] = lds_base + (unsigned)(xs * WR_XSLOT);
            }
            uint2 alo[4], ahi[4], blo[9], bhi[9];
#pragma unroll
            for (int i = 0; i < 4; ++i) { wr_tr<0>(alo[i], ab + a_off[i]); wr_tr<2048>(ahi[i], ab + a_off[i]); }
            wr_tr<0>(blo[0], bb[0] + b_off[0]); tr_bhi(bhi[0], bb[0], 0);
            wr_tr<0>(blo[1], bb[0] + b_off[1]); tr_bhi(bhi[1], bb[0], 1);
            wr_static_for<9>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (t + 2 < 9) {
                    constexpr int t2 = t + 2;
                    wr_tr<0>(blo[t2], bb[t2 / 3] + b_off[t2 % 3]);
                    tr_bhi(bhi[t2], bb[t2 / 3], t2 % 3);
                    wr_wait_lgkm<4>();
                } else if constexpr (t + 1 < 9) {
                    wr_wait_lgkm<2>();
                } else {
                    wr_wait_lgkm<0>();
                }
                const bf16x8_t bf = wr_frag(blo[t], bhi[t]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr_frag(alo[i], ahi[i]), bf, acc[t][i], 0, 0, 0);
            });
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr_frag(alo[i], ahi[i]), ones, accb[i], 0, 0, 0);
            }
        }
        c_yslot += WR_R;
        if (c_yslot >= WR_NSY) c_yslot -= WR_NSY;
        c_xbase += (++c_k == SPU) ? WR_R + 2 : WR_R;       // a new unit starts two (halo) rows further on in the ring
        if (c_k == SPU) c_k = 0;
        if (c_xbase >= WR_NSX) c_xbase -= WR_NSX;
    }

    if (thin && a.thin_mode != 3) {
        // the four partial tiles [tap][co][16 ci] are summed in LDS (the rings are idle after the barrier), one wave at a time
        constexpr int TPT = 17;
        float* tile = reinterpret_cast<float*>(wr_smem);
        __syncthreads();
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* q = tile + (t * 64 + i * 16 + g * 4 + r) * TPT + i16;
                            *q = w == 0 ? acc[t][i][r] : *q + acc[t][i][r];
                        }
            }
            __syncthreads();
        }
        float* slab = a.slabs != nullptr ? a.slabs + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (9 * 64 * 64) : nullptr;
        for (int e = tid; e < 9 * 64 * 16; e += 256) {
            const int row = e >> 4, c = e & 15;                    // row = tap * 64 + co (tile-local)
            const float v = tile[row * TPT + c] * oscale;
            if (slab != nullptr) {
                slab[row * 64 + c] = v;
            } else {
                const int t = row >> 6, co = co0 + (row & 63), ci = ci0 + c;
                if (co < a.COUT && ci < CIN) atomicAdd(a.dw + ((long)co * 9 + t) * CIN + ci, v);
            }
        }
        if (a.dbias != nullptr && ci0 == 0) {
            // the four waves' bias partials meet in LDS too (behind the tile), then wave 0 hands the block's sum over
            float* bsum = tile + 9 * 64 * TPT;
            if (i16 == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bsum[wave * 64 + i * 16 + g * 4 + r] = accb[i][r];
            }
            __syncthreads();
            if (tid < 64) {
                const float v = (bsum[tid] + bsum[64 + tid] + bsum[128 + tid] + bsum[192 + tid]) * oscale;
                if (a.bias_part != nullptr) a.bias_part[((long)(blockIdx.y / a.ci_tiles) * gridDim.x + blockIdx.x) * 64 + tid] = v;
                else if (co0 + tid < a.COUT) atomicAdd(a.dbias + co0 + tid, v);
            }
        }
        return;
    }
    // ---- merge: lane (ci = ci0 + wave*16 + i16, co = co0 + i*16 + g*4 + r)
    if (a.slabs != nullptr) {
        // partial tile of this block, tile-local [tap][co][ci]; conv_wgrad_rows_reduce_kernel adds the slabs of a (co, ci) pair
        float* slab = a.slabs + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (9 * 64 * 64) + wave * 16 + i16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(t * 64 + i * 16 + g * 4 + r) * 64] = acc[t][i][r] * oscale;
    }
    const int ci = ci0 + wave * 16 + i16;
    if (a.slabs == nullptr && ci < CIN) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + i * 16 + g * 4 + r;
                    if (co < a.COUT) atomicAdd(a.dw + ((long)co * 9 + t) * CIN + ci, acc[t][i][r] * oscale);
                }
    }
    if (do_bias && i16 == 0) {
        float* bp = a.bias_part != nullptr ? a.bias_part + ((long)(blockIdx.y / a.ci_tiles) * gridDim.x + blockIdx.x) * 64 : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = i * 16 + g * 4 + r;
                if (bp != nullptr) bp[col] = accb[i][r] * oscale;
                else if (co0 + col < a.COUT) atomicAdd(a.dbias + co0 + col, accb[i][r] * oscale);
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Ping-pong form of the row walker (round 3; wide maps, W % 32 == 0).  Same tile (64 co x 64 ci x 9 taps per block, a wave =
// 64 co x 16 ci), same LDS image and transposed fragment reads as conv_wgrad_rows_kernel<0>, but
//   * ONE block of 8 waves per CU instead of two of 4: the two waves of a SIMD hold the SAME dW tile and take alternate image
//     rows, one segment apart (MI355X_MICROARCH.md, "Two waves per SIMD"): while one multiplies its row (36 + 4 MFMAs on
//     registers only), its partner reads the 26 fragments of the next row and issues its share of the LDS-DMA;
//   * a block walks a CONTIGUOUS range of image rows (flattened over image, strip, y), so the two halo rows are re-fetched only
//     where a column starts; every step - halo or row - loads exactly one X row (+ one dY row) into ring slot `step mod 8`,
//     every loading wave issues three requests per step (dummies where it has none), so all waits are `vmcnt(3)`;
//   * at the end the second half hands its accumulators to the first through LDS (the rings are dead by then) and ONE partial
//     tile per block goes to the slab area: 256 slabs of 147 KB per layer instead of 512 (the merge was 16 % of the family's time).
// ------------------------------------------------------------------------------------------------------------
constexpr int WP_NS = 8, WP_D = 4;                           // ring slots (X and dY), request distance in steps
constexpr int WP_XBYTES = WP_NS * WR_XSLOT, WP_YBYTES = WP_NS * WR_YSLOT;
constexpr int WP_DUMMY = WP_XBYTES + WP_YBYTES;
constexpr int WP_LDS = WP_DUMMY + 1024;                      // 74 752 B; the hand-over of 72 accumulator registers needs 73 728

__global__ __launch_bounds__(512) void conv_wgrad_pp_kernel(WrArgs a, int rows_per_block, int rows_total) {
    extern __shared__ __attribute__((aligned(16))) char wp_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, wq = wave & 3;
    const int H = a.H, W = a.W, CIN = a.CIN;
    const int co0 = (blockIdx.y / a.ci_tiles) * 64, ci0 = (blockIdx.y % a.ci_tiles) * 64;
    const int strips = W / 32;
    const int R0 = (int)blockIdx.x * rows_per_block;
    const int R1 = R0 + rows_per_block < rows_total ? R0 + rows_per_block : rows_total;
    if (R0 >= R1) return;
    const int segs = (R1 - 1) / H - R0 / H + 1;            // columns touched: each starts with two halo steps
    const int S = (R1 - R0) + 2 * segs;
    const int S_pad = S + (S & 1);
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)wp_smem);
    constexpr unsigned OOB = 0x80000000u;
    auto uniform_ptr = [](const void* q) {
        const unsigned long long v = (unsigned long long)(uintptr_t)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (void*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    };
    const int up = a.dy_up2 ? 1 : 0;
    const int HY = H >> up, WY = W >> up;
    const float oscale = up ? 0.25f : 1.f;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.x), 0, __builtin_amdgcn_readfirstlane(a.N * H * W * CIN * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.dy), 0, __builtin_amdgcn_readfirstlane(a.N * HY * WY * a.LD_DY * 2), 0x00020000);

    // ---- DMA lanes: pixel (lane >> 3) of an 8-pixel piece, logical 16-byte slot swizzled by (pixel >> 1) & 3 (see the kernel above)
    const int dpx = lane >> 3;
    const int dls = ((((lane & 7) >> 1) ^ ((dpx >> 1) & 3)) << 1) | (lane & 1);
    const bool x_ch_ok = ci0 + dls * 8 < CIN;
    const bool y_ch_ok = co0 + dls * 8 < a.LD_DY;
    const unsigned x_lane = (unsigned)((ci0 + dls * 8) * 2), y_lane = (unsigned)((co0 + dls * 8) * 2);

    // step cursor: (image n, strip origin x0, row y) of the step being REQUESTED; phase 0 / 1: halo steps (X rows y-1, y), phase 2: row
    // step (X row y + 1, dY row y, the MFMAs of row y).  All incremental - an integer division costs ~50 scalar instructions, and
    // a LOAD segment that spends them is longer than the partner's MFMA segment: the byte offsets of the column advance by one
    // row pitch per row and are rebuilt only where a column starts.
    struct Cursor { int n, x0, y, phase; unsigned xoff, yoff; };     // xoff: pixel (n, y, x0 - 1) of x; yoff: column origin of dy
    const unsigned x_pitch = (unsigned)(W * CIN * 2), y_pitch = (unsigned)(WY * a.LD_DY * 2);
    auto rebase = [&](Cursor& c) {
        c.xoff = (unsigned)(((c.n * H + c.y) * W + c.x0 - 1) * CIN * 2);
        c.yoff = (unsigned)(((c.n * HY) * WY + (c.x0 >> up)) * a.LD_DY * 2);
    };
    auto advance = [&](Cursor& c) {
        if (c.phase < 2) { ++c.phase; return; }
        c.xoff += x_pitch;
        if (++c.y == H) {
            c.y = 0; c.phase = 0;
            c.x0 += 32;
            if (c.x0 == W) { c.x0 = 0; ++c.n; }
            rebase(c);
        }
    };
    // per-lane constants of this wave's pieces: X piece wq (and piece 4 for wave 0), dY piece wq
    const int px_a = wq * 8 + dpx, px_b = 32 + dpx;
    const unsigned xl_a = (unsigned)(px_a * CIN * 2) + x_lane, xl_b = (unsigned)(px_b * CIN * 2) + x_lane;
    const unsigned yl = (unsigned)((((wq * 8 + dpx) >> up) * a.LD_DY) * 2) + y_lane;
    // the requests of step `c` (index s), issued by the four waves of the half that owns step s - D: X pieces 0..4 of one row
    // (wave wq takes piece wq, wave 0 also piece 4), dY pieces 0..3 (wave wq takes piece wq); everything else is a dummy
    auto issue_step = [&](const Cursor& c, int s) {
        const bool live = s < S;
        const int yx = c.y + c.phase - 1;
        const bool xrow_ok = live && (unsigned)yx < (unsigned)H;
        const unsigned xbase = c.xoff + (unsigned)(c.phase - 1) * x_pitch;
        const unsigned slot_x = (unsigned)((s & (WP_NS - 1)) * WR_XSLOT), slot_y = (unsigned)(WP_XBYTES + (s & (WP_NS - 1)) * WR_YSLOT);
        const bool left_edge = c.x0 == 0, right_edge = c.x0 + 32 == W;
        {
            const bool ok = xrow_ok && x_ch_ok && !(left_edge && px_a == 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(wp_smem + slot_x + (unsigned)wq * 1024u),
                                                     16, (int)(ok ? xbase + xl_a : OOB), 0, 0, 0);
        }
        {
            const bool real = wq == 0;
            const bool ok = real && xrow_ok && x_ch_ok && px_b < 34 && !(right_edge && px_b == 33);
            const unsigned m = real ? 0xffffffffu : 0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(wp_smem + (((slot_x + 4096u) & m) | ((unsigned)WP_DUMMY & ~m))),
                                                     16, (int)(ok ? xbase + xl_b : OOB), 0, 0, 0);
        }
        {
            const bool real = live && c.phase == 2;
            const unsigned off = (real && y_ch_ok) ? c.yoff + (unsigned)(c.y >> up) * y_pitch + yl : OOB;
            const unsigned m = real ? 0xffffffffu : 0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, (__attribute__((address_space(3))) void*)(wp_smem + (((slot_y + (unsigned)wq * 1024u) & m) | ((unsigned)WP_DUMMY & ~m))),
                                                     16, (int)off, 0, 0, 0);
        }
    };

    // ---- fragment read offsets (the row walker's)
    const int i16 = lane & 15, g = lane >> 4;
    const int prow = g * 4 + (i16 >> 2);
    unsigned a_off[4], b_off[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[i] = (unsigned)(prow * 128 + ((i ^ ((prow >> 1) & 3)) << 5) + (i16 & 3) * 8);
#pragma unroll
    for (int ds = 0; ds < 3; ++ds) b_off[ds] = (unsigned)((prow + ds) * 128 + ((wq ^ (((prow + ds) >> 1) & 3)) << 5) + (i16 & 3) * 8);

    f32x4_t acc[9][4], accb[4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = a.dbias != nullptr && ci0 == 0 && wq == 0;
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(SP_H16_ONE_PAIR, SP_H16_ONE_PAIR, SP_H16_ONE_PAIR, SP_H16_ONE_PAIR));

    // ---- prologue: steps 0 .. D-1 (each half requests the steps of its parity), everything landed before the first read.  Only
    // the REQUEST cursor exists; whether the step being computed is a row step comes out of a two-entry history of its phases
    Cursor rq;
    {
        const int col = R0 / H;
        rq.y = R0 - col * H; rq.n = col / strips; rq.x0 = (col - rq.n * strips) * 32; rq.phase = 0;
        rebase(rq);
    }
    if (half) advance(rq);
    int ph_now = rq.phase;                                  // phase of the step this half computes next ...
    issue_step(rq, half);
    advance(rq); advance(rq);
    int ph_next = rq.phase;                                 // ... and of the one after (requested 2 iterations = D steps ahead)
    issue_step(rq, half + 2);
    advance(rq); advance(rq);
    wr_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (half) __builtin_amdgcn_s_barrier();                 // this half runs one segment behind

    for (int s = half; s < S_pad; s += 2) {
        const bool compute = s < S && ph_now == 2;
        ph_now = ph_next;
        ph_next = rq.phase;                                 // the step requested in this iteration: s + D
        // ================= LOAD segment =================
        uint2 alo[4], ahi[4], blo[9], bhi[9];
        const unsigned ab = lds_base + (unsigned)(WP_XBYTES + (s & (WP_NS - 1)) * WR_YSLOT);
        unsigned bb[3];
#pragma unroll
        for (int dr = 0; dr < 3; ++dr) bb[dr] = lds_base + (unsigned)(((s - 2 + dr) & (WP_NS - 1)) * WR_XSLOT);
        if (compute) {
            // fragment halves pinned to adjacent registers v180 .. v231 (A: 180-195, B: 196-231): with free allocation hipcc builds every
            // MFMA operand tuple with two v_mov_b64 (80 moves per 40 MFMAs inside the MFMA segment)
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[180:181]}"(alo[0]) : "v"(ab + a_off[0]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[182:183]}"(ahi[0]) : "v"(ab + a_off[0]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[184:185]}"(alo[1]) : "v"(ab + a_off[1]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[186:187]}"(ahi[1]) : "v"(ab + a_off[1]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[188:189]}"(alo[2]) : "v"(ab + a_off[2]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[190:191]}"(ahi[2]) : "v"(ab + a_off[2]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[192:193]}"(alo[3]) : "v"(ab + a_off[3]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[194:195]}"(ahi[3]) : "v"(ab + a_off[3]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[196:197]}"(blo[0]) : "v"(bb[0] + b_off[0]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[198:199]}"(bhi[0]) : "v"(bb[0] + b_off[0]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[200:201]}"(blo[1]) : "v"(bb[0] + b_off[1]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[202:203]}"(bhi[1]) : "v"(bb[0] + b_off[1]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[204:205]}"(blo[2]) : "v"(bb[0] + b_off[2]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[206:207]}"(bhi[2]) : "v"(bb[0] + b_off[2]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[208:209]}"(blo[3]) : "v"(bb[1] + b_off[0]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[210:211]}"(bhi[3]) : "v"(bb[1] + b_off[0]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[212:213]}"(blo[4]) : "v"(bb[1] + b_off[1]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[214:215]}"(bhi[4]) : "v"(bb[1] + b_off[1]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[216:217]}"(blo[5]) : "v"(bb[1] + b_off[2]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[218:219]}"(bhi[5]) : "v"(bb[1] + b_off[2]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[220:221]}"(blo[6]) : "v"(bb[2] + b_off[0]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[222:223]}"(bhi[6]) : "v"(bb[2] + b_off[0]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[224:225]}"(blo[7]) : "v"(bb[2] + b_off[1]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[226:227]}"(bhi[7]) : "v"(bb[2] + b_off[1]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[228:229]}"(blo[8]) : "v"(bb[2] + b_off[2]));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[230:231]}"(bhi[8]) : "v"(bb[2] + b_off[2]));
        }
        issue_step(rq, s + WP_D);
        advance(rq); advance(rq);
        wr_wait_vmcnt<3>();                                 // everything this wave requested before this segment has landed
        wr_wait_lgkm<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ================= MFMA segment =================
        if (compute) {
            __builtin_amdgcn_s_setprio(1);
            wr_static_for<9>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const bf16x8_t bf = wr_frag(blo[t], bhi[t]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr_frag(alo[i], ahi[i]), bf, acc[t][i], 0, 0, 0);
            });
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr_frag(alo[i], ahi[i]), ones, accb[i], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!half) __builtin_amdgcn_s_barrier();               // the barrier the other half passes after its last segment

    // ---- the second half's accumulators join the first half's through LDS (two passes of 72 registers + the bias sums)
    wr_wait_vmcnt<0>();                                     // no request may land in the area any more
    __syncthreads();
    float* xch = reinterpret_cast<float*>(wp_smem);
    const int ltid = tid & 255;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (half) {
            int k = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (((t * 4 + i) & 1) == pass) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) xch[(k * 4 + r) * 256 + ltid] = acc[t][i][r];
                        ++k;
                    }
                }
        }
        __syncthreads();
        if (!half) {
            int k = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (((t * 4 + i) & 1) == pass) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[t][i][r] += xch[(k * 4 + r) * 256 + ltid];
                        ++k;
                    }
                }
        }
        __syncthreads();
    }
    if (a.dbias != nullptr && ci0 == 0) {                   // ... and the bias sums (block-uniform condition)
        if (half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) xch[(i * 4 + r) * 256 + ltid] = accb[i][r];
        }
        __syncthreads();
        if (!half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) accb[i][r] += xch[(i * 4 + r) * 256 + ltid];
        }
    }
    if (half) return;
    // ---- merge of the block's tile: lane (ci = ci0 + wq*16 + i16, co = co0 + i*16 + g*4 + r), as the row walker
    if (a.slabs != nullptr) {
        float* slab = a.slabs + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (9 * 64 * 64) + wq * 16 + i16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(t * 64 + i * 16 + g * 4 + r) * 64] = acc[t][i][r] * oscale;
    }
    const int ci = ci0 + wq * 16 + i16;
    if (a.slabs == nullptr && ci < CIN) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + i * 16 + g * 4 + r;
                    if (co < a.COUT) atomicAdd(a.dw + ((long)co * 9 + t) * CIN + ci, acc[t][i][r] * oscale);
                }
    }
    if (do_bias && i16 == 0) {
        float* bp = a.bias_part != nullptr ? a.bias_part + ((long)(blockIdx.y / a.ci_tiles) * gridDim.x + blockIdx.x) * 64 : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = i * 16 + g * 4 + r;
                if (bp != nullptr) bp[col] = accb[i][r] * oscale;
                else if (co0 + col < a.COUT) atomicAdd(a.dbias + co0 + col, accb[i][r] * oscale);
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same walk with the X rows carried in REGISTERS.  In the kernel above the halves take alternate row steps, so a wave
// reads the fragments of all three X rows of its step from LDS: 8 + 18 transposed reads per 36 MFMAs - the LOAD segment
// (~830 cycles of LDS pipe for four waves) is longer than the MFMA segment it pairs with (576).  Here each half walks its OWN
// contiguous half of the block's row range (own request cursor, own 4-slot ring: the same LDS and the same L2 traffic): two of
// the three X rows of a step are the previous step's, still in registers in fragment form, and a step reads only the row that is
// new - 8 + 6 reads per 36 MFMAs.  Three register sets rotate their roles (tap row 0 / 1 / 2), so the step loop is unrolled by
// three with the set of every read and of every MFMA operand fixed at compile time (sets pinned to v196.., as above).
// A ring slot is read in exactly one step (fragments of both operands), so a request distance of three steps needs four slots.
// ------------------------------------------------------------------------------------------------------------------------------
#define WP3_BREAD(LO0, HI0, LO1, HI1, LO2, HI2, SET)                                                                               \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[" LO0 "]}"(Blo[SET][0]) : "v"(bbase + b_off[0]));                     \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[" HI0 "]}"(Bhi[SET][0]) : "v"(bbase + b_off[0]));                  \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[" LO1 "]}"(Blo[SET][1]) : "v"(bbase + b_off[1]));                     \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[" HI1 "]}"(Bhi[SET][1]) : "v"(bbase + b_off[1]));                  \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[" LO2 "]}"(Blo[SET][2]) : "v"(bbase + b_off[2]));                     \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[" HI2 "]}"(Bhi[SET][2]) : "v"(bbase + b_off[2]));

#ifndef WP3_NS_
#define WP3_NS_ 4
#endif
constexpr int WP3_NS = WP3_NS_, WP3_D = WP3_NS - 1;        // ring slots per half, request distance in (own) steps (<= NS - 1)
constexpr int WP3_XBYTES = 2 * WP3_NS * WR_XSLOT, WP3_DUMMY = WP3_XBYTES + 2 * WP3_NS * WR_YSLOT;
constexpr int WP3_LDS = (WP3_DUMMY + 1024) > WP_LDS ? (WP3_DUMMY + 1024) : WP_LDS;      // (the accumulator hand-over needs 73 728 B)
template <bool TIMING>
__global__ __launch_bounds__(512) void conv_wgrad_pp3_kernel(WrArgs a, int rows_per_block, int rows_total) {
    extern __shared__ __attribute__((aligned(16))) char wp_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, wq = wave & 3;
    const int H = a.H, W = a.W, CIN = a.CIN;
    const int co0 = (blockIdx.y / a.ci_tiles) * 64, ci0 = (blockIdx.y % a.ci_tiles) * 64;
    const int strips = W / 32;
    const int B0 = (int)blockIdx.x * rows_per_block;
    const int B1 = B0 + rows_per_block < rows_total ? B0 + rows_per_block : rows_total;
    if (B0 >= B1) return;
    // this half's rows [R0, R1): the first half takes the extra row of an odd range
    const int Bm = B0 + (B1 - B0 + 1) / 2;
    const int R0 = half ? Bm : B0, R1 = half ? B1 : Bm;
    auto steps_of = [&](int r0, int r1) { return r1 > r0 ? (r1 - r0) + 2 * ((r1 - 1) / H - r0 / H + 1) : 0; };   // + two halo steps per column
    const int S = steps_of(R0, R1);
    const int S_other = half ? steps_of(B0, Bm) : steps_of(Bm, B1);
    const int S_max = S > S_other ? S : S_other;
    const int S_pad = (S_max + 2) / 3 * 3;
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)wp_smem);
    constexpr unsigned OOB = 0x80000000u;
    auto uniform_ptr = [](const void* q) {
        const unsigned long long v = (unsigned long long)(uintptr_t)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (void*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    };
    const int up = a.dy_up2 ? 1 : 0;
    const int HY = H >> up, WY = W >> up;
    const float oscale = up ? 0.25f : 1.f;
    // X descriptor based ONE PIXEL BEFORE the tensor: a strip's row offset (pixel x0 - 1) is then never negative - it travels in the
    // scalar offset operand, which the hardware adds as an unsigned 32-bit value (the pixel in front of image 0 is never requested)
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(reinterpret_cast<const char*>(a.x) - CIN * 2), 0, __builtin_amdgcn_readfirstlane((a.N * H * W + 1) * CIN * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.dy), 0, __builtin_amdgcn_readfirstlane(a.N * HY * WY * a.LD_DY * 2), 0x00020000);

    const int dpx = lane >> 3;
    const int dls = ((((lane & 7) >> 1) ^ ((dpx >> 1) & 3)) << 1) | (lane & 1);
    const bool x_ch_ok = ci0 + dls * 8 < CIN;
    const bool y_ch_ok = co0 + dls * 8 < a.LD_DY;
    const unsigned x_lane = (unsigned)((ci0 + dls * 8) * 2), y_lane = (unsigned)((co0 + dls * 8) * 2);

    // Request cursor, kept cheap: the LOAD segment of a step was 814 cycles, most of them ~110 dependent scalar instructions of
    // cursor and predicate arithmetic (the MFMA segment it pairs with: 640).  Now everything that depends on the lane or on the
    // column is computed where a column starts (voff_*: per-lane offsets with the channel / image-edge exclusions folded in as
    // out-of-range values), a row advances two scalar offsets, the row offset travels in the instruction's SCALAR offset operand,
    // and a row outside the image / past the end of the walk selects an empty descriptor (everything reads as zero).
    struct Cursor { int n, x0, y, phase; unsigned xrow, yrow; bool x_ok; };   // xrow: X row fetched by this step (pixel x0 - 1), yrow: dY row y
    const unsigned x_pitch = (unsigned)(W * CIN * 2), y_pitch = (unsigned)(WY * a.LD_DY * 2);
    const int px_a = wq * 8 + dpx, px_b = 32 + dpx;
    const unsigned xl_a = (unsigned)(px_a * CIN * 2) + x_lane, xl_b = (unsigned)(px_b * CIN * 2) + x_lane;
    const unsigned yl = y_ch_ok ? (unsigned)((((wq * 8 + dpx) >> up) * a.LD_DY) * 2) + y_lane : OOB;
    unsigned voff_a = OOB, voff_b = OOB;                    // per-lane offsets of X pieces wq and 4 inside a row (this column)
    const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.x), 0, 0, 0x00020000);
    auto column = [&](Cursor& c) {                          // a column starts: (n, x0, y) set, phase 0
        const long row = ((long)c.n * H + c.y - 1) * W + c.x0;              // the first step fetches X row y - 1 (descriptor: one pixel early)
        c.xrow = (unsigned)(row * CIN * 2);                                  // (row -1 of image 0 wraps: that step selects the empty descriptor)
        c.yrow = (unsigned)((((c.n * HY) + (c.y >> up)) * WY + (c.x0 >> up)) * a.LD_DY * 2);
        c.x_ok = c.y > 0;
        const bool left_edge = c.x0 == 0, right_edge = c.x0 + 32 == W;
        voff_a = (x_ch_ok && !(left_edge && px_a == 0)) ? xl_a : OOB;
        voff_b = (wq == 0 && x_ch_ok && px_b < 34 && !(right_edge && px_b == 33)) ? xl_b : OOB;
    };
    auto advance = [&](Cursor& c) {
        c.xrow += x_pitch;
        if (c.phase < 2) { ++c.phase; c.x_ok = c.y + c.phase - 1 < H; return; }
        if (up == 0 || (c.y & 1)) c.yrow += y_pitch;
        if (++c.y == H) {
            c.y = 0; c.phase = 0;
            c.x0 += 32;
            if (c.x0 == W) { c.x0 = 0; ++c.n; }
            column(c);
        } else {
            c.x_ok = c.y + 1 < H;
        }
    };
    const unsigned ring_x = (unsigned)(half * WP3_NS * WR_XSLOT), ring_y = (unsigned)(WP3_XBYTES + half * WP3_NS * WR_YSLOT);
    const unsigned lds_b = wq == 0 ? 4096u : 0u;            // piece 4 of an X row: wave 0 only (the others: a dropped request into the dummy KB)
    // the requests of (own) step s: X pieces 0..4 of one row (wave wq takes piece wq, wave 0 also piece 4), dY pieces 0..3.
    // PREPARED (descriptor choice, LDS targets, row offsets: all scalar) in the MFMA segment of the step before - scalar
    // instructions issue between MFMAs for free, in the LOAD segment they were a third of its length - and ISSUED in the LOAD segment.
    __amdgpu_buffer_rsrc_t rq_rx = null_rsrc, rq_ry = null_rsrc;
    unsigned rq_lds_a = 0, rq_lds_b = 0, rq_lds_y = 0, rq_xrow = 0, rq_yrow = 0;
    auto prepare = [&](const Cursor& c, int s) {
        const bool live = s < S;
        const unsigned slot = (unsigned)(s & (WP3_NS - 1));
        const unsigned slot_x = ring_x + slot * WR_XSLOT, slot_y = ring_y + slot * WR_YSLOT;
        rq_rx = (live && c.x_ok) ? x_rsrc : null_rsrc;
        rq_ry = (live && c.phase == 2) ? y_rsrc : null_rsrc;
        rq_lds_a = slot_x + (unsigned)wq * 1024u;
        rq_lds_b = wq == 0 ? slot_x + lds_b : (unsigned)WP3_DUMMY;
        rq_lds_y = slot_y + (unsigned)wq * 1024u;
        rq_xrow = c.xrow; rq_yrow = c.yrow;
    };
    auto issue = [&]() {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq_rx, (__attribute__((address_space(3))) void*)(wp_smem + rq_lds_a), 16, (int)voff_a, (int)rq_xrow, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq_rx, (__attribute__((address_space(3))) void*)(wp_smem + rq_lds_b), 16, (int)voff_b, (int)rq_xrow, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq_ry, (__attribute__((address_space(3))) void*)(wp_smem + rq_lds_y), 16, (int)yl, (int)rq_yrow, 0, 0);
    };

    const int i16 = lane & 15, g = lane >> 4;
    const int prow = g * 4 + (i16 >> 2);
    unsigned a_off[4], b_off[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[i] = (unsigned)(prow * 128 + ((i ^ ((prow >> 1) & 3)) << 5) + (i16 & 3) * 8);
#pragma unroll
    for (int ds = 0; ds < 3; ++ds) b_off[ds] = (unsigned)((prow + ds) * 128 + ((wq ^ (((prow + ds) >> 1) & 3)) << 5) + (i16 & 3) * 8);

    f32x4_t acc[9][4], accb[4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = a.dbias != nullptr && ci0 == 0 && wq == 0;
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, make_uint4(SP_H16_ONE_PAIR, SP_H16_ONE_PAIR, SP_H16_ONE_PAIR, SP_H16_ONE_PAIR));

    // ---- cursors: rq = the step being requested (D steps ahead), (cy, cphase) = row / phase of the step being computed
    Cursor rq;
    {
        const int r0 = R1 > R0 ? R0 : 0;
        const int col = r0 / H;
        rq.y = r0 - col * H; rq.n = col / strips; rq.x0 = (col - rq.n * strips) * 32; rq.phase = 0;
        column(rq);
    }
    int cy = rq.y, cphase = 0;
#pragma unroll
    for (int d = 0; d < WP3_D; ++d) { prepare(rq, d); issue(); advance(rq); }
    prepare(rq, WP3_D);                                     // the requests of the first LOAD segment
    wr_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (half) __builtin_amdgcn_s_barrier();                 // this half runs one segment behind

    // TIMING build (SP_TUNE_WGRAD_PP = 3): cycles per wave in [0] LOAD segment up to the request issue, [1] vmcnt wait, [2] lgkm wait,
    // [3] barrier after LOAD, [4] MFMA segment, [5] barrier after MFMA; written to the slab area instead of the tile
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int k) {
        if constexpr (TIMING) { const unsigned long long t = __builtin_readcyclecounter(); tacc[k] += t - tprev; tprev = t; }
    };
    if constexpr (TIMING) tprev = __builtin_readcyclecounter();
    uint2 Blo[3][3], Bhi[3][3];                              // [register set][tap column]
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k = 0; k < 3; ++k) { Blo[q][k] = make_uint2(0, 0); Bhi[q][k] = make_uint2(0, 0); }

    for (int s0 = 0; s0 < S_pad; s0 += 3) {
        wr_static_for<3>([&](auto qc) {
            constexpr int q = decltype(qc)::value;                 // register set written in this step = step index mod 3
            const int s = s0 + q;
            const bool compute = s < S && cphase == 2;
            // ================= LOAD segment =================
            uint2 alo[4], ahi[4];
            const unsigned slot = (unsigned)(s & (WP3_NS - 1));
            const unsigned bbase = lds_base + ring_x + slot * WR_XSLOT;
            const unsigned ab = lds_base + ring_y + slot * WR_YSLOT;
            if constexpr (q == 0) { WP3_BREAD("196:197", "198:199", "200:201", "202:203", "204:205", "206:207", 0) }
            else if constexpr (q == 1) { WP3_BREAD("208:209", "210:211", "212:213", "214:215", "216:217", "218:219", 1) }
            else { WP3_BREAD("220:221", "222:223", "224:225", "226:227", "228:229", "230:231", 2) }
            if (compute && !(TIMING && a.thin_mode == 9)) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[180:181]}"(alo[0]) : "v"(ab + a_off[0]));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[182:183]}"(ahi[0]) : "v"(ab + a_off[0]));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[184:185]}"(alo[1]) : "v"(ab + a_off[1]));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[186:187]}"(ahi[1]) : "v"(ab + a_off[1]));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[188:189]}"(alo[2]) : "v"(ab + a_off[2]));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[190:191]}"(ahi[2]) : "v"(ab + a_off[2]));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "={v[192:193]}"(alo[3]) : "v"(ab + a_off[3]));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "={v[194:195]}"(ahi[3]) : "v"(ab + a_off[3]));
            }
            if (!(TIMING && a.thin_mode == 8)) issue();
            stamp(0);
            wr_wait_vmcnt<3 * (WP3_D - 1)>();               // the requests of the last D - 1 LOAD segments may fly: step s + 1 has landed
            stamp(1);
            wr_wait_lgkm<0>();
            stamp(2);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp(3);
            // ================= MFMA segment: tap row 0 / 1 / 2 = the sets written two / one / zero steps ago =================
            if (compute) {
                __builtin_amdgcn_s_setprio(1);
                wr_static_for<9>([&](auto tc) {
                    constexpr int t = decltype(tc)::value, dr = t / 3, ds = t % 3, set = (q + 1 + dr) % 3;
                    const bf16x8_t bf = wr_frag(Blo[set][ds], Bhi[set][ds]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr_frag(alo[i], ahi[i]), bf, acc[t][i], 0, 0, 0);
                });
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr_frag(alo[i], ahi[i]), ones, accb[i], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
            }
            advance(rq);                                    // (scalar; voff_* change where a column starts)
            prepare(rq, s + 1 + WP3_D);
            __builtin_amdgcn_sched_barrier(0);
            stamp(4);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp(5);
            // the phase / row of the next step of this half
            if (cphase < 2) ++cphase;
            else if (++cy == H) { cy = 0; cphase = 0; }
        });
    }
    if (!half) __builtin_amdgcn_s_barrier();               // the barrier the other half passes after its last segment
    if constexpr (TIMING) {
        if (lane == 0 && a.slabs != nullptr) {
            float* out = a.slabs + (((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8;
            for (int k = 0; k < 6; ++k) out[k] = (float)tacc[k];
            out[6] = (float)S;
        }
        return;
    }

    // ---- the second half's accumulators join the first half's through LDS (two passes of 72 registers + the bias sums)
    wr_wait_vmcnt<0>();
    __syncthreads();
    float* xch = reinterpret_cast<float*>(wp_smem);
    const int ltid = tid & 255;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (half) {
            int k = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (((t * 4 + i) & 1) == pass) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) xch[(k * 4 + r) * 256 + ltid] = acc[t][i][r];
                        ++k;
                    }
                }
        }
        __syncthreads();
        if (!half) {
            int k = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (((t * 4 + i) & 1) == pass) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[t][i][r] += xch[(k * 4 + r) * 256 + ltid];
                        ++k;
                    }
                }
        }
        __syncthreads();
    }
    if (a.dbias != nullptr && ci0 == 0) {
        if (half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) xch[(i * 4 + r) * 256 + ltid] = accb[i][r];
        }
        __syncthreads();
        if (!half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) accb[i][r] += xch[(i * 4 + r) * 256 + ltid];
        }
    }
    if (half) return;
    if (a.slabs != nullptr) {
        float* slab = a.slabs + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (9 * 64 * 64) + wq * 16 + i16;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(t * 64 + i * 16 + g * 4 + r) * 64] = acc[t][i][r] * oscale;
    }
    const int ci = ci0 + wq * 16 + i16;
    if (a.slabs == nullptr && ci < CIN) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + i * 16 + g * 4 + r;
                    if (co < a.COUT) atomicAdd(a.dw + ((long)co * 9 + t) * CIN + ci, acc[t][i][r] * oscale);
                }
    }
    if (do_bias && i16 == 0) {
        float* bp = a.bias_part != nullptr ? a.bias_part + ((long)(blockIdx.y / a.ci_tiles) * gridDim.x + blockIdx.x) * 64 : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = i * 16 + g * 4 + r;
                if (bp != nullptr) bp[col] = accb[i][r] * oscale;
                else if (co0 + col < a.COUT) atomicAdd(a.dbias + co0 + col, accb[i][r] * oscale);
            }
    }
}

// dW[co][tap][ci] += sum over the nblk slabs of pair blockIdx.y; blockIdx.z takes every gridDim.z-th slab (a pair with
// hundreds of slabs would otherwise be summed by 36 blocks in one long dependent chain) and the few partial sums meet in
// dW through atomics.
__global__ __launch_bounds__(256) void conv_wgrad_rows_reduce_kernel(const float* __restrict__ slabs, int nblk, float* __restrict__ dw,
                                                                     int CIN, int COUT, int ci_tiles, const float* __restrict__ bias_part,
                                                                     float* __restrict__ dbias, int ordered, int kbeg, int kend) {
    // [kbeg, kend): the blocks (of every pair) whose partial tiles this launch adds - all of them, or the blocks that walked the rows
    // of ONE group of a two-group batch (sp_wgrad_rows_launch_pair); nblk stays the number of slabs per pair (the slab stride)
    const int co0 = (blockIdx.y / ci_tiles) * 64, ci0 = (blockIdx.y % ci_tiles) * 64;
    if (bias_part != nullptr && blockIdx.z == 0 && ci0 == 0 && (!ordered || blockIdx.x == 0)) {
        // bias gradient: the per-block partial sums of this co tile, split over the gridDim.x blocks of the pair and the four
        // 64-thread groups of a block (a single chain over 512 partials took longer than the whole tile reduction)
        __shared__ float bred[256];
        const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
        const float* bp = bias_part + (long)(blockIdx.y / ci_tiles) * nblk * 64 + col;
        float t = 0.f;
        // ordered: block 0 of the pair alone sums all partials (fixed order) and is the only writer of its dbias entries
        const int kstep = ordered ? 4 : gridDim.x * 4;
#pragma unroll 8
        for (int k = kbeg + (ordered ? 0 : blockIdx.x * 4) + grp; k < kend; k += kstep) t += bp[(long)k * 64];
        bred[threadIdx.x] = t;
        __syncthreads();
        if (threadIdx.x < 64 && co0 + col < COUT) {
            const float tot = bred[col] + bred[64 + col] + bred[128 + col] + bred[192 + col];
            if (ordered) dbias[co0 + col] += tot; else atomicAdd(dbias + co0 + col, tot);
        }
    }
    const float* base = slabs + (long)blockIdx.y * nblk * (9 * 64 * 64);
    const int e4 = blockIdx.x * 256 + threadIdx.x;                 // float4 index inside the tile: [tap][co][ci / 4]
    if (e4 >= 9 * 64 * 16) return;
    const int t = e4 / (64 * 16), co = co0 + (e4 / 16) % 64, ci = ci0 + (e4 % 16) * 4;
    if (co >= COUT || ci >= CIN || kbeg + (int)blockIdx.z >= kend) return;
    float4 s = reinterpret_cast<const float4*>(base + (long)(kbeg + blockIdx.z) * (9 * 64 * 64))[e4];
    // (unrolled: the loads of eight slabs fly together - the plain loop waited for each: 17 us per launch for 75 MB)
#pragma unroll 8
    for (int k = kbeg + blockIdx.z + gridDim.z; k < kend; k += gridDim.z) {
        const float4 v = reinterpret_cast<const float4*>(base + (long)k * (9 * 64 * 64))[e4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* dst = dw + ((long)co * 9 + t) * CIN + ci;
    if (gridDim.z == 1) {
        float4 d = *reinterpret_cast<float4*>(dst);
        d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
        *reinterpret_cast<float4*>(dst) = d;
    } else {
        atomicAdd(dst, s.x); atomicAdd(dst + 1, s.y); atomicAdd(dst + 2, s.z); atomicAdd(dst + 3, s.w);
    }
}

}  // namespace

// Returns SP_OK after launching, or 1 if the shape is not covered (caller falls back to the per-tap kernel).
// narrow maps walk 2 (16 wide) or 4 (8 wide) images side by side.  16 x 16 maps gain (512 -> 512 at batch 20: 74.9 -> 49.7 us);
// 8 x 8 maps have too few rows per block to amortise the pipeline ramp (24.0 -> 29.3 us) and stay on the per-tap kernel unless
// SP_TUNE_WGRAD_ROWS = 3; 2 keeps every narrow map on the per-tap kernel
static int wr_narrow(int h, int w, int dy_up2) {
    if (w % 32 == 0) return 0;
    const int mode = sp_tune(SP_TUNE_WGRAD_ROWS, 1);
    const bool covered = (w == 16 && mode != 2) || (w == 8 && mode == 3);
    return (covered && h % WR_R == 0 && !dy_up2) ? w : -1;
}

long sp_wgrad_rows_workspace(int n, int h, int w, int cin, int cout) {
    if (wr_narrow(h, w, 0) < 0 || h % WR_R != 0) return 0;
    return 512L * (9 * 64 * 64 + 64);          // partial tiles + partial bias sums of at most 512 blocks
}

int sp_wgrad_rows_launch(const void* x, const void* dy, float* dw, float* dbias, int n, int h, int w, int cin, int cout,
                         int ld_dy, float* ws, long ws_floats, int dy_up2, hipStream_t s) {
    const int nw = wr_narrow(h, w, dy_up2);
    if (nw < 0 || h % WR_R != 0) return 1;
    const int strips = nw ? 1 : w / 32;                    // strips per image (wide maps) ...
    const int groups = nw ? (n + 32 / nw - 1) / (32 / nw) : n;   // ... or images per strip (narrow maps)
    if ((long)n * h * w * cin * 2 >= (1L << 30) || (long)n * h * w * ld_dy * 2 >= (1L << 30)) return 1;
    WrArgs a;
    a.x = reinterpret_cast<const bf16*>(x);
    a.dy = reinterpret_cast<const bf16*>(dy);
    a.dw = dw;
    a.dbias = dbias;
    a.N = n; a.H = h; a.W = w; a.CIN = cin; a.COUT = cout; a.LD_DY = ld_dy;
    a.dy_up2 = dy_up2;
    a.thin_mode = sp_tune(SP_TUNE_WGRAD_ROWS_THIN, 1);
    const int co_tiles = (cout + 63) / 64;
    a.ci_tiles = (cin + 63) / 64;
    const int pairs = co_tiles * a.ci_tiles;
    // Number of blocks: every block ends with one fp32-atomic merge of its 64 x 576 tile (measured ~0.17 us per block,
    // serialised: all blocks of a (co, ci) pair hit the same addresses), so time ~ F / (T r) + T m with r ~ 2.4 TFLOP/s per
    // block: T_opt = sqrt(F / (r m)) (scratch/bench_wgrad.py sweep, profiles/README.md), at most two blocks per CU.
    const int env_blocks = sp_tune(SP_TUNE_WGRAD_ROWS_BLOCKS, 0);
    const bool det = sp_deterministic(SP_BF16);
    const int env_slabs = det ? 1 : sp_tune(SP_TUNE_WGRAD_ROWS_SLABS, 1);
    const bool use_slabs = env_slabs && ws != nullptr && ws_floats >= 512L * (9 * 64 * 64 + 64) && cin % 4 == 0 && (env_slabs == 1 || pairs <= env_slabs);
    if (det && !use_slabs) return 1;                       // deterministic mode never merges with atomics: the caller's ordered per-tap path takes the layer
    const double flops = 2.0 * n * h * w * 9.0 * (64.0 * co_tiles) * (64.0 * a.ci_tiles);
    // with slabs the merge is a plain 147 KB store per block + one reduce pass (no serialisation): fill the chip
    int total = env_blocks > 0 ? env_blocks : (use_slabs ? 512 : (int)(sqrt(flops * 2.45e-6) + 0.5));
    if (total > 512) total = 512;
    int target = (total + pairs / 2) / pairs;
    if (target < 1) target = 1;
    // units = (image, strip, row chunk of ru rows); block b takes units b, b + nblk, ...  Time ~ (units per block) x ru image rows,
    // so ru and nblk are chosen to make every block walk the same number of rows: k = ceil(units / target) units each,
    // nblk = ceil(units / k) blocks, cost k * ru - minimised over the power-of-two chunk heights down to 8 rows (a unit
    // re-reads two halo rows, so ties go to the longer chunk).  The first version took the longest chunk that gave >= target
    // units: 40 units on 32 blocks (256 -> 256 @64^2, batch 20) left 24 blocks idle half of the time (cost 128 instead of 80).
    int ru = h, best_cost = 1 << 30, nblk = 1;
    for (int r = h; r >= 8 || r == h; r /= 2) {
        if (r % WR_R != 0) break;
        const long units = (long)groups * strips * (h / r);
        const long k = (units + target - 1) / target;
        const long cost = k * (r + 1);                     // + 1: a mild preference for long chunks beyond exact ties
        if (cost < best_cost) { best_cost = (int)cost; ru = r; nblk = (int)((units + k - 1) / k); }
        if (r % 2 != 0 || r / 2 < 8) break;
    }
    a.rows_per_unit = ru;
    a.units = groups * strips * (h / ru);
    // wide maps: the ping-pong form (one 8-wave block per CU, contiguous row ranges, half the partial tiles)
    if (nw == 0 && sp_tune(SP_TUNE_WGRAD_PP, 1)) {
        const int rows_total = n * strips * h;
        int nb = 256 / pairs;                                // one block per CU over all (co, ci) pairs of the layer
        if (nb < 1) nb = 1;
        int rpb = (rows_total + nb - 1) / nb;
        if (rpb < 8) rpb = rows_total < 8 ? rows_total : 8;
        nb = (rows_total + rpb - 1) / rpb;
        static bool pp_attr = false;
        if (!pp_attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WP_LDS);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_pp3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WP3_LDS);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_pp3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WP3_LDS);
            if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", WP_LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
            pp_attr = true;
        }
        a.slabs = use_slabs && (long)nb * pairs <= 512 ? ws : nullptr;
        a.bias_part = (a.slabs != nullptr && dbias != nullptr) ? ws + 512L * 9 * 64 * 64 : nullptr;
        a.thin_mode = 0;
        sp_note_route("conv_wgrad_pp3 (row walker, ping-pong) + rows_reduce");
        if (sp_tune(SP_TUNE_WGRAD_PP, 1) == 2) hipLaunchKernelGGL(conv_wgrad_pp_kernel, dim3((unsigned)nb, (unsigned)pairs), dim3(512), WP_LDS, s, a, rpb, rows_total);
        else if (sp_tune(SP_TUNE_WGRAD_PP, 1) == 3) { a.thin_mode = sp_tune(SP_TUNE_WGRAD_ROWS_THIN, 0); hipLaunchKernelGGL(conv_wgrad_pp3_kernel<true>, dim3((unsigned)nb, (unsigned)pairs), dim3(512), WP3_LDS, s, a, rpb, rows_total); SP_LAUNCH_CHECK(); return SP_OK; }
        else hipLaunchKernelGGL(conv_wgrad_pp3_kernel<false>, dim3((unsigned)nb, (unsigned)pairs), dim3(512), WP3_LDS, s, a, rpb, rows_total);
        if (a.slabs != nullptr) {
            int z = 512 / (36 * pairs);
            if (z > nb / 4) z = nb / 4;
            if (z < 1 || det) z = 1;
            hipLaunchKernelGGL(conv_wgrad_rows_reduce_kernel, dim3(9 * 64 * 16 / 256, (unsigned)pairs, (unsigned)z), dim3(256), 0, s, ws, nb, dw, cin, cout, a.ci_tiles, a.bias_part, dbias, det ? 1 : 0, 0, nb);
        }
        SP_LAUNCH_CHECK();
        return SP_OK;
    }
    // the slab area holds 512 partial tiles; the deterministic mode must not fall back to atomics with several blocks per pair
    if (det && (long)nblk * pairs > 512) nblk = 512 / pairs > 0 ? 512 / pairs : 1;
    static bool attr_set = false;
    if (!attr_set) {
        for (const void* k : {reinterpret_cast<const void*>(conv_wgrad_rows_kernel<0>), reinterpret_cast<const void*>(conv_wgrad_rows_kernel<16>),
                              reinterpret_cast<const void*>(conv_wgrad_rows_kernel<8>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, WR_LDS);
            if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", WR_LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        }
        attr_set = true;
    }
    a.slabs = use_slabs && (long)nblk * pairs <= 512 ? ws : nullptr;
    a.bias_part = (a.slabs != nullptr && dbias != nullptr) ? ws + 512L * 9 * 64 * 64 : nullptr;
    sp_note_route(nw == 16 ? "conv_wgrad_rows<16> + rows_reduce" : nw == 8 ? "conv_wgrad_rows<8> + rows_reduce" : "conv_wgrad_rows<0> + rows_reduce");
    if (nw == 16) hipLaunchKernelGGL(conv_wgrad_rows_kernel<16>, dim3((unsigned)nblk, (unsigned)pairs), dim3(256), WR_LDS, s, a);
    else if (nw == 8) hipLaunchKernelGGL(conv_wgrad_rows_kernel<8>, dim3((unsigned)nblk, (unsigned)pairs), dim3(256), WR_LDS, s, a);
    else hipLaunchKernelGGL(conv_wgrad_rows_kernel<0>, dim3((unsigned)nblk, (unsigned)pairs), dim3(256), WR_LDS, s, a);
    if (a.slabs != nullptr) {
        int z = 512 / (36 * pairs);                       // ~512 reducer blocks
        if (z > nblk / 4) z = nblk / 4;
        if (z < 1 || det) z = 1;                          // deterministic mode: one ordered chain per element, no atomics
        hipLaunchKernelGGL(conv_wgrad_rows_reduce_kernel, dim3(9 * 64 * 16 / 256, (unsigned)pairs, (unsigned)z), dim3(256), 0, s, ws, nblk, dw, cin, cout, a.ci_tiles, a.bias_part, dbias, det ? 1 : 0, 0, nblk);
    }
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// Two-group batch (sp_conv2d_wgrad_accum_pair): images [0, split) accumulate into (dw_a, dbias_a), the rest into (dw_b, dbias_b) -
// ONE launch of the ping-pong row walker over all n images (a block walks a contiguous range of image rows, so when the group
// boundary falls between two blocks every partial tile belongs to one group) and one reduce pass per group over that group's
// slabs.  Returns SP_OK after launching, 1 if the shape / plan is not covered (the caller then runs the two groups separately).
int sp_wgrad_rows_launch_pair(const void* x, const void* dy, float* dw_a, float* dbias_a, float* dw_b, float* dbias_b, int n, int split,
                              int h, int w, int cin, int cout, int ld_dy, float* ws, long ws_floats, int dy_up2, hipStream_t s) {
    if (wr_narrow(h, w, dy_up2) != 0 || h % WR_R != 0 || sp_tune(SP_TUNE_WGRAD_PP, 1) != 1 || split <= 0 || split >= n) return 1;
    if ((dbias_a == nullptr) != (dbias_b == nullptr)) return 1;
    if ((long)n * h * w * cin * 2 >= (1L << 30) || (long)n * h * w * ld_dy * 2 >= (1L << 30)) return 1;
    const bool det = sp_deterministic(SP_BF16);
    const int env_slabs = det ? 1 : sp_tune(SP_TUNE_WGRAD_ROWS_SLABS, 1);
    const int co_tiles = (cout + 63) / 64, ci_tiles = (cin + 63) / 64, pairs = co_tiles * ci_tiles;
    if (!(env_slabs == 1 && ws != nullptr && ws_floats >= 512L * (9 * 64 * 64 + 64) && cin % 4 == 0)) return 1;
    const int strips = w / 32;
    const int rows_total = n * strips * h;
    int nb = 256 / pairs;
    if (nb < 1) nb = 1;
    int rpb = (rows_total + nb - 1) / nb;
    if (rpb < 8) rpb = rows_total < 8 ? rows_total : 8;
    nb = (rows_total + rpb - 1) / rpb;
    const long boundary = (long)split * strips * h;                     // first flattened row of the second group
    if (boundary % rpb != 0 || (long)nb * pairs > 512) return 1;
    const int kb = (int)(boundary / rpb);
    WrArgs a;
    a.x = reinterpret_cast<const bf16*>(x);
    a.dy = reinterpret_cast<const bf16*>(dy);
    a.dw = dw_a;
    a.dbias = dbias_a;                                                   // (with slabs: only its being non-null matters)
    a.N = n; a.H = h; a.W = w; a.CIN = cin; a.COUT = cout; a.LD_DY = ld_dy;
    a.dy_up2 = dy_up2;
    a.thin_mode = 0;
    a.ci_tiles = ci_tiles;
    a.rows_per_unit = h;
    a.units = n * strips;
    a.slabs = ws;
    a.bias_part = dbias_a != nullptr ? ws + 512L * 9 * 64 * 64 : nullptr;
    static bool pp_attr = false;
    if (!pp_attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_pp3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WP3_LDS);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", WP3_LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        pp_attr = true;
    }
    sp_note_route("conv_wgrad_pp3 (row walker, ping-pong, two groups) + 2 x rows_reduce");
    hipLaunchKernelGGL(conv_wgrad_pp3_kernel<false>, dim3((unsigned)nb, (unsigned)pairs), dim3(512), WP3_LDS, s, a, rpb, rows_total);
    for (int grp = 0; grp < 2; ++grp) {
        const int k0 = grp ? kb : 0, k1 = grp ? nb : kb;
        int z = 512 / (36 * pairs);
        if (z > (k1 - k0) / 4) z = (k1 - k0) / 4;
        if (z < 1 || det) z = 1;
        hipLaunchKernelGGL(conv_wgrad_rows_reduce_kernel, dim3(9 * 64 * 16 / 256, (unsigned)pairs, (unsigned)z), dim3(256), 0, s, ws, nb, grp ? dw_b : dw_a,
                           cin, cout, ci_tiles, a.bias_part, grp ? dbias_b : dbias_a, det ? 1 : 0, k0, k1);
    }
    SP_LAUNCH_CHECK();
    return SP_OK;
}
