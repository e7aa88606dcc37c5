// EXPERIMENT (round 3) - NOT part of libsempyr.so.  Measured slower than the ping-pong kernel and parked here with its numbers
// (profiles/README.md, "one wave per SIMD").  To run it again: copy to csrc/conv_sw.hip, add it to csrc/build.sh, declare
// `int sp_conv_sw_launch(const sp_conv_params&, hipStream_t)` in common.h and call it from conv_igemm.hip's dispatch() where the
// ping-pong kernel is chosen; scratch/test_pp.py compares it bit for bit with the round-2 kernels (it was identical on all 28 cases).
// SP_TUNE_CONV_PP_PRIO bits 8-10 select the ablations used for the breakdown: 1 = no LDS-DMA requests in the stages, 2 = no fragment
// reloads, 4 = no barrier (results are then garbage; timing only).
//
// 3x3 convolution (forward / input gradient), bf16, Cout > 64, W % 32 == 0: ONE WAVE PER SIMD, software pipelined.
//
// Same tile (128 co x 8 x 32 px per block), LDS image, swizzles, LDS-DMA rings and epilogue as the ping-pong kernel
// (conv_pp.hip) - but FOUR waves per block instead of eight: a wave owns 64 co x (4 rows x 32 columns), i.e. 128 accumulator
// registers (the 512-register file of a SIMD belongs to one wave), so a stage's 96 MFMAs need 12 weight + 12 pixel fragment reads:
// 0.25 LDS kilobytes per MFMA instead of 0.42 - the ping-pong kernel's LOAD segment (LDS-bound, as long as its MFMA segment)
// disappears as a phase.  The wave pipelines ITSELF: between the MFMA groups of stage g it issues the fragment reads of stage
// g + 1 (second register set) and its share of the LDS-DMA requests of stage g + 3; one counted wait and ONE barrier (four
// waves) end the stage.  What measurements of round 3 say this buys: no LOAD segment, half the barrier releases, and the
// epilogue's VALU work no longer waits for a partner half.
//
//   stage g:   requests R(g) = weights of stage g + 3 (+ halo pieces: at tap column 2 the first part of chunk c + 2's halo, at tap
//              column 0 the second part of chunk c + 1's)      \
//              fragment reads of stage g + 1 (into the registers the 96 MFMAs of stage g have just used for the last time)
//              >  interleaved behind the twelve MFMA groups     /
//              s_waitcnt vmcnt(|R(g)|): everything older has landed - in particular the weights of stage g + 2 (requested in
//              stage g - 1) and, at tap column 1, the whole halo of chunk c + 1 (requested in the two stages before)
//              s_waitcnt lgkmcnt(0); s_barrier
//   hazards:   a weight slot is re-requested (stage g, slot (g + 3) & 3 = slot of stage g - 1) two barriers after its reads (issued
//              in stage g - 2) returned; halo buffer (c & 1) is re-requested from stage (c, 2) on, its last reads (stage (c, 2)'s
//              fragments) were issued in stage (c, 1) and returned before that stage's barrier; data is read one barrier after the
//              counted wait of every wave that requested a piece of it.
// Restrictions (the launcher falls back to conv_pp.hip): bf16, Cin >= 64 (two chunks: the request look-ahead spans at most two
// work items), whole 16-channel groups, no pooling, no tanh (the one-pass-per-operand epilogue only).
#include "conv_common.h"

namespace {

constexpr int SW_TW = 32, SW_CO_T = 128, SW_RW = 4, SW_TH = 8, SW_NB = 6, SW_NFR = 8, SW_HR = 10, SW_HP = 40;
constexpr int SW_HALO_INSTR = (SW_HR * SW_HP * 64 + 1023) / 1024;        // 25 wave-instructions of 1 KB per halo chunk
constexpr int SW_HALO_BUF = SW_HALO_INSTR * 1024;
constexpr int SW_HPW = (SW_HALO_INSTR + 3) / 4;                           // 7 per wave (round robin; the tail ones are dummies)
constexpr int SW_HPS0 = (SW_HPW + 1) / 2, SW_HPS1 = SW_HPW - SW_HPS0;     // issued at tap column 2 (two chunks ahead) / 0 (one ahead)
constexpr int SW_W_BYTES = 3 * SW_CO_T * 64, SW_W_INSTR = SW_W_BYTES / 1024, SW_W_PER = SW_W_INSTR / 4;   // 24 KB, 6 per wave
constexpr int SW_NWS = 4;
constexpr int SW_BIAS_MAX = 1024;
constexpr int SW_OFF_W = 2 * SW_HALO_BUF, SW_OFF_BIAS = SW_OFF_W + SW_NWS * SW_W_BYTES, SW_OFF_DUMMY = SW_OFF_BIAS + SW_BIAS_MAX * 4;
constexpr int SW_LDS = SW_OFF_DUMMY + 1024;
constexpr int SW_NUM_CU = 256;

template <int HACK>
__global__ __launch_bounds__(256) void conv3x3_sw_kernel(sp_conv_params p, int cotiles, int total) {
    using T = bf16;
    constexpr int E = 8, KC = 32, CO_T = SW_CO_T, RW = SW_RW, NB = SW_NB, NFR = SW_NFR, HR = SW_HR, HP = SW_HP, TH = SW_TH;
    constexpr int HPW = SW_HPW, W_PER = SW_W_PER, W_BYTES = SW_W_BYTES, HALO_BUF = SW_HALO_BUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave >> 1, wpx = wave & 1;
    const int H = p.h, W = p.w_, CIN = p.cin_p;
    const int tiles_x = W / SW_TW, tiles_y = H / TH;
    const int kchunks = (CIN + KC - 1) / KC;                // >= 2 (launcher)
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int GR = gridDim.x;
    int bid = blockIdx.x;
    if ((GR & 7) == 0) bid = (bid & 7) * (GR >> 3) + (bid >> 3);
    const int my_items = (total - bid + GR - 1) / GR;
    const int nchunks = my_items * kchunks;
    if (nchunks <= 0) return;

    {   // bias -> LDS (fp32, zero padded to whole co-tiles)
        float* bias_l = reinterpret_cast<float*>(smem + SW_OFF_BIAS);
        const int nb = cotiles * CO_T < SW_BIAS_MAX ? cotiles * CO_T : SW_BIAS_MAX;
        for (int i = tid; i < nb; i += 256) bias_l[i] = (p.bias != nullptr && i < p.cout) ? p.bias[i] : 0.f;
    }

    constexpr unsigned OOB = 0x80000000u, OOB_C = 0x40000000u;
    const int up = p.in_up2 ? 1 : 0;
    const int HS = H >> up, WS = W >> up;
    auto uniform_ptr = [](const void* q) {
        const unsigned long long v = (unsigned long long)(uintptr_t)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (void*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    };
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.x), 0,
        __builtin_amdgcn_readfirstlane(p.n * HS * WS * CIN * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(wg), 0,
        __builtin_amdgcn_readfirstlane(p.cout * 9 * CIN * 2), 0x00020000);
    const int ls = ((lane & 3) ^ ((lane >> 3) & 3)) * E;
    unsigned h_off[HPW], w_off[W_PER];
    struct Coords { int co_i, tx_i, ty_i, n; };
    Coords cur, nxt;
    int s_co, s_tx, s_ty, s_n;
    {
        int t = bid;
        cur.co_i = t % cotiles; t /= cotiles;
        cur.tx_i = t % tiles_x; t /= tiles_x;
        cur.ty_i = t % tiles_y; cur.n = t / tiles_y;
        t = GR;
        s_co = t % cotiles; t /= cotiles;
        s_tx = t % tiles_x; t /= tiles_x;
        s_ty = t % tiles_y; s_n = t / tiles_y;
    }
    auto advance = [&](const Coords& c) {
        Coords r;
        r.co_i = c.co_i + s_co; int cy = r.co_i >= cotiles ? 1 : 0; r.co_i -= cy ? cotiles : 0;
        r.tx_i = c.tx_i + s_tx + cy; cy = r.tx_i >= tiles_x ? 1 : 0; r.tx_i -= cy ? tiles_x : 0;
        r.ty_i = c.ty_i + s_ty + cy; cy = r.ty_i >= tiles_y ? 1 : 0; r.ty_i -= cy ? tiles_y : 0;
        r.n = c.n + s_n + cy;
        return r;
    };
    nxt = advance(cur);
    auto set_halo_desc = [&](const Coords& c) {
        const int n = c.n, ty0 = c.ty_i * TH, tx0 = c.tx_i * SW_TW;
        int l4 = lane >> 2;
        asm volatile("" : "+v"(l4));
#pragma unroll
        for (int i = 0; i < HPW; ++i) {
            const int hp = (i * 4 + wave) * 16 + l4;                   // piece q = i * 4 + wave
            const int hy = hp / HP, hx = hp - hy * HP;
            const int yy = ty0 - 1 + hy, xx = tx0 - 1 + hx;
            const bool ok = hx < SW_TW + 2 && hy < HR && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            h_off[i] = ok ? (unsigned)((((n * HS + (yy >> up)) * WS + (xx >> up)) * CIN + ls) * 2) : OOB;
        }
    };
    auto w_ls = [&](int i) { return ((lane & 3) ^ (((lane >> 3) & 1) | (((wave * W_PER + i) & 1) << 1))) * E; };
    auto set_w_desc = [&](const Coords& c) {
        const int co0 = c.co_i * CO_T;
        int l4 = lane >> 2;
        asm volatile("" : "+v"(l4));
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            const int row = (wave * W_PER + i) * 16 + l4;
            const int ts = row / CO_T, co = co0 + row % CO_T;
            w_off[i] = co < p.cout ? (unsigned)(((co * 9 + ts * 3) * CIN + w_ls(i)) * 2) : OOB;
        }
    };
    auto dma = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned dst, unsigned voff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + dst), 16, (int)voff, 0, 0, 0);
    };
    auto issue_halo_piece = [&](int i, bool valid, int c0, int slot) {
        const unsigned add = c0 + ls < CIN ? (unsigned)(c0 * 2) : OOB_C;
        const int q = i * 4 + wave;
        const unsigned m = (valid && q < SW_HALO_INSTR) ? 0xffffffffu : 0u;
        dma(x_rsrc, ((unsigned)(slot * HALO_BUF + q * 1024) & m) | ((unsigned)SW_OFF_DUMMY & ~m), ((h_off[i] + add) & m) | (OOB & ~m));
    };
    auto issue_w_piece = [&](int i, bool valid, int c0, int ds, int slot) {
        const unsigned base = (unsigned)((ds * CIN + c0) * 2);
        const unsigned add = c0 + w_ls(i) < CIN ? base : OOB_C;
        const unsigned m = valid ? 0xffffffffu : 0u;
        dma(w_rsrc, ((unsigned)(SW_OFF_W + slot * W_BYTES + (wave * W_PER + i) * 1024) & m) | ((unsigned)SW_OFF_DUMMY & ~m),
            ((w_off[i] + add) & m) | (OOB & ~m));
    };

    const int frow = lane & 15, fslot = lane >> 4;
    const unsigned a_addr = lds_base + SW_OFF_W + (wco * 64 + (frow >> 2) * 16 + (frow & 3)) * 64 +
                            ((fslot ^ (((frow >> 1) & 1) | (((frow >> 2) & 1) << 1))) << 4);
    unsigned b_addr[3];
#pragma unroll
    for (int ds = 0; ds < 3; ++ds)
        b_addr[ds] = lds_base + ((RW * wpx) * HP + frow + ds) * 64 + ((fslot ^ (((frow + ds) >> 1) & 3)) << 4);
    const unsigned bias_addr = lds_base + SW_OFF_BIAS + (wco * 64 + (lane >> 4) * 16) * 4;
    const bool bias_in_acc = p.bias != nullptr && !up && cotiles * CO_T <= SW_BIAS_MAX;

    f32x4_t acc[4][NFR];
    uint4 b4[4];
    auto bias_fetch = [&](bool live, int co0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) b4[i] = make_uint4(0, 0, 0, 0);
        if (bias_in_acc && live) {
            const unsigned ba = bias_addr + (unsigned)co0 * 4u;
            lds_rd128<0>(b4[0], ba); lds_rd128<16>(b4[1], ba); lds_rd128<32>(b4[2], ba); lds_rd128<48>(b4[3], ba);
            wait_lgkm<0>();                                 // (asm LDS reads are waited for in their own straight-line region)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NFR; ++j) acc[i][j] = __builtin_bit_cast(f32x4_t, b4[i]);
    };

    // fragment registers: ONE set.  A register is reloaded with the NEXT stage's fragment right behind the MFMA group that used it
    // last (weights of tap row dr: group (h = dr + 3, dr); pixel row h: group (h, min(h, 2))), so the next stage starts with its
    // first fragments long landed and the last-freed ones (tap row 2, pixel row 5) are not needed before its sixth / last group
    uint4 fa[3][4], fb[NB][2];
    auto read_a = [&](auto drc, unsigned ab) {
        constexpr int dr = decltype(drc)::value;
        static_for<4>([&](auto ic) { constexpr int i = decltype(ic)::value; lds_rd128<dr * (CO_T * 64) + i * 256>(fa[dr][i], ab); });
    };
    auto read_b = [&](auto hc, auto stc, unsigned hb) {
        constexpr int h = decltype(hc)::value, ST = decltype(stc)::value;
        lds_rd128<h * (HP * 64)>(fb[h][0], b_addr[ST] + hb);
        lds_rd128<h * (HP * 64) + 1024>(fb[h][1], b_addr[ST] + hb);
    };

    // ---- prologue: chunk 0's halo, weight stages 0..2, first part of chunk 1's halo; fragments of stage 0
    set_halo_desc(cur);
    set_w_desc(cur);
#pragma unroll
    for (int i = 0; i < HPW; ++i) issue_halo_piece(i, true, 0, 0);
#pragma unroll
    for (int ds = 0; ds < 3; ++ds)
#pragma unroll
        for (int i = 0; i < W_PER; ++i) issue_w_piece(i, true, 0, ds, ds);
    {
        const bool more1 = nchunks > 1;                     // chunk 1: position 1 of the first item (kchunks >= 2)
#pragma unroll
        for (int i = 0; i < SW_HPS0; ++i) issue_halo_piece(i, more1, KC, 1);
    }
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the bias copy
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    bias_fetch(true, cur.co_i * CO_T);
    static_for<3>([&](auto d) { read_a(d, a_addr); });
    static_for<NB>([&](auto hq) { read_b(hq, std::integral_constant<int, 0>{}, 0u); });
    wait_lgkm<0>();

    int g4 = 0, kc = 0;
    // one stage: ST = tap column, S = register set of its fragments
    auto stage = [&](auto stc, int gc) {
        constexpr int ST = decltype(stc)::value;
        constexpr int NH = ST == 2 ? SW_HPS0 : ST == 0 ? SW_HPS1 : 0, H0 = ST == 2 ? 0 : SW_HPS0;
        constexpr int NPIECE = NH + W_PER;
        const bool more1 = gc + 1 < nchunks, more2 = gc + 2 < nchunks;
        const bool last1 = kc + 1 == kchunks;               // chunk gc + 1 opens the next item
        const bool last2 = kc + 2 >= kchunks;               // chunk gc + 2 lies in the next item
        // descriptors of the request targets (before any read of this stage is in flight: the branches stay out of the region)
        if constexpr (ST == 0) { if (last1 && more1) set_w_desc(nxt); }
        if constexpr (ST == 2) { if (kc + 2 == kchunks && more2) set_halo_desc(nxt); }
        const int c0_w = last1 ? 0 : (kc + 1) * KC;                                     // weights: chunk gc + 1
        const int c0_h = ST == 2 ? (last2 ? (kc + 2 - kchunks) * KC : (kc + 2) * KC)    // halo at tap column 2: chunk gc + 2
                                 : c0_w;                                                 // at tap column 0: chunk gc + 1
        const int hslot = ST == 2 ? (gc & 1) : ((gc + 1) & 1);
        const int ws = (g4 + 3) & 3;
        // fragment source of stage g + 1: next tap column of this chunk, or column 0 of the next chunk
        constexpr int STN = (ST + 1) % 3;
        const unsigned ab_n = a_addr + (unsigned)(((g4 + 1) & 3) * W_BYTES);
        const unsigned hb_n = (unsigned)((ST == 2 ? ((gc + 1) & 1) : (gc & 1)) * HALO_BUF);
        // request k of this stage rides behind MFMA group k.  Tap column 2: the halo pieces (buffer of THIS chunk, whose last fragment
        // reads the other waves issue until the end of the stage before) come behind the mid-stage barrier: groups 7 .. 10
        auto piece = [&](auto kq) {
            constexpr int k = decltype(kq)::value;
            if constexpr (ST == 2) {
                if constexpr (k < W_PER) issue_w_piece(k, more1, c0_w, ST, ws);
                else if constexpr (k >= 7 && k < 7 + NH) issue_halo_piece(H0 + k - 7, more2, c0_h, hslot);
            } else {
                if constexpr (k < NH) issue_halo_piece(H0 + k, more1, c0_h, hslot);
                else if constexpr (k < NPIECE) issue_w_piece(k - NH, more1, c0_w, ST, ws);
            }
        };
        // requests of this stage issued up to and including group 6, and the whole previous stage's
        constexpr int ISSUED7 = ST == 2 ? W_PER : (NPIECE < 7 ? NPIECE : 7);
        constexpr int NPREV = ST == 0 ? SW_HPS0 + W_PER : ST == 1 ? SW_HPS1 + W_PER : W_PER;
        __builtin_amdgcn_s_setprio(1);
        // 12 MFMA groups (halo row h, tap row dr -> output row h - dr)
        static_for<NB * 3>([&](auto gi) {
            constexpr int h = decltype(gi)::value / 3, dr = decltype(gi)::value % 3, rr = h - dr;
            if constexpr (rr >= 0 && rr < RW) {
                constexpr int grp = [] { int k = 0; for (int hh = 0; hh <= h; ++hh) for (int d = 0; d < 3; ++d) { if (hh == h && d == dr) return k; if (hh - d >= 0 && hh - d < RW) ++k; } return k; }();
                // the group's other instructions go out one at a time in the shadows of its MFMAs (a wave alone on its SIMD hides
                // nothing behind another wave): the request behind the first, the pixel-row reload behind the fourth and the last,
                // the weight reload behind the last four
                constexpr bool RB = dr == (h < 2 ? h : 2), RA = h == dr + 3 && grp != 6;
                static_for<8>([&](auto mc) {
                    constexpr int m = decltype(mc)::value, i = m & 3, hh = m >> 2;
                    acc[i][rr * 2 + hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[dr][i]), __builtin_bit_cast(bf16x8_t, fb[h][hh]), acc[i][rr * 2 + hh], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (m == 0 && !(HACK & 1)) piece(std::integral_constant<int, grp>{});
                    if constexpr (RB && m == 3 && !(HACK & 2)) lds_rd128<h * (HP * 64)>(fb[h][0], b_addr[STN] + hb_n);
                    if constexpr (RB && m == 7 && !(HACK & 2)) lds_rd128<h * (HP * 64) + 1024>(fb[h][1], b_addr[STN] + hb_n);
                    if constexpr (RA && m >= 4 && !(HACK & 2)) lds_rd128<dr * (CO_T * 64) + (m - 4) * 256>(fa[dr][m - 4], ab_n);
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (grp == 6) {
                    // the ONE synchronisation point of the stage, in front of the first weight-fragment reload: the weights of stage
                    // g + 1 (requested in stage g - 2) have landed when everything but the previous stage's requests and this stage's
                    // first seven has; at tap column 1 the wait also covers the previous stage (the rest of the next chunk's halo,
                    // read from the start of the next stage on)
                    wait_vmcnt<ST == 1 ? ISSUED7 : NPREV + ISSUED7>();
                    if constexpr (!(HACK & 4)) __builtin_amdgcn_s_barrier();
                    if constexpr (!(HACK & 2)) read_a(std::integral_constant<int, dr>{}, ab_n);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        __builtin_amdgcn_s_setprio(0);
        wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
        g4 = (g4 + 1) & 3;
    };
    auto epilogue = [&](bool more_chunks) {
        const int n = cur.n, ty0 = cur.ty_i * TH, tx0 = cur.tx_i * SW_TW, co0 = cur.co_i * CO_T;
        const long pix0 = ((long)n * H + ty0 + RW * wpx) * W + tx0 + (lane & 15);
        const int co_b = co0 + wco * 64 + (lane >> 4) * 16;
        if (up) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NFR; ++j) acc[i][j] *= 0.25f;
        }
        if (co_b < p.cout) {
            const long off0 = pix0 * p.ldy + co_b;
            auto foff = [&](int j) { return off0 + ((long)(j >> 1) * W + (j & 1) * 16) * p.ldy; };
            if (!bias_in_acc && p.bias != nullptr) {
                float t[16];
                Wide16<float>::ld(p.bias + co_b, t);
#pragma unroll
                for (int c = 0; c < 16; ++c)
#pragma unroll
                    for (int jj = 0; jj < NFR; ++jj) acc[c >> 2][jj][c & 3] += t[c];
            }
            // (operand passes over half the tile at a time: 8 fragments x 2 x 16 bytes would be 64 registers of loads in flight)
            auto with_operand = [&](const T* src, auto&& apply) {
                static_for<2>([&](auto hc) {
                    constexpr int j0 = decltype(hc)::value * 4;
                    uint4 t[4][2];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const T* q = src + foff(j0 + jj);
                        t[jj][0] = *reinterpret_cast<const uint4*>(q);
                        t[jj][1] = *reinterpret_cast<const uint4*>(q + 8);
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const uint4& u = t[jj][k >> 2];
                            const uint32_t w = (k & 3) == 0 ? u.x : (k & 3) == 1 ? u.y : (k & 3) == 2 ? u.z : u.w;
                            acc[k >> 1][j0 + jj][2 * (k & 1)] = apply(acc[k >> 1][j0 + jj][2 * (k & 1)], __uint_as_float(w << 16));
                            acc[k >> 1][j0 + jj][2 * (k & 1) + 1] = apply(acc[k >> 1][j0 + jj][2 * (k & 1) + 1], __uint_as_float(w & 0xffff0000u));
                        }
                });
            };
            if (p.mask_src != nullptr) {
                const float slope = p.mask_neg_slope;
                with_operand(reinterpret_cast<const T*>(p.mask_src), [&](float a, float t) { return a * (t > 0.f ? 1.f : slope); });
            }
            if (p.res1 != nullptr) with_operand(reinterpret_cast<const T*>(p.res1), [](float a, float t) { return a + t; });
            if (p.res2 != nullptr) with_operand(reinterpret_cast<const T*>(p.res2), [](float a, float t) { return a + t; });
            if (p.act == SP_ACT_LRELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < NFR; ++jj) {
                        const f32x4_t sv = acc[i][jj] * 0.2f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float av = acc[i][jj][r], s1 = sv[r];
                            float mv;
                            asm("v_max_f32 %0, %1, %2" : "=v"(mv) : "v"(av), "v"(s1));
                            acc[i][jj][r] = mv;
                        }
                    }
            } else if (p.act == SP_ACT_RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < NFR; ++jj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][jj][r] = fmaxf(acc[i][jj][r], 0.f);
            }
            static_for<NFR>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                unsigned w8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) w8[k] = f32x2_to_bf16x2(acc[k >> 1][j][2 * (k & 1)], acc[k >> 1][j][2 * (k & 1) + 1]);
                T* q = reinterpret_cast<T*>(p.y) + foff(j);
                *reinterpret_cast<uint4*>(q) = make_uint4(w8[0], w8[1], w8[2], w8[3]);
                *reinterpret_cast<uint4*>(q + 8) = make_uint4(w8[4], w8[5], w8[6], w8[7]);
            });
        }
        kc = 0;
        cur = nxt;
        nxt = advance(nxt);
        bias_fetch(more_chunks, cur.co_i * CO_T);
    };
    for (int gc = 0; gc < nchunks; ++gc) {
        stage(std::integral_constant<int, 0>{}, gc);
        stage(std::integral_constant<int, 1>{}, gc);
        stage(std::integral_constant<int, 2>{}, gc);
        if (kc + 1 == kchunks) epilogue(gc + 1 < nchunks);
        else ++kc;
    }
    wait_vmcnt<0>();                                        // (dummy requests of the last stages)
}

}  // namespace

// conv_igemm.hip's dispatch(): SP_OK after launching, 1 if the shape is not covered (the caller keeps the ping-pong kernel)
int sp_conv_sw_launch(const sp_conv_params& p, hipStream_t s) {
    if (p.dtype != SP_BF16 || p.ksize != 3 || p.cout <= 64 || p.h % SW_TH != 0 || p.w_ % SW_TW != 0) return 1;
    if (p.cin_p < 64 || p.pool2 != 0 || (p.cout & 15) != 0 || (p.ldy & 7) != 0 || p.act == SP_ACT_TANH) return 1;
    if ((long)p.n * p.h * p.w_ * p.cin_p * 2 >= (1L << 30) || (long)p.cout * 9 * p.cin_p * 2 >= (1L << 30)) return 1;
    const int cotiles = (p.cout + SW_CO_T - 1) / SW_CO_T;
    const int total = p.n * (p.h / SW_TH) * (p.w_ / SW_TW) * cotiles;
    int grid = total < SW_NUM_CU ? total : SW_NUM_CU;
    if (grid >= 8) grid -= grid % 8;
    const int prio = sp_tune(SP_TUNE_CONV_PP_PRIO, 0);
    const int hack = prio > 0 ? (prio >> 8) & 7 : 0;
    auto go = [&](auto hc) {
        constexpr int HK = decltype(hc)::value;
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_sw_kernel<HK>), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS);
        hipLaunchKernelGGL(conv3x3_sw_kernel<HK>, dim3((unsigned)grid), dim3(256), SW_LDS, s, p, cotiles, total);
    };
    switch (hack) {
        case 1: go(std::integral_constant<int, 1>{}); break;
        case 2: go(std::integral_constant<int, 2>{}); break;
        case 3: go(std::integral_constant<int, 3>{}); break;
        case 7: go(std::integral_constant<int, 7>{}); break;
        default: go(std::integral_constant<int, 0>{}); break;
    }
    SP_LAUNCH_CHECK();
    return SP_OK;
}
