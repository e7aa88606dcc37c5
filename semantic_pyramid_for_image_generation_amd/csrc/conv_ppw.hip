// 3x3 convolution (forward / input gradient), Cout > 64, 16-row patches: the ping-pong kernel with a 64 co x (4 rows x 32 px) WAVE
// tile (128 accumulator registers) - conv_pp.hip's schedule with half the LDS traffic per MFMA.
//
// Why: conv_pp.hip's wave owns 64 co x (2 rows x 32 px).  A stage (one tap column of a 32-channel chunk) reads 12 weight fragments
// + 8 pixel fragments for 48 MFMAs - 20 KB per wave, and the LOAD segment that holds those reads takes ~1 050 cycles against 786 for
// the MFMA segment it feeds (TIMING build, scratch/time_pp.py: "frag reads issue" 30 % of a wave's time, MFMA segment 22 %): four
// waves x 20 KB + the LDS-DMA writes of an interval is ~100 B / cycle of the LDS's 128.  The kernel is LDS-bandwidth bound, its
// matrix pipe waits for the partner wave's reads.  Here a wave owns FOUR rows: 12 weight + 12 pixel fragments feed 96 MFMAs
// (0.25 reads per MFMA instead of 0.42 - the lockstep tall kernel's ratio, now under the two-waves-per-SIMD schedule).
//
// 128 accumulators leave ~128 registers for everything else (two waves per SIMD: 256 each), so a stage is cut in two SUB-SEGMENTS by
// pixel column half:
//     L_A  12 weight fragments + the 6 LEFT pixel fragments (rows h .. h + 5)      18 reads
//     M_A  48 MFMAs into the left-half accumulators
//     L_B  the 6 RIGHT pixel fragments (into the same registers; the weights stay)   6 reads  (+ this stage's LDS-DMA requests, the counted wait)
//     M_B  48 MFMAs into the right-half accumulators
// - 72 fragment registers live instead of 96.  The halves of the block run one segment apart as in conv_pp.hip; every interval has
// one half in an MFMA segment.
//
// LDS (162 816 of 163 840 bytes): two halo buffers of 18 x 36 pixels x 64 B (pitch 36: odd rows flip swizzle-key bit 1), a
// THREE-slot weight ring (24 KB per stage: requested two stages ahead = ~6 000 cycles), bias copy, dummy kilobyte.
// Hazards as in conv_pp.hip: a LOAD segment ends with lgkmcnt(0) before its barrier; data is read >= 1 barrier after the counted
// wait of every wave that requested a piece of it (the wait sits at the end of L_B, the reads of the next stage in L_A two
// segments later; with the requests in L_B a stage's own pieces are the only ones that may still fly at its wait); a ring slot / halo buffer is re-requested >= 1 barrier after its last read.
//
// Scope: 16-bit storage, Cout > 64, h % 16 == 0, w % 32 == 0, whole 16-channel groups (Cout % 16 == 0, ldy % 8 == 0), no tanh, no
// recorded pooling positions, no fused tail.  Everything else stays on conv_pp.hip / the tall kernel (conv_igemm.hip decides).
#include "conv_common.h"

namespace {

constexpr int PW_NUM_CU = 256;
constexpr int PW_BIAS_MAX = 1024;

// ---- K-split of the last, partial round of work items: conv_pp.hip's scheme ("tail split", top of that file) on this kernel's items.
// A piece is 8 waves x 64 lanes x 128 accumulator registers = 256 KB of fp32; an item takes ~1.8 x an 8-row item of conv_pp.hip
// (6.7 us per 32-channel chunk), a piece handed over ~1.5 x (twice the bytes, the same latency chain).
constexpr int PW_SK_MAX_PARTS = 4;
constexpr int PW_SK_SLAB_FLOATS = 8 * 64 * 128;

struct PWSplit { int parts, tail_items, grid; };
inline PWSplit pw_split_plan(int total, int kchunks, long workspace_bytes) {
    PWSplit r{0, 0, total < PW_NUM_CU ? total : PW_NUM_CU};
    // (less than one round: one block per item - rounded down to a multiple of 8 for the XCD remap, 100 items became 96 blocks of
    // which four took two items, i.e. two rounds; the kernels skip the remap when the grid is not a multiple of 8)
    const int mode = sp_tune(SP_TUNE_CONV_PP_SPLIT, 1);
    if (!mode || total <= 0 || (total < PW_NUM_CU && mode == 2)) return r;
    const int R = total % PW_NUM_CU;
    if (R == 0) return r;
    int pmax = PW_NUM_CU / R;
    if (pmax > PW_SK_MAX_PARTS) pmax = PW_SK_MAX_PARTS;
    int P = 1;
    long best = 67L * kchunks;
    const long handover = total >= 2 * PW_NUM_CU ? 60 : 90;
    for (int q = 2; q <= pmax && kchunks / q >= 2; ++q) {
        const long c = 67L * ((kchunks + q - 1) / q) + handover * (q - 1);
        if (c < best && 20 * c < 19 * 67L * kchunks) { best = c; P = q; }
    }
    if (P < 2 || (long)R * P * PW_SK_SLAB_FLOATS * 4 > workspace_bytes) return r;
    r.parts = P; r.tail_items = R;
    r.grid = total < PW_NUM_CU ? R * P : PW_NUM_CU;
    return r;
}

template <typename T>
struct PWGeom {
    static constexpr int E = 16 / (int)sizeof(T), KC = 4 * E;
    static constexpr int CO_T = 128, WPX = 4, RW = 4, FW = 2, TH = WPX * RW, NB = RW + 2, NFR = RW * FW, HR = TH + 2, TW = 32;
    static constexpr int HP = 36;
    static constexpr int HALO_INSTR = (HR * HP * 64 + 1023) / 1024;   // 41 wave-instructions of 1 KB per halo chunk
    static constexpr int HALO_BUF = HALO_INSTR * 1024;
    static constexpr int HPW = (HALO_INSTR + 7) / 8;                  // 6 per wave (round robin; the tail ones are dummies)
    static constexpr int HPS0 = (HPW + 1) / 2, HPS1 = HPW - HPS0;     // issued in stage 0 / stage 1 of the previous chunk
    static constexpr int W_BYTES = 3 * CO_T * 64, W_INSTR = W_BYTES / 1024, W_PER = (W_INSTR + 7) / 8;   // 24 KB per stage, 3 per wave
    static constexpr int NWS = 3, LA = 2;                             // ring slots; stages of look-ahead
    static constexpr int OFF_W = 2 * HALO_BUF, OFF_BIAS = OFF_W + NWS * W_BYTES, OFF_DUMMY = OFF_BIAS + PW_BIAS_MAX * 4;
    static constexpr int LDS = OFF_DUMMY + 1024;
};

template <typename T, bool TIMING = false, bool DMA_LB = true, bool POOL = false>
__global__ __launch_bounds__(512) void conv3x3_ppw_kernel(sp_conv_params p, int cotiles, int total, int prio, int sk_arg) {
    const int sk_parts = sk_arg & 255;                      // (conv_pp.hip: pieces per tail item; bit 8: the closing piece does not peek)
    const bool sk_peek = !(sk_arg & 256);
    static_assert(sizeof(T) == 2, "16-bit storage");
    using G = PWGeom<T>;
    constexpr int E = G::E, KC = G::KC, CO_T = G::CO_T, WPX = G::WPX, RW = G::RW, FW = G::FW, NB = G::NB, NFR = G::NFR, HR = G::HR, HP = G::HP, TH = G::TH;
    constexpr int PW_TW = G::TW;
    constexpr int HALO_INSTR = G::HALO_INSTR, HALO_BUF = G::HALO_BUF, HPW = G::HPW, W_BYTES = G::W_BYTES, W_PER = G::W_PER, W_INSTR = G::W_INSTR;
    constexpr int NWS = G::NWS, LA = G::LA;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / WPX, wpx = wave % WPX;
    const bool half_b = wave >= 4;                          // the half of the block that runs one segment behind
    const int H = p.h, W = p.w_, CIN = p.cin_p;
    const int tiles_x = W / PW_TW, tiles_y = H / TH;
    const int kchunks = (CIN + KC - 1) / KC;
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    // XCD-aware order (as conv_pp.hip): blocks of one XCD get consecutive work items, co-tiles of a patch adjacent
    const int GR = gridDim.x;
    int bid = blockIdx.x;
    if ((GR & 7) == 0) bid = (bid & 7) * (GR >> 3) + (bid >> 3);
    // tail split (conv_pp.hip): the first full_total items go round robin, the rest in K pieces
    const int full_total = sk_parts > 1 ? (total / GR) * GR : total;
    const int my_items = (full_total - bid + GR - 1) / GR;
    int t_item = -1, t_part = 0, t_k0 = 0, t_k1 = 0, t_j = 0;
    if (sk_parts > 1 && (int)blockIdx.x < (total - full_total) * sk_parts) {
        t_j = (int)blockIdx.x / sk_parts;
        t_part = (int)blockIdx.x - t_j * sk_parts;
        t_item = full_total + t_j;
        t_k0 = t_part * kchunks / sk_parts;
        t_k1 = (t_part + 1) * kchunks / sk_parts;
    }
    const bool has_tail = t_item >= 0;
    const bool t_owner = t_part == sk_parts - 1;
    const int nchunks = my_items * kchunks + (has_tail ? t_k1 - t_k0 : 0);
    if (nchunks <= 0) return;

    // ---- bias -> LDS (fp32, zero padded to whole co-tiles), before the first LDS-DMA is in flight
    {
        float* bias_l = reinterpret_cast<float*>(smem + G::OFF_BIAS);
        const int nb = cotiles * CO_T < PW_BIAS_MAX ? cotiles * CO_T : PW_BIAS_MAX;
        for (int i = tid; i < nb; i += 512) bias_l[i] = (p.bias != nullptr && i < p.cout) ? p.bias[i] : 0.f;
    }

    // ---- DMA descriptors (raw buffers: SGPR base + 32-bit byte offset per lane; an offset beyond num_records reads zeros)
    constexpr unsigned OOB = 0x80000000u, OOB_C = 0x40000000u;
    const int up = p.in_up2 ? 1 : 0;
    const int HS = H >> up, WS = W >> up;
    auto uniform_ptr = [](const void* q) {
        const unsigned long long v = (unsigned long long)(uintptr_t)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (void*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    };
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.x), 0,
        __builtin_amdgcn_readfirstlane(p.n * HS * WS * CIN * (int)sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(wg), 0,
        __builtin_amdgcn_readfirstlane(p.cout * 9 * CIN * (int)sizeof(T)), 0x00020000);
    const int ls = ((lane & 3) ^ ((lane >> 3) & 3)) * E;   // halo: logical slot (elements) this lane fetches, key (hp >> 1) & 3
    unsigned h_off[HPW], w_off[W_PER];
    struct Coords { int co_i, tx_i, ty_i, n; };
    Coords cur, nxt, tailc;
    int s_co, s_tx, s_ty, s_n;
    {
        int t = has_tail ? t_item : 0;
        tailc.co_i = t % cotiles; t /= cotiles;
        tailc.tx_i = t % tiles_x; t /= tiles_x;
        tailc.ty_i = t % tiles_y; tailc.n = t / tiles_y;
        t = bid;
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
    int it = 0;
    const int t_pos = has_tail ? (t_owner ? my_items : 0) : -1;     // a contributing piece leads its block, the owning piece closes it
    Coords strided = cur;
    auto item_coords = [&](int idx, bool& is_tail) {
        is_tail = idx == t_pos;
        if (is_tail) return tailc;
        const Coords c = strided;
        strided = advance(strided);
        return c;
    };
    bool cur_is_tail, nxt_is_tail;
    cur = item_coords(0, cur_is_tail);
    nxt = item_coords(1, nxt_is_tail);
    auto set_halo_desc = [&](const Coords& c) {
        const int n = c.n, ty0 = c.ty_i * TH, tx0 = c.tx_i * PW_TW;
        int l4 = lane >> 2;
        asm volatile("" : "+v"(l4));                       // recomputed per item (hoisted, the pairs of every piece stay live and spill)
#pragma unroll
        for (int i = 0; i < HPW; ++i) {
            const int hp = (i * 8 + wave) * 16 + l4;                   // piece q = i * 8 + wave
            const int hy = hp / HP, hx = hp - hy * HP;
            const int yy = ty0 - 1 + hy, xx = tx0 - 1 + hx;
            const bool ok = hx < PW_TW + 2 && hy < HR && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            h_off[i] = ok ? (unsigned)((((n * HS + (yy >> up)) * WS + (xx >> up)) * CIN + ls) * (int)sizeof(T)) : OOB;
        }
    };
    // weight rows (stage row = tap row * 128 + co) are swizzled by key = ((co >> 1) & 1) | (((co >> 4) & 1) << 1)
    auto w_ls = [&](int i) { return ((lane & 3) ^ (((lane >> 3) & 1) | (((wave * W_PER + i) & 1) << 1))) * E; };
    auto set_w_desc = [&](const Coords& c) {
        const int co0 = c.co_i * CO_T;
        int l4 = lane >> 2;
        asm volatile("" : "+v"(l4));
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            const int row = (wave * W_PER + i) * 16 + l4;
            const int ts = row / CO_T, co = co0 + row % CO_T;          // tap row inside the stage; the stage adds the tap column
            w_off[i] = (wave * W_PER + i < W_INSTR && co < p.cout) ? (unsigned)(((co * 9 + ts * 3) * CIN + w_ls(i)) * (int)sizeof(T)) : OOB;
        }
    };
    auto dma = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned dst /* wave-uniform LDS byte offset */, unsigned voff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + dst), 16, (int)voff, 0, 0, 0);
    };
    // one halo piece / one weight piece; !valid: a dummy request (same count) - selects written as masks (conv_pp.hip)
    auto issue_halo_piece = [&](int i, bool valid, int c0, int slot) {
        const unsigned add = c0 + ls < CIN ? (unsigned)(c0 * (int)sizeof(T)) : OOB_C;
        const int q = i * 8 + wave;
        const unsigned m = (valid && q < HALO_INSTR) ? 0xffffffffu : 0u;    // wave-uniform
        dma(x_rsrc, ((unsigned)(slot * HALO_BUF + q * 1024) & m) | ((unsigned)G::OFF_DUMMY & ~m), ((h_off[i] + add) & m) | (OOB & ~m));
    };
    auto issue_w_piece = [&](int i, bool valid, int c0, int ds, int slot) {
        const unsigned base = (unsigned)((ds * CIN + c0) * (int)sizeof(T));
        const unsigned add = c0 + w_ls(i) < CIN ? base : OOB_C;
        const unsigned m = (valid && wave * W_PER + i < W_INSTR) ? 0xffffffffu : 0u;
        dma(w_rsrc, ((unsigned)(G::OFF_W + slot * W_BYTES + (wave * W_PER + i) * 1024) & m) | ((unsigned)G::OFF_DUMMY & ~m),
            ((w_off[i] + add) & m) | (OOB & ~m));
    };

    // ---- fragment read addresses (conv_pp.hip's: permuted A rows -> a lane ends with 16 consecutive channels of its pixel)
    const int frow = lane & 15, fslot = lane >> 4;
    const unsigned a_addr = lds_base + G::OFF_W + (wco * 64 + (frow >> 2) * 16 + (frow & 3)) * 64 +
                            ((fslot ^ (((frow >> 1) & 1) | (((frow >> 2) & 1) << 1))) << 4);
    unsigned b_addr[3];
#pragma unroll
    for (int ds = 0; ds < 3; ++ds)
        b_addr[ds] = lds_base + ((RW * wpx) * HP + frow + ds) * 64 + ((fslot ^ (((frow + ds) >> 1) & 3)) << 4);
    const unsigned bias_addr = lds_base + G::OFF_BIAS + (wco * 64 + (lane >> 4) * 16) * 4;

    const bool bias_in_acc = p.bias != nullptr && !up && cotiles * CO_T <= PW_BIAS_MAX && p.img_scale == nullptr;
    f32x4_t acc[4][NFR];
    uint4 b4[4];
    auto bias_fetch = [&](bool live, int co0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) b4[i] = make_uint4(0, 0, 0, 0);
        if (bias_in_acc && live) {
            const unsigned ba = bias_addr + (unsigned)co0 * 4u;
            lds_rd128<0>(b4[0], ba); lds_rd128<16>(b4[1], ba); lds_rd128<32>(b4[2], ba); lds_rd128<48>(b4[3], ba);
            wait_lgkm<0>();                                 // (RULE of conv_pp.hip: the wait of an asm LDS read follows in the same region)
        }
    };

    // ---- prologue: chunk 0's halo, weight stages 0 .. LA - 1
    set_halo_desc(cur);
    set_w_desc(cur);
    int kc = cur_is_tail ? t_k0 : 0;
    int kc_end = cur_is_tail ? t_k1 : kchunks;
#pragma unroll
    for (int i = 0; i < HPW; ++i) issue_halo_piece(i, true, kc * KC, 0);
#pragma unroll
    for (int ds = 0; ds < LA; ++ds)
#pragma unroll
        for (int i = 0; i < W_PER; ++i) issue_w_piece(i, true, kc * KC, ds, ds);
    wait_vmcnt<(LA - 1) * W_PER>();                         // halo 0 and stage 0 landed (this wave's pieces); stage 1 may fly
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the bias copy
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    bias_fetch(!(cur_is_tail && !t_owner), cur.co_i * CO_T);
    auto acc_from_bias = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NFR; ++j) acc[i][j] = __builtin_bit_cast(f32x4_t, b4[i]);
    };
    acc_from_bias();
    if (half_b) __builtin_amdgcn_s_barrier();               // from here on this half runs one segment behind

    // TIMING build: cycles per wave in [0] L_A reads + requests, [1] L_B reads + counted wait, [2] barriers after LOAD segments,
    // [3] MFMA segments, [4] barriers after MFMA segments, [5] epilogue + item switch
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    auto stamp = [&](int k) {
        if constexpr (TIMING) {
            const unsigned long long t = __builtin_readcyclecounter();
            tacc[k] += t - tprev;
            tprev = t;
        }
    };
    if constexpr (TIMING) tprev = __builtin_readcyclecounter();
    int g3 = 0;                                             // weight ring slot of the stage being computed (stage index mod 3)
    for (int gc = 0; gc < nchunks; ++gc) {
        const bool more_chunks = gc + 1 < nchunks;
        const bool item_ends = kc + 1 == kc_end;
        const unsigned hb = (unsigned)((gc & 1) * HALO_BUF);
        const int c0_next = item_ends ? (nxt_is_tail ? t_k0 * KC : 0) : (kc + 1) * KC;
        auto stage = [&](auto sc) {
            constexpr int st = decltype(sc)::value;        // stage inside the chunk = tap column
            constexpr int TAP_STRIDE = CO_T * 64;
            constexpr int NH = st == 0 ? G::HPS0 : st == 1 ? G::HPS1 : 0, H0 = st == 0 ? 0 : G::HPS0;   // halo pieces requested in this stage
            constexpr int NPIECE = NH + W_PER;
            const unsigned ab = a_addr + (unsigned)(g3 * W_BYTES);
            const unsigned bb = b_addr[st] + hb;
            const unsigned bo = bb ^ 32u;                   // odd halo rows (pitch 36): swizzle key flipped in bit 1
            uint4 a[3][4], bf[NB];
            // the weights of stage g + 2 go into the slot of stage g - 1 (its reads ended >= two barriers ago)
            const int ws = g3 == 0 ? 2 : g3 - 1;
            // stage g + 2 = tap column (st + 2) % 3; from stage 0 it is still in THIS chunk, from stages 1 and 2 in the next one
            constexpr int TS = (st + 2) % 3;
            const int c0_w = st == 0 ? kc * KC : c0_next;
            const bool w_valid = st == 0 ? true : more_chunks;
            auto piece = [&](auto kc_) {                    // request number k of this stage: halo of the next chunk, then weights of stage g + 2
                constexpr int k = decltype(kc_)::value;
                if constexpr (k < NH) issue_halo_piece(H0 + k, more_chunks, c0_next, (gc + 1) & 1);
                else if constexpr (k < NPIECE) issue_w_piece(k - NH, w_valid, c0_w, TS, ws);
            };
            // item switch of the request cursors: the halo pieces of the next item start in stage 0, its weights in stage 1
            if constexpr (st == 0) {
                if (item_ends && more_chunks) set_halo_desc(nxt);
            }
            if constexpr (st == 1) {
                if (item_ends && more_chunks) set_w_desc(nxt);
            }
            // ================= L_A: the 12 weight fragments and the left pixel fragments; one request behind every three reads =================
            static_for<6>([&](auto qc) {
                constexpr int q0 = decltype(qc)::value * 3;
                static_for<3>([&](auto rc) {
                    constexpr int r = q0 + decltype(rc)::value;     // read number: 0..11 = A (tap row r / 4, fragment r % 4), then B rows
                    if constexpr (r < 12) {
                        lds_rd128<(r / 4) * TAP_STRIDE + (r % 4) * 256>(a[r / 4][r % 4], ab);
                    } else {
                        constexpr int h = r - 12;
                        lds_rd128<h * (HP * 64)>(bf[h], (h & 1) ? bo : bb);
                    }
                });
                if constexpr (!DMA_LB) piece(qc);
            });
            stamp(0);
            wait_lgkm<0>();                                 // every LDS read of this wave has returned before it signals
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp(2);
            // ================= M_A: left column half =================
            __builtin_amdgcn_s_setprio(1);
            static_for<NB * 3>([&](auto gi) {
                constexpr int h = decltype(gi)::value / 3, dr = decltype(gi)::value % 3, rr = h - dr;
                if constexpr (rr >= 0 && rr < RW) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) Mma<T>::run(a[dr][i], bf[h], acc[i][rr * FW]);
                }
            });
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            stamp(3);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp(4);
            // ================= L_B: the right pixel fragments (the weights stay in registers) =================
            static_for<NB>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                lds_rd128<h * (HP * 64) + 1024>(bf[h], (h & 1) ? bo : bb);
                if constexpr (DMA_LB) piece(hc);            // the stage's requests ride here: this segment has ~450 cycles to spare against the partner's MFMA segment, L_A none (1-3 % per launch, scratch/test_ppw.py)
            });
            // the NEXT stage's weights were requested in the previous stage, the next chunk's halo before this stage's weight pieces:
            // only this stage's own requests may still fly
            wait_vmcnt<NPIECE>();
            wait_lgkm<0>();
            stamp(1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp(2);
            // ================= M_B: right column half =================
            __builtin_amdgcn_s_setprio(1);
            static_for<NB * 3>([&](auto gi) {
                constexpr int h = decltype(gi)::value / 3, dr = decltype(gi)::value % 3, rr = h - dr;
                if constexpr (rr >= 0 && rr < RW) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) Mma<T>::run(a[dr][i], bf[h], acc[i][rr * FW + 1]);
                }
            });
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            stamp(3);
            // (item end, the half that runs behind: its epilogue comes BEFORE this barrier - conv_pp.hip)
            if (!(st == 2 && item_ends && half_b)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp(4);
            g3 = g3 == 2 ? 0 : g3 + 1;
        };
        stage(std::integral_constant<int, 0>{});
        stage(std::integral_constant<int, 1>{});
        stage(std::integral_constant<int, 2>{});
        if (item_ends) {
            const int n = cur.n, ty0 = cur.ty_i * TH, tx0 = cur.tx_i * PW_TW, co0 = cur.co_i * CO_T;
            const long pix0 = ((long)n * H + ty0 + RW * wpx) * W + tx0 + (lane & 15);
            const int co_b = co0 + wco * 64 + (lane >> 4) * 16;          // this lane's 16 consecutive channels
            const bool wide = co_b < p.cout;
            bool run_epilogue = true;
            // conv_pp.hip's WAIT-FREE hand-over on 128 registers per lane: store the own slab, raise the counter, and whoever raises it
            // last sums the P slabs in the order P - 1, 0, ..., P - 2 and runs the epilogue; the closing piece looks first and keeps
            // its accumulators if everybody else is in.  Written as separate regions - stores | counter | ONE loop of loads whose trip
            // count is zero for everyone else - and sixteen registers per group behind an opaque offset, whole-fragment updates: as
            // an if / else the two paths (accumulators unchanged / updated) met in 128 phi copies and the main loop of this kernel
            // (~245 registers) spilled 300
            const bool sk_tail = cur_is_tail && sk_parts > 1;           // wave-uniform
            int woff = wave * 8192 + lane;
            asm volatile("" : "+v"(woff));                               // (keeps the slab addresses out of the chunk loop's live ranges)
            float* slab0 = reinterpret_cast<float*>(p.workspace) + ((long)t_j * sk_parts) * PW_SK_SLAB_FLOATS + woff;
            int* flag = p.split_sync + t_j * 8 + wave;
            bool sk_fast = false, sk_last = false;
            if (sk_tail && t_owner && sk_peek) sk_fast = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sk_parts - 1;
            if (sk_tail && !sk_fast) {
                float* dst = slab0 + (long)t_part * PW_SK_SLAB_FLOATS;
                static_for<NFR>([&](auto gc_) {
                    constexpr int g = decltype(gc_)::value;               // accumulators 16 g .. 16 g + 15 = acc[g / 2][4 (g % 2) .. + 3][0..3]
                    int o = g * 16 * 64;
                    asm volatile("" : "+v"(o));
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        __hip_atomic_store(dst + o + k * 64, acc[g / 2][4 * (g % 2) + (k >> 2)][k & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_sched_barrier(0);
                });
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the slab is at the memory side before the counter says so
                int old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sk_last = __builtin_amdgcn_readfirstlane(old) == sk_parts - 1;
            }
            if (sk_tail) run_epilogue = sk_fast || sk_last;
            const int sk_q0 = sk_fast ? 0 : -1, sk_q1 = (sk_tail && run_epilogue) ? sk_parts - 1 : -1;
#pragma unroll 1
            for (int q = sk_q0; q < sk_q1; ++q) {
                const float* src = slab0 + (long)(q < 0 ? sk_parts - 1 : q) * PW_SK_SLAB_FLOATS;
                const bool replace = q < 0;                              // first slab of the re-read order REPLACES the accumulators
                static_for<NFR>([&](auto gc_) {
                    constexpr int g = decltype(gc_)::value;
                    int o = g * 16 * 64;
                    asm volatile("" : "+v"(o));
                    f32x4_t t[4];                                        // whole fragments: the accumulators are 4-register tuples
#pragma unroll
                    for (int k = 0; k < 16; ++k) t[k >> 2][k & 3] = __hip_atomic_load(src + o + k * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const f32x4_t sum = acc[g / 2][4 * (g % 2) + f] + t[f];
                        acc[g / 2][4 * (g % 2) + f] = replace ? t[f] : sum;   // (a select on a uniform condition: bit for bit the slab)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            if (sk_tail && run_epilogue && lane == 0) __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (run_epilogue) {
            if (up) {                                                    // the 1/4 of the average-pooling gradient
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NFR; ++j) acc[i][j] *= 0.25f;
            }
            if constexpr (POOL) {                                        // (own instantiation: the plain form keeps its register allocation)
                // 2x2 average / maximum pooling in the epilogue (the discriminator's second convolutions, the VGG stages): fragments
                // j0 .. j0 + 3 = rows (2k, 2k + 1) x column halves; bias / scale / residuals / activation at the pooled resolution
                // (conv_common.h: conv_epilogue_pool2, the helper of the other 3x3 kernels - bit-identical outputs)
                if (wide) {
                    static_for<NFR / 4>([&](auto gq) {
                        constexpr int j0 = decltype(gq)::value * 4;
                        const long prow = ((long)n * (H >> 1) + ((ty0 + RW * wpx + (j0 >> 1)) >> 1)) * (W >> 1);
                        float av[16], bv[16];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                av[i * 4 + r] = pool2_combine(acc[i][j0][r], acc[i][j0 + 2][r], p.pool2 == 2);
                                bv[i * 4 + r] = pool2_combine(acc[i][j0 + 1][r], acc[i][j0 + 3][r], p.pool2 == 2);
                            }
                        conv_epilogue_pool2<T>(p, av, bv, lane, prow, tx0 >> 1, co_b, !bias_in_acc);
                    });
                }
            } else {
            if (p.img_scale != nullptr) {                                // two-group batch: the item's image picks the scale (uniform)
                const float sc = p.img_scale[n >= p.img_split ? 1 : 0];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NFR; ++j) acc[i][j] *= sc;
            }
            // conv_pp.hip's FAST epilogue, four fragments (two rows) at a time: one pass per operand, its loads in flight together
            if (wide) {
                const long off0 = pix0 * p.ldy + co_b;
                auto foff = [&](int j) { return off0 + ((long)(j / FW) * W + (j % FW) * 16) * p.ldy; };
                if (!bias_in_acc && p.bias != nullptr) {
                    float t[16];
                    Wide16<float>::ld(p.bias + co_b, t);
#pragma unroll
                    for (int c = 0; c < 16; ++c)
#pragma unroll
                        for (int jj = 0; jj < NFR; ++jj) acc[c >> 2][jj][c & 3] += t[c];
                }
                static_for<NFR / 4>([&](auto gq) {
                    constexpr int j0 = decltype(gq)::value * 4;
                    auto with_operand = [&](const T* src, auto&& apply) {
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
                                acc[k >> 1][j0 + jj][2 * (k & 1)] = apply(acc[k >> 1][j0 + jj][2 * (k & 1)], h16_lo_to_f32(w));
                                acc[k >> 1][j0 + jj][2 * (k & 1) + 1] = apply(acc[k >> 1][j0 + jj][2 * (k & 1) + 1], h16_hi_to_f32(w));
                            }
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
                            for (int jj = 0; jj < 4; ++jj) {
                                const f32x4_t sv = acc[i][j0 + jj] * 0.2f;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float av = acc[i][j0 + jj][r], s1 = sv[r];
                                    float mv;
                                    asm("v_max_f32 %0, %1, %2" : "=v"(mv) : "v"(av), "v"(s1));
                                    acc[i][j0 + jj][r] = mv;
                                }
                            }
                    } else if (p.act == SP_ACT_RELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[i][j0 + jj][r] = fmaxf(acc[i][j0 + jj][r], 0.f);
                    }
                    static_for<4>([&](auto jc) {
                        constexpr int j = j0 + decltype(jc)::value;
                        unsigned w8[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) w8[k] = f32x2_to_bf16x2(acc[k >> 1][j][2 * (k & 1)], acc[k >> 1][j][2 * (k & 1) + 1]);
                        T* q = reinterpret_cast<T*>(p.y) + foff(j);
                        *reinterpret_cast<uint4*>(q) = make_uint4(w8[0], w8[1], w8[2], w8[3]);
                        *reinterpret_cast<uint4*>(q + 8) = make_uint4(w8[4], w8[5], w8[6], w8[7]);
                    });
                });
            }
            }
            }                                               // run_epilogue
            ++it;
            cur = nxt;
            cur_is_tail = nxt_is_tail;
            kc = cur_is_tail ? t_k0 : 0;
            kc_end = cur_is_tail ? t_k1 : kchunks;
            nxt = item_coords(it + 1, nxt_is_tail);
            bias_fetch(more_chunks && !(cur_is_tail && !t_owner), cur.co_i * CO_T);
            acc_from_bias();
            stamp(5);
            if (half_b) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        } else {
            ++kc;
        }
    }
    if (!half_b) __builtin_amdgcn_s_barrier();              // the barrier the other half passes after its last MFMA segment
    if constexpr (TIMING) {
        if (lane == 0 && p.workspace != nullptr) {
            float* out = reinterpret_cast<float*>(p.workspace) + ((long)blockIdx.x * 8 + wave) * 16;
            for (int k = 0; k < 8; ++k) out[k] = (float)tacc[k];
        }
    }
}

template <typename T, bool TIMING, bool DMA_LB = true, bool POOL = false>
int launch_ppw(const sp_conv_params& p, int prio, hipStream_t s) {
    using G = PWGeom<T>;
    static_assert(G::LDS <= 163840, "LDS budget");
    static bool attr_set = false;
    auto kern = conv3x3_ppw_kernel<T, TIMING, DMA_LB, POOL>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", G::LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        attr_set = true;
    }
    const int cotiles = (p.cout + G::CO_T - 1) / G::CO_T;
    const int total = p.n * (p.h / G::TH) * (p.w_ / G::TW) * cotiles;
    // persistent: one block per CU; the items of a last, partial round split along K where the caller lent the scratch
    const PWSplit sk = pw_split_plan(total, (p.cin_p + G::KC - 1) / G::KC, (!TIMING && p.workspace != nullptr && p.split_sync != nullptr) ? p.workspace_bytes : 0);
    sp_note_route("conv3x3_ppw<16bit> (64 co x 4 rows per wave)");
    hipLaunchKernelGGL(kern, dim3((unsigned)sk.grid), dim3(512), G::LDS, s, p, cotiles, total, prio, sk.parts | (sp_tune(SP_TUNE_CONV_PP_SPLIT, 1) == 3 ? 256 : 0));
    SP_LAUNCH_CHECK();
    return SP_OK;
}

}  // namespace

// conv_igemm.hip's dispatch(): 16-bit 3x3 layers with more than 64 output channels on 16 x 32-pixel patches whose epilogue is the
// FAST one.  Returns 1 if the shape is not covered (the caller then keeps its own kernel).
int sp_conv_ppw_covers(const sp_conv_params& p) {
    if (p.dtype != SP_BF16 || p.ksize != 3 || p.cout <= 64) return 0;
    if (p.h % 16 != 0 || p.w_ % 32 != 0) return 0;
    if ((long)p.n * p.h * p.w_ * p.cin_p * 2 >= (1L << 30) || (long)p.cout * 9 * p.cin_p * 2 >= (1L << 30)) return 0;
    return !((p.cout & 15) != 0 || (p.ldy & 7) != 0 || p.act == SP_ACT_TANH || p.tail_w != nullptr || p.pool_idx != nullptr || p.y == nullptr);
}

// sp_conv2d_workspace(): scratch for this kernel's K-split (0: it would not split / does not cover the dims)
long sp_conv_ppw_split_workspace(int n, int h, int w, int cin_p, int cout) {
    if (h % 16 != 0 || w % 32 != 0 || cout <= 64) return 0;
    const long total = (long)n * (h / 16) * (w / 32) * ((cout + 127) / 128);
    if (total >= (1L << 30)) return 0;
    const PWSplit sk = pw_split_plan((int)total, (cin_p + 31) / 32, 1L << 40);
    return (long)sk.tail_items * (sk.parts > 1 ? sk.parts : 0) * PW_SK_SLAB_FLOATS * 4;
}

// dispatch(): cost of `total` 16-row items in hundredths of ONE item's time on every CU (whole rounds without the split)
long sp_conv_ppw_rounds100(long total, int cin_p, long workspace_bytes) {
    const int kchunks = (cin_p + 31) / 32;
    if (total < (1L << 30)) {
        const PWSplit sk = pw_split_plan((int)total, kchunks, workspace_bytes);
        if (sk.parts > 1 && total >= PW_NUM_CU)
            return 100 * (total / PW_NUM_CU) + 100 * ((kchunks + sk.parts - 1) / sk.parts) / kchunks + 100 * (total >= 2 * PW_NUM_CU ? 60 : 90) * (sk.parts - 1) / (67 * kchunks) + 1;
    }
    return 100 * ((total + PW_NUM_CU - 1) / PW_NUM_CU);
}

int sp_conv_ppw_launch(const sp_conv_params& p, hipStream_t s) {
    if (!sp_conv_ppw_covers(p)) return 1;
    const int prio = sp_tune(SP_TUNE_CONV_PP_PRIO, 1);
    if (p.pool2) return launch_ppw<bf16, false, true, true>(p, prio, s);
    if ((prio & 4) && p.workspace != nullptr && p.workspace_bytes >= 256L * 8 * 16 * 4) return launch_ppw<bf16, true>(p, prio, s);
    if (prio & 64) return launch_ppw<bf16, false, false>(p, prio, s);      // (A/B: the requests in L_A, one behind every three reads)
    return launch_ppw<bf16, false>(p, prio, s);
}
