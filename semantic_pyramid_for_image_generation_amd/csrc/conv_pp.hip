// 3x3 convolution (forward / input gradient), Cout > 64: the "ping-pong" kernel.
//
// Same tile, LDS image and epilogues as conv3x3_tall_kernel<T, 2, TH> (conv_igemm.hip): a block of 8 waves owns 128 output
// channels x (TH x 32) pixels, wave (wco, wpx) = 64 channels x (TH/4 rows x 32 columns); K is walked in 64-byte channel chunks,
// one STAGE = one tap column (3 taps) of a chunk; both operands arrive by LDS-DMA (buffer_load ... lds); the block is
// persistent and the DMA stream runs across work items.  What is new is the SCHEDULE (MI355X_MICROARCH.md, "Two waves per
// SIMD"; cdna_hip_programming.md, 8-phase template):
//
//   * the tall kernel opens every stage with `s_waitcnt vmcnt(0); s_barrier` for all eight waves, so the two waves of a SIMD
//     request together, read LDS together and multiply together (matrix || matrix, memory || memory: its matrix pipe is busy
//     44 % of the time, profiles/README.md).  Here a wave alternates a LOAD segment (all fragment reads of the coming MFMA
//     segment + its share of the LDS-DMA requests + the counted wait for the data of the NEXT stage) with an MFMA segment
//     (48 back-to-back MFMAs on registers only), one barrier between segments, and the two halves of the block (waves 0-3 /
//     4-7 = the two waves of every SIMD) run ONE SEGMENT APART: while one wave of a SIMD multiplies, its partner loads.
//
//        interval   0        1        2        3        4
//        waves 0-3  L(g)     M(g)     L(g+1)   M(g+1)   L(g+2)  ...
//        waves 4-7  M(g-1)   L(g)     M(g)     L(g+1)   M(g+1)  ...
//
//   * vmcnt is never drained in the loop.  Weights live in a 4-slot ring and are requested THREE stages ahead, the halo of
//     chunk c+1 during the first two stages of chunk c; the requests sit in the LOAD segment, one LDS-DMA instruction behind
//     every four fragment reads (the reads queue on the LDS pipe, the requests on the texture path; inside the MFMA segment they
//     cost the matrix stream ~45 cycles each - measured, SP_TUNE_CONV_PP_PRIO history in profiles/README.md).  Every wave issues
//     the same number of wave-instructions per stage (requests past the end of the block's work, and halo pieces past the end of
//     the tile, go to a dummy LDS kilobyte with an out-of-range offset = zero fill, no memory traffic), so every wait is
//     `vmcnt(pieces issued since)` with a compile-time count.
//   * item end: BOTH halves run their epilogue in the same barrier interval (the half that runs behind: before the barrier that
//     follows its last MFMA segment), and the epilogue the launcher can promise whole 16-channel groups for is one pass per
//     operand over the lane's 64 values (FAST) - see the comments at the kernel and at `item_ends`.
//   * hazards by construction: a LOAD segment ends with `lgkmcnt(0)` BEFORE its barrier, so when any wave passes barrier k all
//     LDS reads issued before it have returned; a ring slot is re-requested >= 1 barrier after its last read (WAR), and data is
//     read >= 1 barrier after the counted wait of EVERY wave that requested a piece of it (RAW; the half that runs ahead reads
//     two barriers after its own wait).
//   * the bias vector sits in LDS (copied once per block): the accumulators of an item start at the bias without a global
//     load whose compiler-inserted wait would drain the DMA queue.
#include "conv_common.h"

namespace {

constexpr int PP_TW_WIDE = 32;
constexpr int PP_NUM_CU = 256;                 // MI355X
constexpr int PP_BIAS_MAX = 1024;              // output channels whose bias fits the LDS copy

// ---- K-split of the LAST, partial round of work items ("tail split", round 6) ----
// The block is persistent with one block per CU, so a launch takes ceil(items / 256) rounds: 320 items (512 -> 512 on 32 x 32 maps at
// batch 20) cost two rounds with 3/4 of the chip idle in the second; at the metric's batch of 20 most mid-network layers sit at 1.25 /
// 2.5 rounds, which batch 32 does not (profiles/README.md, round 6).  With sk_parts = P > 1 the R = items mod 256 items of the last round
// are cut into P pieces along K (consecutive ranges of channel chunks; P from a cost model, pp_split_plan) and piece q of tail item j
// goes to the block with the physical index j * P + q - the DMA stream runs into a piece like into any other item.  Pieces q < P - 1 are
// their block's FIRST piece of work, piece P - 1 its block's LAST.  The hand-over is WAIT-FREE: a piece stores its raw fp32
// accumulators in its own slab of the caller's scratch (sp_conv_params.workspace), lets them land (`s_waitcnt vmcnt(0)`) and raises a
// per-(item, wave) counter; the wave that raises it LAST adds the P slabs in a fixed order (results do not depend on timing) and
// runs the item's ordinary epilogue - normally the closing piece, which arrives an item after the others, finds the counter full
// and keeps its own accumulators in registers.  Nobody ever waits for another block: a version in which the closing piece spun on
// the counter was 0.5 us per launch faster (the early pieces raised their counters one chunk later, off the critical path) and
// DEADLOCKED when two processes shared one GPU - each kernel's spinning blocks kept the other's not-yet-dispatched pieces off the
// CUs (tests/test_gpu_two_ranks.py hung).  (Before that, every piece ran LAST in its block: store -> signal -> load round trips at the
// very end of the launch, 10 - 20 us, more than the split saves below K = 512.)  Slabs and counters move with agent-scope relaxed
// atomics (sc1: through the XCD's L2 to the memory side - the L2s of different XCDs are not coherent for plain accesses); the wave that
// runs the epilogue zeroes the counter again, so the counters (sp_conv_params.split_sync: the caller's zero-at-rest area, one per
// stream - the library keeps no device state) are clean whenever no launch is in flight.
constexpr int PP_SK_MAX_PARTS = 4;
constexpr int PP_SK_SLAB_FLOATS = 8 * 64 * 64;           // one piece: 8 waves x 64 lanes x 64 accumulator registers = 128 KB

struct PPSplit { int parts, tail_items, grid; };
// the plan for `total` items of `kchunks` chunks each; parts <= 1: no split (grid: the unsplit launch's).  total < 256 (less than one
// round: e.g. 80 items of 128 co x 16 x 16 px for 512 -> 512 on 16 x 16 maps at batch 20): every item is a tail item and the grid is
// tail_items * parts blocks of one piece each (SP_TUNE_CONV_PP_SPLIT = 2 keeps the split to launches of at least one full round).
inline PPSplit pp_split_plan(int total, int kchunks, long workspace_bytes) {
    PPSplit r{0, 0, total < PP_NUM_CU ? total : PP_NUM_CU};
    // (less than one round: one block per item - rounded down to a multiple of 8 for the XCD remap, 100 items became 96 blocks of
    // which four took two items, i.e. two rounds; the kernels skip the remap when the grid is not a multiple of 8)
    const int mode = sp_tune(SP_TUNE_CONV_PP_SPLIT, 1);
    if (!mode || total <= 0 || (total < PP_NUM_CU && mode == 2)) return r;
    const int R = total % PP_NUM_CU;
    if (R == 0) return r;
    int pmax = PP_NUM_CU / R;
    if (pmax > PP_SK_MAX_PARTS) pmax = PP_SK_MAX_PARTS;
    // the last round takes ~3.7 us per chunk of its longest piece + 4 - 6 us per piece handed over (measured, profiles/README.md round 6:
    // the slabs of P - 1 contributors drain through the memory side, the owner fetches them one round trip each); a piece has at
    // least two chunks, and the split must save at least a twentieth of the round
    int P = 1;
    long best = 37L * kchunks;
    const long handover = total >= 2 * PP_NUM_CU ? 40 : 60;  // (two full items in front of the owner's piece hide more of it)
    for (int q = 2; q <= pmax && kchunks / q >= 2; ++q) {
        const long c = 37L * ((kchunks + q - 1) / q) + handover * (q - 1);
        if (c < best && 20 * c < 19 * 37L * kchunks) { best = c; P = q; }
    }
    if (P < 2 || (long)R * P * PP_SK_SLAB_FLOATS * 4 > workspace_bytes) return r;
    r.parts = P; r.tail_items = R;
    r.grid = total < PP_NUM_CU ? R * P : PP_NUM_CU;
    return r;
}

template <typename T, int WCO, int FW = 2>
struct PPGeom {
    static constexpr int E = 16 / (int)sizeof(T), KC = 4 * E;
    // WCO = 2: 128 co x (8 x 32) px per block, waves = 2 (co halves) x 4 (row pairs);  WCO = 1 (Cout <= 64): 64 co x (16 x 32) px,
    // waves = 8 row pairs.  Either way a wave owns 64 co x (2 rows x 32 columns): the same fragment / MFMA program.
    // FW = 16-pixel fragments per tile row: 2 = 32-pixel-wide tiles (a wave: 2 rows x 32 columns), 1 = 16-pixel-wide tiles for maps
    // 16 wide (a wave: 4 rows x 16 columns; 128 co x 16 x 16 px per block = one whole 16 x 16 image).  Same 48 MFMAs per stage.
    static constexpr int CO_T = 64 * WCO, WPX = 8 / WCO, RW = 4 / FW, TH = WPX * RW, NB = RW + 2, NFR = RW * FW, HR = TH + 2, TW = 16 * FW;
    static constexpr int HP = FW == 2 ? 40 : 20;                      // halo pitch in pixels (20: odd rows flip swizzle-key bit 1, like 36)
    static constexpr int HALO_INSTR = (HR * HP * 64 + 1023) / 1024;   // wave-instructions of 1 KB per halo chunk
    static constexpr int HALO_BUF = HALO_INSTR * 1024;
    static constexpr int HPW = (HALO_INSTR + 7) / 8;                  // ... per wave (round robin; the tail ones are dummies)
    static constexpr int HPS0 = (HPW + 1) / 2, HPS1 = HPW - HPS0;     // issued in stage 0 / stage 1 of the previous chunk
    static constexpr int W_BYTES = 3 * CO_T * 64, W_INSTR = W_BYTES / 1024, W_PER = (W_INSTR + 7) / 8;   // one weight stage: 24 / 12 wave-instructions, 3 / 2 per wave
    static constexpr int NWS = 4;                                     // weight ring slots
    static constexpr bool F8 = sizeof(T) == 1;                        // SP_F8: e4m3 operands, per-channel dequantisation scales in LDS
    static constexpr int OFF_W = 2 * HALO_BUF, OFF_BIAS = OFF_W + NWS * W_BYTES, OFF_SCALE = OFF_BIAS + PP_BIAS_MAX * 4;
    static constexpr int OFF_DUMMY = OFF_SCALE + (F8 ? PP_BIAS_MAX * 4 : 0);
    static constexpr int OFF_TAIL = OFF_DUMMY + 1024;                 // fused 1x1 tail (sp_conv_params.tail_w): [4][64] weights + [4] bias, fp32
    static constexpr int LDS_TAIL = OFF_TAIL + 4 * 64 * 4 + 16;       // the TAIL instantiation's size
    static constexpr int LDS = OFF_TAIL;
};

// number of (halo row h', tap row dr') MFMA groups that precede group (h, dr) in the MFMA segment's order (h outer, dr inner)
constexpr int pp_group_index(int h, int dr, int rw) {
    int k = 0;
    for (int hh = 0; hh <= h; ++hh)
        for (int d = 0; d < 3; ++d) {
            if (hh == h && d == dr) return k;
            if (hh - d >= 0 && hh - d < rw) ++k;
        }
    return k;
}

// FAST: the epilogue for what the launcher can promise - whole 16-channel groups (Cout % 16 == 0, ldy % 8 == 0), no pooling, no
// tanh.  The general epilogue (per-lane `wide` test, 4-channel and scalar tails, pooling, an inlined tanh per value and call site) is
// 18 K instructions in ~1 100 basic blocks around a 1.7 K-instruction loop; measured with compile-time assumptions in its place
// (scratch: -DPP_ASSUME_SIMPLE), a launch of 128->128 @128^2 drops from 170 K to 148 K cycles per block, 64->128 from 111 K to 92 K.
template <typename T, int WCO, int PRIO, bool TIMING = false, bool DMA_IN_L = true, bool FAST = false, int FW = 2, bool TAIL = false, bool IDX = false>
__global__ __launch_bounds__(512) void conv3x3_pp_kernel(sp_conv_params p, int cotiles, int total, int prio, int sk_arg) {
    const int sk_parts = sk_arg & 255;                      // pieces per tail item (0 / 1: no split); bit 8: the closing piece does not peek (tests)
    const bool sk_peek = !(sk_arg & 256);
    static_assert(!IDX || (!FAST && sizeof(T) == 2), "pool_idx: the general 16-bit epilogue");
    static_assert(!TAIL || (FAST && WCO == 1 && FW == 2), "the fused 1x1 tail lives in the 64-channel FAST form");
    using G = PPGeom<T, WCO, FW>;
    static_assert(FW == 2 || (sizeof(T) == 2 && FAST), "16-wide tiles: bf16, FAST epilogue (no pooling)");
    constexpr int PP_TW = G::TW;
    static_assert(!FAST || sizeof(T) == 2, "FAST epilogue: bf16");
    constexpr int E = G::E, KC = G::KC, CO_T = G::CO_T, WPX = G::WPX, RW = G::RW, NB = G::NB, NFR = G::NFR, HR = G::HR, HP = G::HP, TH = G::TH;
    constexpr int HALO_INSTR = G::HALO_INSTR, HALO_BUF = G::HALO_BUF, HPW = G::HPW, W_BYTES = G::W_BYTES, W_PER = G::W_PER, W_INSTR = G::W_INSTR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / WPX, wpx = wave % WPX;
    const bool half_b = wave >= 4;                          // the half of the block that runs one segment behind
    const int H = p.h, W = p.w_, CIN = p.cin_p;
    const int tiles_x = W / PP_TW, tiles_y = H / TH;
    const int kchunks = (CIN + KC - 1) / KC;
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    // XCD-aware order (as the tall kernel): blocks of one XCD get consecutive work items, co-tiles of a patch adjacent
    const int GR = gridDim.x;
    int bid = blockIdx.x;
    if ((GR & 7) == 0) bid = (bid & 7) * (GR >> 3) + (bid >> 3);
    // tail split (see the top of the file): the first full_total items go round robin as ever, the rest in K pieces
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
    const bool t_owner = t_part == sk_parts - 1;            // the piece that adds the others and runs the epilogue
    const int nchunks = my_items * kchunks + (has_tail ? t_k1 - t_k0 : 0);
    if (nchunks <= 0) return;

    // ---- bias -> LDS (fp32, zero padded to whole co-tiles), before the first LDS-DMA is in flight
    {
        float* bias_l = reinterpret_cast<float*>(smem + G::OFF_BIAS);
        const int nb = cotiles * CO_T < PP_BIAS_MAX ? cotiles * CO_T : PP_BIAS_MAX;
        for (int i = tid; i < nb; i += 512) bias_l[i] = (p.bias != nullptr && i < p.cout) ? p.bias[i] : 0.f;
        if constexpr (TAIL) {                                // (its own instantiation: compiled into the plain FAST form it cost every
                                                             // launch of the 64-channel kernel 4 %)
            if (p.tail_w != nullptr) {                       // fused 1x1 tail: weights [tail_cout][64] (16-bit, the 1x1 layer's forward packing) -> fp32
                float* tail_l = reinterpret_cast<float*>(smem + G::OFF_TAIL);
                const T* tw = reinterpret_cast<const T*>(p.tail_w);
                for (int i = tid; i < 4 * 64; i += 512) tail_l[i] = (i >> 6) < p.tail_cout ? Elem<T>::ld(tw + i) : 0.f;
                if (tid < 4) tail_l[4 * 64 + tid] = (p.tail_bias != nullptr && tid < p.tail_cout) ? p.tail_bias[tid] : 0.f;
            }
        }
        if constexpr (G::F8) {                               // dequantisation scale of (x, w[co]) per output channel
            float* scale_l = reinterpret_cast<float*>(smem + G::OFF_SCALE);
            const float sx = p.x_scale[0];
            for (int i = tid; i < nb; i += 512) scale_l[i] = i < p.cout ? sx * p.w_scale[i] : 0.f;
        }
    }

    // ---- DMA descriptors (raw buffers: SGPR base + 32-bit byte offset per lane; an offset beyond num_records reads zeros)
    constexpr unsigned OOB = 0x80000000u, OOB_C = 0x40000000u;
    const int up = p.in_up2 ? 1 : 0;
    const int HS = H >> up, WS = W >> up;
    // descriptor inputs through readfirstlane: with the words provably wave-uniform the requests are plain buffer_load ... lds;
    // otherwise hipcc parks descriptor words in VGPRs under SGPR pressure and wraps every request in a waterfall loop
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
    // work item -> (co-tile, patch column, patch row, image): item k of this block is number bid + k * GR; the mixed-radix digits
    // advance by the digits of GR with carries (a dozen scalar instructions per item instead of three integer divisions per use)
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
    // item number `it` of this block: its my_items full items (round robin) with the tail piece in front of them (a contributing piece)
    // or behind them (the piece that owns the item's epilogue)
    int it = 0;
    const int t_pos = has_tail ? (t_owner ? my_items : 0) : -1;
    Coords strided = cur;                                   // the next full item to hand out
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
        const int n = c.n, ty0 = c.ty_i * TH, tx0 = c.tx_i * PP_TW;
        int l4 = lane >> 2;
        asm volatile("" : "+v"(l4));                       // recomputed per item: hoisted out of the loop, the (row, column) pairs of
                                                           // every piece stay live across it (and spill)
#pragma unroll
        for (int i = 0; i < HPW; ++i) {
            const int hp = (i * 8 + wave) * 16 + l4;                   // piece q = i * 8 + wave
            const int hy = hp / HP, hx = hp - hy * HP;
            const int yy = ty0 - 1 + hy, xx = tx0 - 1 + hx;
            const bool ok = hx < PP_TW + 2 && hy < HR && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
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
    // one halo piece (index i of this wave's share of a chunk) / one weight piece; !valid: a dummy request (same count)
    // (selects written as masks: with `?:` on the wave-uniform condition hipcc branches around every request, which cuts the
    // MFMA segment into basic blocks)
    auto issue_halo_piece = [&](int i, bool valid, int c0, int slot) {
        const unsigned add = c0 + ls < CIN ? (unsigned)(c0 * (int)sizeof(T)) : OOB_C;
        const int q = i * 8 + wave;
        const unsigned m = (valid && q < HALO_INSTR) ? 0xffffffffu : 0u;    // wave-uniform
        dma(x_rsrc, ((unsigned)(slot * HALO_BUF + q * 1024) & m) | ((unsigned)G::OFF_DUMMY & ~m), ((h_off[i] + add) & m) | (OOB & ~m));
    };
    auto issue_w_piece = [&](int i, bool valid, int c0, int ds, int slot) {
        const unsigned base = (unsigned)((ds * CIN + c0) * (int)sizeof(T));
        const unsigned add = c0 + w_ls(i) < CIN ? base : OOB_C;
        const unsigned m = (valid && wave * W_PER + i < W_INSTR) ? 0xffffffffu : 0u;   // (WCO = 1: 12 pieces on 16 slots)
        dma(w_rsrc, ((unsigned)(G::OFF_W + slot * W_BYTES + (wave * W_PER + i) * 1024) & m) | ((unsigned)G::OFF_DUMMY & ~m),
            ((w_off[i] + add) & m) | (OOB & ~m));
    };

    // ---- fragment read addresses (the tall kernel's: permuted A rows -> a lane ends with 16 consecutive channels of its pixel)
    const int frow = lane & 15, fslot = lane >> 4;
    const unsigned a_addr = lds_base + G::OFF_W + (wco * 64 + (frow >> 2) * 16 + (frow & 3)) * 64 +
                            ((fslot ^ (((frow >> 1) & 1) | (((frow >> 2) & 1) << 1))) << 4);
    unsigned b_addr[3];
#pragma unroll
    for (int ds = 0; ds < 3; ++ds)
        b_addr[ds] = lds_base + ((RW * wpx) * HP + frow + ds) * 64 + ((fslot ^ (((frow + ds) >> 1) & 3)) << 4);
    const unsigned bias_addr = lds_base + G::OFF_BIAS + (wco * 64 + (lane >> 4) * 16) * 4;

    const bool bias_in_acc = !G::F8 && p.bias != nullptr && !up && cotiles * CO_T <= PP_BIAS_MAX && p.img_scale == nullptr;   // (SP_F8 / two-group batches: the bias follows the scale)
    f32x4_t acc[4][NFR];
    // accumulators of an item start at its bias (LDS copy: no global load whose wait would drain the DMA queue).  Measured and not
    // kept: the first stage of an item taking the bias registers as the MFMA C operand instead of 64 moves (a fourth copy of the
    // stage code: step 17.63 vs 17.59 ms)
    uint4 b4[4];
    auto bias_fetch = [&](bool live, int co0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) b4[i] = make_uint4(0, 0, 0, 0);
        if (bias_in_acc && live) {
            const unsigned ba = bias_addr + (unsigned)co0 * 4u;
            lds_rd128<0>(b4[0], ba); lds_rd128<16>(b4[1], ba); lds_rd128<32>(b4[2], ba); lds_rd128<48>(b4[3], ba);
            // RULE for every asm LDS read in this file: its lgkmcnt wait follows in the same straight-line region.  The compiler does
            // not know that the destination registers are written LATER: with a branch or a barrier in between it reused registers of
            // a read still in flight for address arithmetic (values it considered dead on that path) and the late data overwrote
            // them - a launch wrong once in a few hundred (round 3: pixel fragments read ahead across a barrier; the variant was
            // also slower once it waited correctly, and is gone).  tools/check_async_lds.py screens the assembly for this.
            wait_lgkm<0>();
        }
    };

    // ---- prologue: chunk 0's halo, weight stages 0, 1 and 2
    set_halo_desc(cur);
    set_w_desc(cur);
    int kc = cur_is_tail ? t_k0 : 0;                        // chunk of the current item; a tail piece starts inside its item
    int kc_end = cur_is_tail ? t_k1 : kchunks;
#pragma unroll
    for (int i = 0; i < HPW; ++i) issue_halo_piece(i, true, kc * KC, 0);
#pragma unroll
    for (int ds = 0; ds < 3; ++ds)
#pragma unroll
        for (int i = 0; i < W_PER; ++i) issue_w_piece(i, true, kc * KC, ds, ds);
    wait_vmcnt<2 * W_PER>();                                // halo 0 and stage 0 landed (this wave's pieces); stages 1, 2 may fly
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the bias copy
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    bias_fetch(!(cur_is_tail && !t_owner), cur.co_i * CO_T);   // (a tail piece that only contributes starts from zero)
    auto acc_from_bias = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NFR; ++j) acc[i][j] = __builtin_bit_cast(f32x4_t, b4[i]);
    };
    acc_from_bias();
    // Measured and not kept (profiles/README.md): ONE barrier per stage (the leading half synchronises only behind its MFMA segment,
    // the other only behind its LOAD segment; every hazard still has a barrier in between) - the half that multiplies first in an
    // interval loses the matrix pipe to its partner's tail (MFMA segment 780 -> 1 050 cycles), full step 18.19 vs 18.01 ms.
    if (half_b) __builtin_amdgcn_s_barrier();               // from here on this half runs one segment behind

    if ((prio & 2) && half_b) __builtin_amdgcn_s_setprio(1);
    // TIMING build (diagnostics, SP_TUNE_CONV_PP_PRIO bit 2): cycles per wave in [0] LOAD segment up to the counted wait (fragment
    // reads + DMA issue + LDS latency), [1] the vmcnt wait, [2] barrier after LOAD, [3] MFMA segment, [4] barrier after MFMA,
    // [5] epilogue + item switch; written to p.workspace[(block * 8 + wave) * 8 + k] (fp32 scratch pointer, unused by this kernel)
    unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    bool stamp_on = true;
    auto stamp = [&](int k) {
        if constexpr (TIMING) {
            const unsigned long long t = __builtin_readcyclecounter();
            if (!(prio & 512) || stamp_on) tacc[k] += t - tprev;     // bit 9: only the chunks in the middle of an item
            tprev = t;
        }
    };
    if constexpr (TIMING) tprev = __builtin_readcyclecounter();
    float vmax = 0.f;                                       // SP_F8: running max of this lane's outputs (merged once per block at the end)
    int g4 = 0;                                             // weight ring slot of the stage being computed (stage index mod 4)
    const bool vec_ok = ((p.ldy & 3) == 0) && ((p.cout & 3) == 0);
    for (int gc = 0; gc < nchunks; ++gc) {
        const bool more_chunks = gc + 1 < nchunks;
        const bool item_ends = kc + 1 == kc_end;
        if constexpr (TIMING) stamp_on = kc != 0 && !item_ends;
        const unsigned hb = (unsigned)((gc & 1) * HALO_BUF);
        const int c0_next = item_ends ? (nxt_is_tail ? t_k0 * KC : 0) : (kc + 1) * KC;
        auto stage = [&](auto sc) {
            constexpr int st = decltype(sc)::value;        // stage inside the chunk = tap column
            constexpr int NREAD = 12 + FW * NB;
            constexpr int TAP_STRIDE = CO_T * 64;
            constexpr int NH = st == 0 ? G::HPS0 : st == 1 ? G::HPS1 : 0, H0 = st == 0 ? 0 : G::HPS0;   // halo pieces requested in this stage
            constexpr int NPIECE = NH + W_PER;
            const unsigned ab = a_addr + (unsigned)(g4 * W_BYTES);
            const unsigned bb = b_addr[st] + hb;
            const unsigned bo = (HP == 36 || HP == 20) ? (bb ^ 32u) : bb; // odd halo rows (pitch 36 / 20): swizzle key flipped in bit 1
            uint4 a[3][4], bf[NB][FW];
            // ================= LOAD segment: every fragment of the stage =================
            const int ws = (g4 + 3) & 3;                    // slot of stage g + 3 = slot of stage g - 1 (its reads ended >= two barriers ago)
            auto piece = [&](auto kc_) {                    // request number k of this stage: halo of the next chunk, then weights of stage g + 3
                constexpr int k = decltype(kc_)::value;
                if constexpr (k < NH) issue_halo_piece(H0 + k, more_chunks, c0_next, (gc + 1) & 1);
                else if constexpr (k < NPIECE) issue_w_piece(k - NH, more_chunks, c0_next, st, ws);
            };
            if constexpr (DMA_IN_L && st == 0) {
                if (item_ends && more_chunks) { set_halo_desc(nxt); set_w_desc(nxt); }
            }
            // DMA_IN_L: one request behind every four fragment reads - the reads queue on the LDS pipe, the requests on the texture
            // path, so neither waits for the other's queue to drain
            static_for<((NREAD + 3) / 4 > NPIECE ? (NREAD + 3) / 4 : NPIECE)>([&](auto qc) {
                constexpr int q0 = decltype(qc)::value * 4;
                static_for<4>([&](auto rc) {
                    constexpr int r = q0 + decltype(rc)::value;     // read number: 0..11 = A (tap row r / 4, fragment r % 4), then B rows
                    if constexpr (r < 12) {
                        lds_rd128<(r / 4) * TAP_STRIDE + (r % 4) * 256>(a[r / 4][r % 4], ab);
                    } else if constexpr (r < NREAD) {
                        constexpr int h = (r - 12) / FW, hh = (r - 12) % FW;
                        lds_rd128<h * (HP * 64) + hh * 1024>(bf[h][hh], (h & 1) ? bo : bb);
                    }
                });
                if constexpr (DMA_IN_L) piece(qc);
            });
            stamp(0);
            // the NEXT stage's weights were requested two MFMA segments ago, everything younger in the last one: H + W pieces of the
            // previous stage; before a chunk's first stage the wait also covers the last halo pieces (3 younger weight pieces)
            if constexpr (DMA_IN_L)     // requests of this and of the previous LOAD segment are younger than the next stage's weights
                wait_vmcnt<st == 0 ? 2 * W_PER + G::HPS0 : st == 1 ? 2 * W_PER + G::HPS0 + G::HPS1 : 2 * W_PER>();
            else
                wait_vmcnt<st == 0 ? W_PER : st == 1 ? G::HPS0 + W_PER : W_PER>();
            stamp(1);
            wait_lgkm<0>();                                 // every LDS read of this wave has returned before it signals
            stamp(7);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp(2);
            // ================= MFMA segment, with this stage's requests behind its MFMA groups =================
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
            if constexpr (!DMA_IN_L && st == 0) {
                if (item_ends && more_chunks) { set_halo_desc(nxt); set_w_desc(nxt); }
            }
            static_for<NB * 3>([&](auto gi) {
                constexpr int h = decltype(gi)::value / 3, dr = decltype(gi)::value % 3, rr = h - dr;
                if constexpr (rr >= 0 && rr < RW) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int hh = 0; hh < FW; ++hh) Mma<T>::run(a[dr][i], bf[h][hh], acc[i][rr * FW + hh]);
                    if constexpr (!DMA_IN_L) piece(std::integral_constant<int, pp_group_index(h, dr, RW)>{});   // behind MFMA group k: request k
                }
            });
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            stamp(3);
            // (item end, the half that runs behind: its epilogue comes BEFORE this barrier - see below)
            if (!(st == 2 && item_ends && half_b)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp(4);
            g4 = (g4 + 1) & 3;
        };
        stage(std::integral_constant<int, 0>{});
        stage(std::integral_constant<int, 1>{});
        stage(std::integral_constant<int, 2>{});
        if (item_ends) {
            // BOTH epilogues in one barrier interval.  After its last MFMA segment the leading half passes the barrier, runs its
            // epilogue and the next item's first LOAD segment; the other half is in its last MFMA segment meanwhile and would then
            // wait ~3 000 cycles at the barrier - and hold the leading half up for just as long one interval later, with its own
            // epilogue.  So it runs its epilogue right behind its MFMAs, in front of the barrier: the interval takes
            // max(M + E, E + L) instead of two intervals of E + L each.
            const int n = cur.n, ty0 = cur.ty_i * TH, tx0 = cur.tx_i * PP_TW, co0 = cur.co_i * CO_T;
            const long pix0 = ((long)n * H + ty0 + RW * wpx) * W + tx0 + (lane & 15);
            const int co_b = co0 + wco * 64 + (lane >> 4) * 16;          // this lane's 16 consecutive channels
            const bool wide = FAST ? co_b < p.cout : vec_ok && (p.ldy & 7) == 0 && co_b + 16 <= p.cout;
            bool run_epilogue = true;
            if (cur_is_tail && sk_parts > 1) {                           // wave-uniform
                // tail split, WAIT-FREE: nobody ever waits for another block (two kernels that spin for blocks of their own which the
                // other one keeps off the CUs deadlock - seen with two ranks sharing one GPU).  A piece stores its raw accumulators in
                // ITS slab, lets the stores land (vmcnt(0)) and raises the item's per-wave counter; whoever raises it last sums the P
                // slabs in the fixed order P - 1, 0, 1, ..., P - 2 and runs the epilogue.  The closing piece (P - 1, its block's last
                // work) looks first: with every other piece in it keeps its accumulators (the usual case - the others arrived an item
                // ago), which is that same order.
                int woff = wave * 4096 + lane;
                asm volatile("" : "+v"(woff));                           // (computed HERE: hoisted out of the chunk loop, the sixty-odd
                                                                         // addresses below stay live across it and spill)
                float* slab0 = reinterpret_cast<float*>(p.workspace) + ((long)t_j * sk_parts) * PP_SK_SLAB_FLOATS + woff;
                int* flag = p.split_sync + t_j * 8 + wave;
                bool fast = false, last = false;
                if (t_owner && sk_peek) fast = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sk_parts - 1;
                if (!fast) {
                    float* dst = slab0 + (long)t_part * PP_SK_SLAB_FLOATS;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < NFR; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                __hip_atomic_store(dst + ((i * NFR + j) * 4 + r) * 64, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the slab is at the memory side before the counter says so
                    int old = 0;
                    if (lane == 0) old = __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = __builtin_amdgcn_readfirstlane(old) == sk_parts - 1;
                }
                run_epilogue = fast || last;
                // slabs to add, in order: the closing piece's own (kept in registers on the fast path, re-read otherwise), then 0 .. P - 2
                const int q_first = fast ? 0 : -1, q_end = run_epilogue ? sk_parts - 1 : -1;
#pragma unroll 1
                for (int q = q_first; q < q_end; ++q) {
                    const float* src = slab0 + (long)(q < 0 ? sk_parts - 1 : q) * PP_SK_SLAB_FLOATS;
                    // all 64 loads of a slab in flight: one round trip to the memory side per slab; whole fragments (the accumulators are
                    // 4-register tuples: element-wise updates cost conv_ppw.hip, with its 128 accumulators, 300 spilled registers)
                    f32x4_t t[4 * NFR];
#pragma unroll
                    for (int k = 0; k < NFR * 16; ++k)
                        t[k >> 2][k & 3] = __hip_atomic_load(src + k * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (q < 0) {
#pragma unroll
                        for (int f = 0; f < 4 * NFR; ++f) acc[f / NFR][f % NFR] = t[f];
                    } else {
#pragma unroll
                        for (int f = 0; f < 4 * NFR; ++f) acc[f / NFR][f % NFR] += t[f];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (run_epilogue && lane == 0) __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
            }
            if (run_epilogue) {
            if (up) {                                                    // the 1/4 of the average-pooling gradient
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NFR; ++j) acc[i][j] *= 0.25f;
            }
            stamp(9);
            if constexpr (FAST) {
                if (p.img_scale != nullptr) {                            // two-group batch: the item's image picks the scale (uniform)
                    const float sc = p.img_scale[n >= p.img_split ? 1 : 0];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < NFR; ++j) acc[i][j] *= sc;
                }
                // ONE pass per epilogue operand over the whole 64-value tile of the lane (a handful of uniform branches per item, the
                // eight loads of an operand in flight together) - the per-fragment form (each fragment: its own tests for bias / mask /
                // residuals / activation, its own load -> wait) cost ~640 cycles per fragment, 5x its VALU work
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
                    // channel c = 4 i + r of the lane's 16 is acc[i][j][r]; word k of the 32 bytes of (pixel, 16 channels) = channels 2k, 2k + 1
                    auto with_operand = [&](const T* src, auto&& apply) {
                        uint4 t[NFR][2];
#pragma unroll
                        for (int jj = 0; jj < NFR; ++jj) {
                            const T* q = src + foff(jj);
                            t[jj][0] = *reinterpret_cast<const uint4*>(q);
                            t[jj][1] = *reinterpret_cast<const uint4*>(q + 8);
                        }
#pragma unroll
                        for (int jj = 0; jj < NFR; ++jj)
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const uint4& u = t[jj][k >> 2];
                                const uint32_t w = (k & 3) == 0 ? u.x : (k & 3) == 1 ? u.y : (k & 3) == 2 ? u.z : u.w;
                                acc[k >> 1][jj][2 * (k & 1)] = apply(acc[k >> 1][jj][2 * (k & 1)], h16_lo_to_f32(w));
                                acc[k >> 1][jj][2 * (k & 1) + 1] = apply(acc[k >> 1][jj][2 * (k & 1) + 1], h16_hi_to_f32(w));
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
                            for (int jj = 0; jj < NFR; ++jj) {
                                const f32x4_t sv = acc[i][jj] * 0.2f;                        // two packed multiplies
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float av = acc[i][jj][r], s1 = sv[r];
                                    float mv;
                                    asm("v_max_f32 %0, %1, %2" : "=v"(mv) : "v"(av), "v"(s1));   // (apply_act_vec, common.h)
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
                    if constexpr (TAIL) {
                        // fused 1x1 tail (the generator's last two layers, models.py:55-61: conv3x3 -> LeakyReLU -> conv1x1 -> tanh): the 64
                        // channels of a pixel sit in the four lanes l, l + 16, l + 32, l + 48 - each takes the dot products of its 16
                        // channels with the tail's rows (LDS copy), two butterfly steps add them up, lane group 0 stores the pixel's
                        // tail_cout values.  The 16-bit rounding of the intermediate tensor is applied first: the separate 1x1 layer reads it
                        // from memory in that precision.
                        if (p.tail_w != nullptr) {
                            const float* tail_l = reinterpret_cast<const float*>(smem + G::OFF_TAIL);
                            const int cg16 = (lane >> 4) * 16;
                            T* ty = reinterpret_cast<T*>(p.tail_y);
                            for (int o = 0; o < p.tail_cout; ++o) {
                                float wv[16];
#pragma unroll
                                for (int c4 = 0; c4 < 4; ++c4) {
                                    const float4 t4 = *reinterpret_cast<const float4*>(tail_l + o * 64 + cg16 + c4 * 4);
                                    wv[c4 * 4] = t4.x; wv[c4 * 4 + 1] = t4.y; wv[c4 * 4 + 2] = t4.z; wv[c4 * 4 + 3] = t4.w;
                                }
                                const float tb = tail_l[4 * 64 + o];
#pragma unroll
                                for (int jj = 0; jj < NFR; ++jj) {
                                    float part = 0.f;
#pragma unroll
                                    for (int k = 0; k < 8; ++k) {
                                        const uint32_t w2 = f32x2_to_bf16x2(acc[k >> 1][jj][2 * (k & 1)], acc[k >> 1][jj][2 * (k & 1) + 1]);
                                        part = fmaf(h16_lo_to_f32(w2), wv[2 * k], part);
                                        part = fmaf(h16_hi_to_f32(w2), wv[2 * k + 1], part);
                                    }
                                    part += __shfl_xor(part, 16, 64);
                                    part += __shfl_xor(part, 32, 64);
                                    if (lane < 16) {
                                        float tv = part + tb;
                                        if (p.tail_act == SP_ACT_TANH) tv = tanhf(tv);
                                        const long tpix = pix0 + (long)(jj / FW) * W + (jj % FW) * 16;
                                        Elem<T>::st(ty + tpix * p.tail_ld + o, tv);
                                    }
                                }
                            }
                        }
                    }
                    if (p.y != nullptr)
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
            } else if constexpr (G::F8) {
                // SP_F8 epilogue: v = relu(acc * scale[co] + bias[co]) (the 2x2 maximum first: it commutes with the positive scale,
                // the bias and the ReLU), stored as bf16 and / or re-quantised to e4m3 for the next layer; running max for its scale
                uint4 sc4[4], bi4[4];
                const unsigned sa = bias_addr + (unsigned)(G::OFF_SCALE - G::OFF_BIAS) + (unsigned)co0 * 4u, ba = bias_addr + (unsigned)co0 * 4u;
                lds_rd128<0>(sc4[0], sa); lds_rd128<16>(sc4[1], sa); lds_rd128<32>(sc4[2], sa); lds_rd128<48>(sc4[3], sa);
                lds_rd128<0>(bi4[0], ba); lds_rd128<16>(bi4[1], ba); lds_rd128<32>(bi4[2], ba); lds_rd128<48>(bi4[3], ba);
                wait_lgkm<0>();
                float sc[16], bi[16];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sc[i * 4] = __uint_as_float(sc4[i].x); sc[i * 4 + 1] = __uint_as_float(sc4[i].y); sc[i * 4 + 2] = __uint_as_float(sc4[i].z); sc[i * 4 + 3] = __uint_as_float(sc4[i].w);
                    bi[i * 4] = __uint_as_float(bi4[i].x); bi[i * 4 + 1] = __uint_as_float(bi4[i].y); bi[i * 4 + 2] = __uint_as_float(bi4[i].z); bi[i * 4 + 3] = __uint_as_float(bi4[i].w);
                }
                const float inv_sy = p.y8 != nullptr ? p.y8_inv_scale[0] : 0.f;
                bf16* yb = reinterpret_cast<bf16*>(p.y);
                uint8_t* y8 = reinterpret_cast<uint8_t*>(p.y8);
                const bool relu = p.act == SP_ACT_RELU;
                auto finish = [&](float (&v)[16], long opix) {          // one pixel x 16 channels of the (pooled) output
                    // a co-tile past Cout (Cout % 128 != 0: the launcher guarantees Cout % 16 == 0) has whole 16-channel groups
                    // outside the tensor: they must not be stored - they would land on channels 0.. of the NEXT pixel
                    if (co_b >= p.cout) return;
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        v[c] = v[c] * sc[c] + bi[c];
                        if (relu) v[c] = fmaxf(v[c], 0.f);
                        vmax = fmaxf(vmax, fabsf(v[c]));
                    }
                    const long off = opix * p.ldy + co_b;
                    if (yb != nullptr) Wide16<bf16>::st(yb + off, v);
                    if (y8 != nullptr) {
                        unsigned w4[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float q[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) q[e] = fminf(fmaxf(v[k * 4 + e] * inv_sy, -448.f), 448.f);
                            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
                            pk = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], pk, true);
                            w4[k] = (unsigned)pk;
                        }
                        *reinterpret_cast<uint4*>(y8 + off) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                    }
                };
                static_for<NFR>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if (p.pool2) {
                        if constexpr ((j & 3) == 0) {
                            const bool odd = lane & 1;
                            float v[16];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float a = fmaxf(acc[i][j][r], acc[i][j + 2][r]), b = fmaxf(acc[i][j + 1][r], acc[i][j + 3][r]);
                                    const float recv = dpp_xor1(odd ? a : b);
                                    v[i * 4 + r] = fmaxf(odd ? b : a, recv);
                                }
                            const long prow = ((long)n * (H >> 1) + ((ty0 + RW * wpx + (j >> 1)) >> 1)) * (W >> 1);
                            finish(v, prow + (tx0 >> 1) + (odd ? 8 : 0) + ((lane & 15) >> 1));
                        }
                    } else {
                        float v[16];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r];
                        finish(v, pix0 + (long)(j >> 1) * W + (j & 1) * 16);
                    }
                });
            } else
            static_for<NFR>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const long pix = pix0 + (long)(j >> 1) * W + (j & 1) * 16;
                if (p.pool2) {                                           // launcher guarantees `wide`
                    if constexpr ((j & 3) == 0) {                        // fragments j..j+3 = rows (j>>1, j>>1 + 1) x column halves
                        const long prow = ((long)n * (H >> 1) + ((ty0 + RW * wpx + (j >> 1)) >> 1)) * (W >> 1);
                        if constexpr (IDX) {                             // maximum + its window position (the VGG pass with gradient)
                            float a0[16], a1[16], b0[16], b1[16];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    a0[i * 4 + r] = acc[i][j][r]; a1[i * 4 + r] = acc[i][j + 2][r];
                                    b0[i * 4 + r] = acc[i][j + 1][r]; b1[i * 4 + r] = acc[i][j + 3][r];
                                }
                            conv_epilogue_pool2_idx<T>(p, a0, a1, b0, b1, lane, prow, tx0 >> 1, co_b, !bias_in_acc);
                        } else {
                            float av[16], bv[16];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    av[i * 4 + r] = pool2_combine(acc[i][j][r], acc[i][j + 2][r], p.pool2 == 2);
                                    bv[i * 4 + r] = pool2_combine(acc[i][j + 1][r], acc[i][j + 3][r], p.pool2 == 2);
                                }
                            conv_epilogue_pool2<T>(p, av, bv, lane, prow, tx0 >> 1, co_b, !bias_in_acc);
                        }
                    }
                } else if (wide) {
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r];
                    conv_epilogue16<T>(p, v, pix, co_b, !bias_in_acc);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int co = co_b + i * 4;
                        if (co < p.cout) {
                            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                            conv_epilogue4<T>(p, v, pix, co, vec_ok, !bias_in_acc);
                        }
                    }
                }
            });
            }                                               // run_epilogue
            stamp(10);
            ++it;
            cur = nxt;
            cur_is_tail = nxt_is_tail;
            kc = cur_is_tail ? t_k0 : 0;
            kc_end = cur_is_tail ? t_k1 : kchunks;
            nxt = item_coords(it + 1, nxt_is_tail);          // (past the block's last item: coordinates nobody uses)
            stamp(5);
            bias_fetch(more_chunks && !(cur_is_tail && !t_owner), cur.co_i * CO_T);   // (after the epilogue's own reads of the bias / scale tables)
            acc_from_bias();
            stamp(8);
            if (half_b) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        } else {
            ++kc;
        }
    }
    if (!half_b) __builtin_amdgcn_s_barrier();              // the barrier the other half passes after its last MFMA segment
    if constexpr (G::F8) {
        // one atomic per BLOCK on the running maximum (per item and wave they serialise on one L2 word: 10 K atomics = 100+ us)
        if (p.y8_amax != nullptr) {
            float* red = reinterpret_cast<float*>(smem + G::OFF_DUMMY);
            vmax = wave_max(vmax);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy kilobyte is the target of zero-fill requests: let them land
            __builtin_amdgcn_s_barrier();
            if (lane == 0) red[wave] = vmax;
            __syncthreads();
            if (tid == 0) {
                float m = red[0];
                for (int k = 1; k < 8; ++k) m = fmaxf(m, red[k]);
                atomicMax(reinterpret_cast<unsigned*>(p.y8_amax), __float_as_uint(m));
            }
        }
    }
    if constexpr (TIMING) {
        if (lane == 0 && p.workspace != nullptr) {
            // 16 floats per wave: [0..7] as above ([5] = the whole item switch), [8] accumulator init, [9] item-end setup (the 1/4 of
            // a pooling gradient), [10] the fragments (values, packing, stores), [11] coordinates of the next item
            float* out = reinterpret_cast<float*>(p.workspace) + ((long)blockIdx.x * 8 + wave) * 16;
            for (int k = 0; k < 12; ++k) out[k] = (float)tacc[k];
            out[11] = out[5];
            out[5] += (float)(tacc[8] + tacc[9] + tacc[10]);
        }
    }
}

template <typename T, int WCO, int PRIO, bool TIMING = false, bool DMA_IN_L = true, bool FAST = false, int FW = 2, bool TAIL = false, bool IDX = false>
int launch_pp(const sp_conv_params& p, int prio, hipStream_t s) {
    using G = PPGeom<T, WCO, FW>;
    constexpr int TH = G::TH;
    constexpr int LDS_BYTES = TAIL ? G::LDS_TAIL : G::LDS;
    static_assert(LDS_BYTES <= 163840, "LDS budget");
    static bool attr_set = false;
    auto kern = conv3x3_pp_kernel<T, WCO, PRIO, TIMING, DMA_IN_L, FAST, FW, TAIL, IDX>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", LDS_BYTES, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        attr_set = true;
    }
    const int cotiles = (p.cout + G::CO_T - 1) / G::CO_T;
    const int total = p.n * (p.h / TH) * (p.w_ / G::TW) * cotiles;
    // persistent: one block per CU; the items of a last, partial round split along K where the caller lent the scratch (top of the file)
    PPSplit sk = pp_split_plan(total, (p.cin_p + G::KC - 1) / G::KC,
                               (!G::F8 && !TIMING && !TAIL && p.workspace != nullptr && p.split_sync != nullptr) ? p.workspace_bytes : 0);
    const int grid = sk.grid;
    sp_note_route(G::F8 ? "conv3x3_pp<f8,2>" : FW == 1 ? "conv3x3_pp<16bit,2,FAST,w16>" : WCO == 2 ? (FAST ? "conv3x3_pp<16bit,2,FAST>" : "conv3x3_pp<16bit,2>")
                                                                                  : (FAST ? "conv3x3_pp<16bit,1,FAST>" : "conv3x3_pp<16bit,1>"));
    // SP_TUNE_CONV_PP_SPLIT = 3 (tests): the closing piece stores and counts like every other piece, so the re-read order runs on every split launch
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS_BYTES, s, p, cotiles, total, prio, sk.parts | (sp_tune(SP_TUNE_CONV_PP_SPLIT, 1) == 3 ? 256 : 0));
    SP_LAUNCH_CHECK();
    return SP_OK;
}

}  // namespace

// sp_conv2d_workspace(): bytes of fp32 scratch with which a 16-bit 3x3 launch of these dims can split its last round (0: it would not)
long sp_conv_pp_split_workspace(int n, int h, int w, int cin_p, int cout) {
    if (h % 8 != 0 || w % PP_TW_WIDE != 0 || cout <= 16) return 0;
    const int kchunks = (cin_p + 31) / 32;
    long need = 0;
    auto plan = [&](long total) {
        if (total >= (1L << 30)) return;
        const PPSplit sk = pp_split_plan((int)total, kchunks, 1L << 40);
        const long b = (long)sk.tail_items * (sk.parts > 1 ? sk.parts : 0) * PP_SK_SLAB_FLOATS * 4;
        if (b > need) need = b;
    };
    if (cout <= 64) { if (h % 16 == 0) plan((long)n * (h / 16) * (w / PP_TW_WIDE)); }
    else plan((long)n * (h / 8) * (w / PP_TW_WIDE) * ((cout + 127) / 128));
    return need;
}

// the same for the 16-pixel-wide tiles (128 co x 16 x 16 px per item, maps 16 wide)
long sp_conv_pp_split_workspace_w16(int n, int h, int cin_p, int cout) {
    if (h % 16 != 0 || cout <= 64 || (long)n * (h / 16) * ((cout + 127) / 128) < 64) return 0;      // (sp_conv_pp_launch's admission)
    const PPSplit sk = pp_split_plan(n * (h / 16) * ((cout + 127) / 128), (cin_p + 31) / 32, 1L << 40);
    return (long)sk.tail_items * (sk.parts > 1 ? sk.parts : 0) * PP_SK_SLAB_FLOATS * 4;
}

// dispatch(): what a launch of `total` 8-row items costs, in hundredths of the time of one item on every CU - whole rounds without the
// split; with it the longest piece of the last round plus the hand-over of the partial tiles (pp_split_plan's model)
long sp_conv_pp_rounds100(long total, int cin_p, long workspace_bytes) {
    const int kchunks = (cin_p + 31) / 32;
    if (total < (1L << 30)) {
        const PPSplit sk = pp_split_plan((int)total, kchunks, workspace_bytes);
        if (sk.parts > 1 && total >= PP_NUM_CU) return 100 * (total / PP_NUM_CU) + 100 / sk.parts + 162 * (sk.parts - 1) / kchunks + 1;
    }
    return 100 * ((total + PP_NUM_CU - 1) / PP_NUM_CU);
}

// conv_igemm.hip's dispatch(): bf16 3x3 layers with more than 64 output channels on (th x 32)-pixel patches, th = 8 or 16.
// Returns 1 if the shape is not covered (the caller then keeps its own kernel).
int sp_conv_pp_launch(const sp_conv_params& p, int th, hipStream_t s) {
    if (p.dtype == SP_F8) {
        if (p.ksize != 3 || p.cout <= 64 || p.h % 8 != 0 || p.w_ % PP_TW_WIDE != 0 || (long)p.n * p.h * p.w_ * p.cin_p >= (1L << 30) ||
            (long)p.cout * 9 * p.cin_p >= (1L << 30) || (p.cout + 127) / 128 * 128 > PP_BIAS_MAX) return 1;
        return launch_pp<f8, 2, 1>(p, 1, s);
    }
    if (p.dtype != SP_BF16 || p.ksize != 3) return 1;
    const long esz = 2;
    if (th == 1616) {
        // maps 16 wide (th code 1616): 128 co x 16 x 16 px tiles - one whole image of the 16 x 16 layers per block.  Few items (80 for
        // 512 -> 512 at batch 20), but each runs the ping-pong pipeline on a 128 x 256 tile instead of sixteen 64 x 64 tiles
        // that re-read their operands from L2 (the LDS-DMA igemm these layers used): 512 -> 512: 53.4 -> 44.6 us, 520 -> 512: 70.8 ->
        // 46.8, 256 -> 512: 29.0 -> 25.2; with 40 items (Cout 256) it is no faster - those stay on the igemm.  Split-K over 2-4 K
        // ranges (fp32 partial tiles + the finalize pass) was built and measured: 60 us - a 128 x 256 fp32 tile per 12 stages is
        // more store-path time than the extra parallelism buys
        if (p.w_ != 16 || p.h % 16 != 0 || (long)p.n * (p.h / 16) * ((p.cout + 127) / 128) < 64 || p.pool2 != 0 || (p.cout & 15) != 0 || (p.ldy & 7) != 0 || p.act == SP_ACT_TANH) return 1;
        if ((long)p.n * p.h * p.w_ * p.cin_p * esz >= (1L << 30) || (long)p.cout * 9 * p.cin_p * esz >= (1L << 30)) return 1;
        return launch_pp<bf16, 2, 1, false, true, true, 1>(p, 1, s);
    }
    if (p.h % th != 0 || p.w_ % PP_TW_WIDE != 0) return 1;
    if ((long)p.n * p.h * p.w_ * p.cin_p * esz >= (1L << 30) || (long)p.cout * 9 * p.cin_p * esz >= (1L << 30)) return 1;
    // SP_TUNE_CONV_PP_PRIO (diagnostics / A-B): bit 2 = the TIMING build (bit 9 with it: stamps of mid-item chunks only),
    // bit 4 = the general epilogue everywhere
    const int prio = sp_tune(SP_TUNE_CONV_PP_PRIO, 1);
    const bool fast = !(prio & 16) && p.pool2 == 0 && (p.cout & 15) == 0 && (p.ldy & 7) == 0 && p.act != SP_ACT_TANH;
    if (p.tail_w != nullptr) {                             // the fused 1x1 tail: its own instantiation of the 64-channel FAST form
        if (!(th == 16 && p.cout == 64 && fast)) return 1;
        return launch_pp<bf16, 1, 1, false, true, true, 2, true>(p, prio, s);
    }
    if (p.pool_idx != nullptr) {                           // ReLU + MaxPool with recorded window positions: own instantiations (general epilogue)
        if (p.pool2 != 2) return 1;
        if (th == 16 && p.cout <= 64) return launch_pp<bf16, 1, 1, false, true, false, 2, false, true>(p, prio, s);
        if (th == 8 && p.cout > 64) return launch_pp<bf16, 2, 1, false, true, false, 2, false, true>(p, prio, s);
        return 1;
    }
    if (th == 16 && p.cout <= 64) {                        // 64 co x 16x32 px
        if ((prio & 4) && fast && p.workspace != nullptr && p.workspace_bytes >= 256L * 8 * 16 * 4)
            return launch_pp<bf16, 1, 1, true, true, true>(p, prio, s);       // (TIMING build, see below)
        return fast ? launch_pp<bf16, 1, 1, false, true, true>(p, prio, s) : launch_pp<bf16, 1, 1>(p, prio, s);
    }
    if (th != 8 || p.cout <= 64) return 1;
    // the TIMING build writes 256 x 8 x 16 floats of stamps into the caller's workspace: only with a workspace that holds them
    if ((prio & 4) && p.workspace != nullptr && p.workspace_bytes >= 256L * 8 * 16 * 4)
        return fast ? launch_pp<bf16, 2, 1, true, true, true>(p, prio, s) : launch_pp<bf16, 2, 1, true>(p, prio, s);
    return fast ? launch_pp<bf16, 2, 1, false, true, true>(p, prio, s) : launch_pp<bf16, 2, 1>(p, prio, s);
}
