// NHWC implicit-GEMM convolution (1x1 and 3x3 s1 p1) on gfx950 MFMA.  Forward and input-gradient.
//
// GEMM view: D[co][px] = sum_k Wp[co][k] * Xcol[px][k],  k = (tap, ci).  The WEIGHT tile is the MFMA
// "A" operand (rows) and the PIXEL tile the "B" operand (cols), so each lane ends up holding 4
// consecutive output channels of one pixel (C/D layout: col = lane&15, row = (lane>>4)*4 + r) and
// stores them as one 8/16-byte vector into the NHWC output.
//
// K is walked in 128-byte chunks of input channels per tap (64 bf16 / 32 fp32).  Both operand tiles
// are staged global -> registers -> LDS (double buffered, one barrier per K-step) as rows of 128 B with
// the 16-byte slot index XOR-swizzled by (row & 7): the ds_read_b128 fragment reads (16 rows x one
// slot column) and the ds_write_b128 staging writes are then bank-conflict free.
//   bf16: one ds_read_b128 = 8 consecutive k of one row = one v_mfma_f32_16x16x32_bf16 operand.
//   fp32: one ds_read_b128 = 4 k values; element j feeds the j-th of four v_mfma_f32_16x16x4_f32
//         (A and B use the same k permutation, so the sum over the chunk is complete).  Exact fp32.
#include "conv_common.h"

namespace {

// WCO x WPX waves (product 4), each computing FCO x FPX fragments of 16x16.
template <typename T, int WCO, int WPX, int FCO, int FPX>
__global__ __launch_bounds__(256) void conv_igemm_kernel(sp_conv_params p) {
    constexpr int CO_T = WCO * FCO * 16, PX_T = WPX * FPX * 16;
    constexpr int E = 16 / (int)sizeof(T);       // elements per 16-byte chunk
    constexpr int KC = 8 * E;                    // elements per 128-byte K chunk
    constexpr int W_CH = CO_T * 8, X_CH = PX_T * 8;
    constexpr int W_PER = (W_CH + 255) / 256, X_PER = (X_CH + 255) / 256;
    constexpr int STAGE = (CO_T + PX_T) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave / WPX, wpx = wave % WPX;
    const int H = p.h, W = p.w_, CIN = p.cin_p;
    const long M = (long)p.n * H * W;
    const long px0 = (long)blockIdx.x * PX_T;
    const int co0 = blockIdx.y * CO_T;
    const int taps = p.ksize * p.ksize;
    const int kchunks = (CIN + KC - 1) / KC;
    const int nk = taps * kchunks;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);

    // ---- staging descriptors, fixed across the K loop: source pointer, 9-bit tap validity mask, LDS byte offset ----
    const int slot = tid & 7;                    // same for every chunk of this thread (256 % 8 == 0)
    const T* x_src[X_PER];
    int x_mask[X_PER], x_dst[X_PER];
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
        const int ch = tid + 256 * i;
        const int row = ch >> 3;
        const long pix = px0 + row;
        const bool ok = (ch < X_CH) && (pix < M);
        const long pc = ok ? pix : 0;
        const int rem = (int)(pc % ((long)H * W));
        const int hh = rem / W, ww = rem - hh * W;
        int m = 0;
        if (ok) {
            if (p.ksize == 3) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int y = hh + t / 3 - 1, x = ww + t % 3 - 1;
                    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) m |= 1 << t;
                }
            } else {
                m = 1;
            }
        }
        x_mask[i] = m;
        x_src[i] = xg + pc * CIN + slot * E;
        x_dst[i] = ch < X_CH ? row * 128 + ((slot ^ (row & 7)) << 4) : -1;
    }
    const T* w_src[W_PER];
    int w_dst[W_PER];
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
        const int ch = tid + 256 * i;
        const int row = ch >> 3;
        const int co = co0 + row;
        w_src[i] = (ch < W_CH && co < p.cout) ? wg + (long)co * taps * CIN + slot * E : nullptr;
        w_dst[i] = ch < W_CH ? row * 128 + ((slot ^ (row & 7)) << 4) : -1;
    }
    const int frow = lane & 15, fslot = lane >> 4;
    int a_off[FCO], b_off[FPX];                  // fragment read offsets for kk = 0; kk = 1 is the same address ^ 64
#pragma unroll
    for (int i = 0; i < FCO; ++i) {
        const int row = (wco * FCO + i) * 16 + frow;
        a_off[i] = row * 128 + ((fslot ^ (row & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < FPX; ++j) {
        const int row = (wpx * FPX + j) * 16 + frow;
        b_off[j] = CO_T * 128 + row * 128 + ((fslot ^ (row & 7)) << 4);
    }

    uint4 xr[X_PER], wr[W_PER];
    auto load_global = [&](int ks) {
        const int tap = ks / kchunks;
        const int c0 = (ks - tap * kchunks) * KC;
        int shift = 0;
        if (p.ksize == 3) shift = (tap / 3 - 1) * W + (tap - (tap / 3) * 3 - 1);
        const bool c_ok = c0 + slot * E < CIN;
        const long xoff = (long)shift * CIN + c0;              // wave-uniform
        const long woff = (long)tap * CIN + c0;
#pragma unroll
        for (int i = 0; i < X_PER; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (((x_mask[i] >> tap) & 1) && c_ok) v = *reinterpret_cast<const uint4*>(x_src[i] + xoff);
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (w_src[i] != nullptr && c_ok) v = *reinterpret_cast<const uint4*>(w_src[i] + woff);
            wr[i] = v;
        }
    };
    auto store_lds = [&](int buf) {
        char* sb = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < W_PER; ++i)
            if (w_dst[i] >= 0) *reinterpret_cast<uint4*>(sb + w_dst[i]) = wr[i];
#pragma unroll
        for (int i = 0; i < X_PER; ++i)
            if (x_dst[i] >= 0) *reinterpret_cast<uint4*>(sb + CO_T * 128 + x_dst[i]) = xr[i];
    };

    f32x4_t acc[FCO][FPX];
#pragma unroll
    for (int i = 0; i < FCO; ++i)
#pragma unroll
        for (int j = 0; j < FPX; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_global(ks + 1);
        const char* sb = smem + buf * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 a[FCO], b[FPX];
#pragma unroll
            for (int i = 0; i < FCO; ++i) a[i] = *reinterpret_cast<const uint4*>(sb + (a_off[i] ^ (kk * 64)));
#pragma unroll
            for (int j = 0; j < FPX; ++j) b[j] = *reinterpret_cast<const uint4*>(sb + (b_off[j] ^ (kk * 64)));
#pragma unroll
            for (int i = 0; i < FCO; ++i)
#pragma unroll
                for (int j = 0; j < FPX; ++j) Mma<T>::run(a[i], b[j], acc[i][j]);
        }
        if (ks + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds 4 consecutive output channels of one pixel per fragment
    T* __restrict__ yg = reinterpret_cast<T*>(p.y);
    const T* r1 = reinterpret_cast<const T*>(p.res1);
    const T* r2 = reinterpret_cast<const T*>(p.res2);
    const T* ms = reinterpret_cast<const T*>(p.mask_src);
    const bool vec_ok = ((p.ldy & 3) == 0) && ((p.cout & 3) == 0);
#pragma unroll
    for (int j = 0; j < FPX; ++j) {
        const long pix = px0 + (wpx * FPX + j) * 16 + (lane & 15);
        if (pix >= M) continue;
#pragma unroll
        for (int i = 0; i < FCO; ++i) {
            const int co = co0 + (wco * FCO + i) * 16 + (lane >> 4) * 4;
            if (co >= p.cout) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            const long off = pix * p.ldy + co;
            if (p.img_scale != nullptr) {
                const float sc = conv_img_scale(p, pix);
                v[0] *= sc; v[1] *= sc; v[2] *= sc; v[3] *= sc;
            }
            if (vec_ok) {
                if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + co);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                float t[4];
                if (ms) {
                    Elem<T>::ld4(ms + off, t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= (t[r] > 0.f ? 1.f : p.mask_neg_slope);
                }
                if (r1) { Elem<T>::ld4(r1 + off, t); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
                if (r2) { Elem<T>::ld4(r2 + off, t); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
                apply_act_vec<4>(v, p.act);
                Elem<T>::st4(yg + off, v);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (co + r >= p.cout) break;
                    float s = v[r];
                    if (p.bias) s += p.bias[co + r];
                    if (ms) s *= (Elem<T>::ld(ms + off + r) > 0.f ? 1.f : p.mask_neg_slope);
                    if (r1) s += Elem<T>::ld(r1 + off + r);
                    if (r2) s += Elem<T>::ld(r2 + off + r);
                    Elem<T>::st(yg + off + r, apply_act(s, p.act));
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// 3x3 convolution with HALO reuse.  A block owns an 8x32 patch of output pixels (256 px) x CO_T output channels.
// Per 128-byte input-channel chunk the (8+2)x(32+2) input halo is brought into LDS ONCE and serves all nine taps
// (the generic kernel above re-loads the shifted pixel tile for every tap); only the CO_T x 128 B weight slice of
// the current tap is streamed (double buffered).  Bytes moved per flop drop ~3x and the per-step staging work of a
// thread from 8+8 to 2+2 (load, LDS write) instructions.  The next chunk's halo is fetched into registers during
// taps 6-8 and written after the chunk's last tap.  Fragment addressing: a B fragment is 16 horizontally
// consecutive pixels of one patch row, shifted by the tap, i.e. 16 consecutive halo rows -> same conflict-free
// XOR-swizzled ds_read_b128 as above.
// ------------------------------------------------------------------------------------------------------------
constexpr int HALO_TH = 8, HALO_TW = 32;
constexpr int HALO_COLS = HALO_TW + 2;           // pixels per halo row that are loaded
constexpr int HALO_W = 40;                       // LDS row pitch in pixels: a multiple of 8, so that moving one halo row
                                                 // down (tap row) leaves the swizzle key (pixel & 7) unchanged
constexpr int HALO_PIX = (HALO_TH + 2) * HALO_W;

template <typename T, int CO_T, int TPS, bool IDX = false>           // IDX: pool2 == 2 with sp_conv_params.pool_idx (its own instantiation)
__global__ __launch_bounds__(CO_T * 4) void conv3x3_halo_kernel(sp_conv_params p) {
    constexpr int NT = CO_T * 4;                 // 8 waves for 128 output channels, 4 waves for 64
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int KC = 8 * E;
    constexpr int H_CH = (HALO_TH + 2) * HALO_COLS * 8;     // 16-byte chunks of the halo tile
    constexpr int H_PER = (H_CH + NT - 1) / NT;
    constexpr int W_PER = (CO_T * 8 * TPS) / NT; // weight chunks per thread per stage (TPS taps are staged together)
    constexpr int WSTAGE = TPS * CO_T * 128;     // bytes of one weight stage
    constexpr int FCO = 4, FPX = 4;              // every wave: 64 co x 64 px (two patch rows x 32 columns)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;                           // [HALO_PIX][128]
    char* wbuf = smem + HALO_PIX * 128;          // 2 x [TPS][CO_T][128]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave >> 2, wpx = wave & 3;
    const int H = p.h, W = p.w_, CIN = p.cin_p;
    const int tiles_x = W / HALO_TW, tiles_y = H / HALO_TH;
    // 1-D grid, XCD-aware: hardware deals consecutive block ids round-robin to the 8 XCDs; every XCD gets a contiguous range
    // of work items with the co-tiles of a patch adjacent, so the tiles that share an input halo share an L2.
    const int cotiles = (p.cout + CO_T - 1) / CO_T;
    int wk = blockIdx.x;
    if ((gridDim.x & 7) == 0) wk = (wk & 7) * (gridDim.x >> 3) + (wk >> 3);
    const int co0 = (wk % cotiles) * CO_T;
    int bid = wk / cotiles;
    const int tx0 = (bid % tiles_x) * HALO_TW;
    bid /= tiles_x;
    const int ty0 = (bid % tiles_y) * HALO_TH;
    const int n = bid / tiles_y;
    const int kchunks = (CIN + KC - 1) / KC;
    // in_up2: x is stored at half resolution; the convolution reads its nearest-neighbour x2 expansion (the x 1/4 of the
    // average-pooling gradient is applied to the accumulators)
    const int up = p.in_up2 ? 1 : 0;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x) + (long)n * (H >> up) * (W >> up) * CIN;
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
    const int slot = tid & 7;

    // ---- staging descriptors, all fixed across the K loop ----
    int h_src[H_PER], h_dst[H_PER];              // source pixel offset (or -1) and LDS byte offset of each halo chunk
#pragma unroll
    for (int i = 0; i < H_PER; ++i) {
        const int ch = tid + NT * i;
        const int hpl = ch >> 3;                                  // linear index over loaded pixels
        const int hy = hpl / HALO_COLS, hx = hpl - hy * HALO_COLS;
        const int yy = ty0 - 1 + hy, xx = tx0 - 1 + hx;
        const int hp = hy * HALO_W + hx;
        h_src[i] = (ch < H_CH && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? (yy >> up) * (W >> up) + (xx >> up) : -1;
        h_dst[i] = ch < H_CH ? hp * 128 + ((slot ^ (hp & 7)) << 4) : -1;
    }
    const T* w_src[W_PER];                       // weight row pointers (first tap of a stage, chunk 0), nullptr for rows >= cout
    int w_dst[W_PER];
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
        const int ch = tid + NT * i;
        const int ts = ch / (CO_T * 8);                       // tap within the stage
        const int row = (ch - ts * (CO_T * 8)) >> 3;
        const int co = co0 + row;
        w_src[i] = co < p.cout ? wg + ((long)co * 9 + ts) * CIN + slot * E : nullptr;
        w_dst[i] = ts * (CO_T * 128) + row * 128 + ((slot ^ ((row & 3) | (((row >> 4) & 1) << 2))) << 4);
    }
    // ---- fragment read offsets (bytes), kk = 0; the kk = 1 half is the same address ^ 64 ----
    const int frow = lane & 15, fslot = lane >> 4;
    int a_off[FCO], b_off[3][FPX];
#pragma unroll
    for (int i = 0; i < FCO; ++i) {
        // fragment rows are permuted (MFMA row rho of fragment i <-> weight row (rho >> 2) * 16 + i * 4 + (rho & 3)): a lane
        // then owns 16 consecutive output channels of its pixel (16-byte epilogue accesses).  8 consecutive lanes read rows
        // {b..b+3, b+16..b+19}: the weight-tile swizzle key takes row bits 0, 1 and 4.
        const int row = wco * 64 + (frow >> 2) * 16 + i * 4 + (frow & 3);
        a_off[i] = row * 128 + ((fslot ^ ((row & 3) | (((row >> 4) & 1) << 2))) << 4);
    }
#pragma unroll
    for (int ds = 0; ds < 3; ++ds)
#pragma unroll
        for (int j = 0; j < FPX; ++j) {
            const int hp = (2 * wpx + (j >> 1)) * HALO_W + (j & 1) * 16 + frow + ds;
            b_off[ds][j] = hp * 128 + ((fslot ^ (hp & 7)) << 4);
        }

    uint4 hr[H_PER], wr[W_PER];
    auto load_halo = [&](int chunk) {
        const int c0 = chunk * KC;
        const bool c_ok = c0 + slot * E < CIN;
#pragma unroll
        for (int i = 0; i < H_PER; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (h_src[i] >= 0 && c_ok) v = *reinterpret_cast<const uint4*>(xg + (long)h_src[i] * CIN + c0 + slot * E);
            hr[i] = v;
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int i = 0; i < H_PER; ++i)
            if (h_dst[i] >= 0) *reinterpret_cast<uint4*>(halo + h_dst[i]) = hr[i];
    };
    auto load_w = [&](int chunk, int stage) {
        const int off = stage * TPS * CIN + chunk * KC;     // wave-uniform element offset
        const bool c_ok = chunk * KC + slot * E < CIN;
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (w_src[i] != nullptr && c_ok) v = *reinterpret_cast<const uint4*>(w_src[i] + off);
            wr[i] = v;
        }
    };
    auto store_w = [&](int buf) {
        char* wb = wbuf + buf * WSTAGE;
#pragma unroll
        for (int i = 0; i < W_PER; ++i) *reinterpret_cast<uint4*>(wb + w_dst[i]) = wr[i];
    };

    f32x4_t acc[FCO][FPX];
#pragma unroll
    for (int i = 0; i < FCO; ++i)
#pragma unroll
        for (int j = 0; j < FPX; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    load_halo(0);
    load_w(0, 0);
    store_halo();
    store_w(0);
    __syncthreads();
    int buf = 0;
    constexpr int NSTAGE = 9 / TPS;
    for (int chunk = 0; chunk < kchunks; ++chunk) {
        const bool next_chunk = chunk + 1 < kchunks;
#pragma unroll
        for (int stage = 0; stage < NSTAGE; ++stage) {         // fully unrolled: tap offsets fold into immediates
            const bool more = stage + 1 < NSTAGE || next_chunk;
            if (more) load_w(stage + 1 < NSTAGE ? chunk : chunk + 1, stage + 1 < NSTAGE ? stage + 1 : 0);
            if (stage == NSTAGE / 2 && next_chunk) load_halo(chunk + 1);
            const char* wb = wbuf + buf * WSTAGE;
#pragma unroll
            for (int ts = 0; ts < TPS; ++ts) {
                const int tap = stage * TPS + ts;
                const int dr = tap / 3, ds = tap % 3;
                const char* wt = wb + ts * (CO_T * 128);
                const char* hb = halo + dr * (HALO_W * 128);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    uint4 a[FCO], b[FPX];
#pragma unroll
                    for (int i = 0; i < FCO; ++i) a[i] = *reinterpret_cast<const uint4*>(wt + (a_off[i] ^ (kk * 64)));
#pragma unroll
                    for (int j = 0; j < FPX; ++j) b[j] = *reinterpret_cast<const uint4*>(hb + (b_off[ds][j] ^ (kk * 64)));
#pragma unroll
                    for (int i = 0; i < FCO; ++i)
#pragma unroll
                        for (int j = 0; j < FPX; ++j) Mma<T>::run(a[i], b[j], acc[i][j]);
                }
            }
            if (more) store_w(buf ^ 1);
            if (stage == NSTAGE - 1 && next_chunk) {
                __syncthreads();        // every wave is done with this chunk's halo
                store_halo();
            }
            __syncthreads();
            buf ^= 1;
        }
    }

    const bool vec_ok = ((p.ldy & 3) == 0) && ((p.cout & 3) == 0);
    const int co_b = co0 + wco * 64 + (lane >> 4) * 16;             // this lane's 16 consecutive channels
    const bool wide = vec_ok && (p.ldy & 7) == 0 && co_b + 16 <= p.cout;
    if (up) {
#pragma unroll
        for (int i = 0; i < FCO; ++i)
#pragma unroll
            for (int j = 0; j < FPX; ++j) acc[i][j] *= 0.25f;
    }
    if (p.pool2) {                                                  // launcher guarantees `wide` for every lane
        const long prow = ((long)n * (H >> 1) + ((ty0 >> 1) + wpx)) * (W >> 1);
        if constexpr (IDX) {                                         // maximum + its window position (the VGG pass with gradient)
            float a0[16], a1[16], b0[16], b1[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a0[i * 4 + r] = acc[i][0][r]; a1[i * 4 + r] = acc[i][2][r];
                    b0[i * 4 + r] = acc[i][1][r]; b1[i * 4 + r] = acc[i][3][r];
                }
            conv_epilogue_pool2_idx<T>(p, a0, a1, b0, b1, lane, prow, tx0 >> 1, co_b);
            return;
        }
        float a[16], b[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[i * 4 + r] = pool2_combine(acc[i][0][r], acc[i][2][r], p.pool2 == 2);
                b[i * 4 + r] = pool2_combine(acc[i][1][r], acc[i][3][r], p.pool2 == 2);
            }
        conv_epilogue_pool2<T>(p, a, b, lane, prow, tx0 >> 1, co_b);
        return;
    }
    static_for<FPX>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int yy = ty0 + 2 * wpx + (j >> 1), xx = tx0 + (j & 1) * 16 + (lane & 15);
        const long pix = ((long)n * H + yy) * W + xx;
        if (wide) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r];
            conv_epilogue16<T>(p, v, pix, co_b);
        } else {
            static_for<4>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int co = co_b + i * 4;
                if (co < p.cout) {
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    conv_epilogue4<T>(p, v, pix, co, vec_ok);
                }
            });
        }
    });
}

template <typename T, int CO_T, int TPS, bool IDX = false>
int launch_halo(const sp_conv_params& p, hipStream_t s) {
    constexpr int LDS = HALO_PIX * 128 + 2 * TPS * CO_T * 128;
    static bool attr_set = false;
    auto kern = conv3x3_halo_kernel<T, CO_T, TPS, IDX>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        attr_set = true;
    }
    dim3 grid((unsigned)(p.n * (p.h / HALO_TH) * (p.w_ / HALO_TW) * ((p.cout + CO_T - 1) / CO_T)));
    sp_note_route(sizeof(T) == 4 ? "conv3x3_halo<f32>" : "conv3x3_halo<16bit>");
    hipLaunchKernelGGL(kern, grid, dim3(CO_T * 4), LDS, s, p);
    SP_LAUNCH_CHECK();
    return SP_OK;
}


// ------------------------------------------------------------------------------------------------------------
// Generic implicit GEMM with LDS-DMA staging (global_load_lds, 16 B per lane): both operand tiles go HBM/L2 -> LDS
// without passing through VGPRs and without ds_write instructions, through a 3-stage ring, so every tile is requested
// TWO K-steps ahead of its use (the register-staged kernel above: one).  This is what the small-spatial layers
// (4x4 .. 16x16: few blocks, long K loops) need - they are latency-bound, not bandwidth- or MFMA-bound.
//   * a wave instruction writes 64 x 16 B = 8 consecutive tile rows, lane-linear; the XOR swizzle is therefore applied
//     on the SOURCE side (lane (row, ps) fetches logical slot ps ^ (row & 7)), the fragment reads stay as above;
//   * out-of-image taps / padded channels / tile rows past the tensor are redirected to a 16-byte zero page;
//   * ordering: own loads by a counted s_waitcnt vmcnt(L) (L = loads of ONE stage stay in flight), everybody's by a
//     raw s_barrier (a __syncthreads() would drain the DMA queue); one barrier per K-step also frees the ring slot
//     that the next request overwrites.
// ------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) uint4 g_zero_page[1];



template <typename T, int WCO, int WPX, int FCO, int FPX>
__global__ __launch_bounds__(256) void conv_igemm_dma_kernel(sp_conv_params p, int ksplit) {
    constexpr int CO_T = WCO * FCO * 16, PX_T = WPX * FPX * 16;
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int KC = 8 * E;
    constexpr int W_PER = (CO_T * 8) / 256, X_PER = (PX_T * 8) / 256;
    static_assert((CO_T * 8) % 256 == 0 && (PX_T * 8) % 256 == 0, "every wave must issue the same number of DMA instructions");
    constexpr int STAGE = (CO_T + PX_T) * 128;
    constexpr int NSTAGE = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / WPX, wpx = wave % WPX;
    const int H = p.h, W = p.w_, CIN = p.cin_p;
    const long M = (long)p.n * H * W;
    const long px0 = (long)blockIdx.x * PX_T;
    const int co0 = blockIdx.y * CO_T;
    const int taps = p.ksize * p.ksize;
    const int kchunks = (CIN + KC - 1) / KC;
    const int nk_all = taps * kchunks;
    // split-K (tiny-spatial layers: too few output tiles to fill the chip): blockIdx.z owns K-steps [k_lo, k_lo + nk)
    const int per = (nk_all + ksplit - 1) / ksplit;
    const int k_lo = blockIdx.z * per;
    const int nk = min(per, nk_all - k_lo);
    if (nk <= 0) return;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    // per-thread chunk descriptors.  Chunk ch = tid + 256*i sits at LDS row ch>>3, physical slot tid&7 and must hold
    // logical slot ls = (tid&7) ^ (row&7) of that row.  Sources are raw buffers (base in SGPRs + 32-bit byte offset per
    // lane); masked lanes (taps outside the image, padded channels, rows past the tensor) use an out-of-range offset, which
    // the buffer unit answers with zeros.  (The first version selected 64-bit pointers against a zero page per load and
    // re-derived tap / chunk with divisions every K-step: 9 SALU + 7 VALU instructions per MFMA on the 16x16 layers.)
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xg), 0, (int)(M * CIN * (long)sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wg), 0, p.cout * taps * CIN * (int)sizeof(T), 0x00020000);
    unsigned x_off[X_PER];
    int x_mask[X_PER], x_ls[X_PER];
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
        const int row = (tid + 256 * i) >> 3;
        const long pix = px0 + row;
        const bool ok = pix < M;
        const long pc = ok ? pix : 0;
        const int rem = (int)(pc % ((long)H * W));
        const int hh = rem / W, ww = rem - hh * W;
        int m = 0;
        if (ok) {
            if (p.ksize == 3) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int y = hh + t / 3 - 1, x = ww + t % 3 - 1;
                    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) m |= 1 << t;
                }
            } else {
                m = 1;
            }
        }
        x_ls[i] = ((tid & 7) ^ (row & 7)) * E;
        x_mask[i] = m;
        x_off[i] = (unsigned)((pc * CIN + x_ls[i]) * (long)sizeof(T));
    }
    unsigned w_off[W_PER];
    int w_ls[W_PER];
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
        const int row = (tid + 256 * i) >> 3;
        const int co = co0 + row;
        w_ls[i] = ((tid & 7) ^ (row & 7)) * E;
        w_off[i] = co < p.cout ? (unsigned)(((long)co * taps * CIN + w_ls[i]) * (long)sizeof(T)) : OOB;
    }
    const int frow = lane & 15, fslot = lane >> 4;
    int a_off[FCO], b_off[FPX];
#pragma unroll
    for (int i = 0; i < FCO; ++i) {
        const int row = (wco * FCO + i) * 16 + frow;
        a_off[i] = row * 128 + ((fslot ^ (row & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < FPX; ++j) {
        const int row = (wpx * FPX + j) * 16 + frow;
        b_off[j] = CO_T * 128 + row * 128 + ((fslot ^ (row & 7)) << 4);
    }

    // K-steps are requested strictly in order: (tap, chunk) advance incrementally
    int i_tap = k_lo / kchunks, i_chunk = k_lo - i_tap * kchunks;
    auto issue = [&](int ks) {
        const int tap = i_tap, c0 = i_chunk * KC;
        if (++i_chunk == kchunks) { i_chunk = 0; ++i_tap; }
        int shift = 0;
        if (p.ksize == 3) shift = (tap / 3 - 1) * W + (tap - (tap / 3) * 3 - 1);
        const unsigned xoff = (unsigned)((shift * CIN + c0) * (int)sizeof(T));     // may be "negative": wraps back in range when valid
        const unsigned woff = (unsigned)((tap * CIN + c0) * (int)sizeof(T));
        char* sb = smem + (ks % NSTAGE) * STAGE + wave * 1024;            // this wave's 1 KB slice of each 4 KB group
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            const unsigned off = (w_off[i] != OOB && c0 + w_ls[i] < CIN) ? w_off[i] + woff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(sb + i * 4096), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < X_PER; ++i) {
            const unsigned off = (((x_mask[i] >> tap) & 1) && c0 + x_ls[i] < CIN) ? x_off[i] + xoff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(sb + CO_T * 128 + i * 4096), 16, (int)off, 0, 0, 0);
        }
    };

    f32x4_t acc[FCO][FPX];
#pragma unroll
    for (int i = 0; i < FCO; ++i)
#pragma unroll
        for (int j = 0; j < FPX; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    issue(0);
    if (nk > 1) issue(1);
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) wait_vmcnt<W_PER + X_PER>();                     // stage ks landed, stage ks+1 may still fly
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (ks + 2 < nk) issue(ks + 2);                                    // ring slot (ks+2)%3 was read in step ks-1
        // Fragment reads are issued as inline asm: hipcc cannot prove that a plain LDS load does not alias the DMA
        // requests still in flight (ring slot index is dynamic) and would put s_waitcnt vmcnt(0) in front of it,
        // draining the prefetch.  The asm reads are covered by the explicit vmcnt/barrier above; their own completion
        // by the lgkmcnt(0) below (+ sched_barrier so no MFMA is hoisted above the wait).
        const unsigned sb = lds_base + (unsigned)((ks % NSTAGE) * STAGE);
        uint4 a0[FCO], b0[FPX], a1[FCO], b1[FPX];
#pragma unroll
        for (int i = 0; i < FCO; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(a0[i]) : "v"(sb + (unsigned)a_off[i]));
#pragma unroll
        for (int j = 0; j < FPX; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(b0[j]) : "v"(sb + (unsigned)b_off[j]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // second half of the K chunk is fetched while the MFMAs of the first half run
#pragma unroll
        for (int i = 0; i < FCO; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(a1[i]) : "v"(sb + (unsigned)(a_off[i] ^ 64)));
#pragma unroll
        for (int j = 0; j < FPX; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(b1[j]) : "v"(sb + (unsigned)(b_off[j] ^ 64)));
#pragma unroll
        for (int i = 0; i < FCO; ++i)
#pragma unroll
            for (int j = 0; j < FPX; ++j) Mma<T>::run(a0[i], b0[j], acc[i][j]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FCO; ++i)
#pragma unroll
            for (int j = 0; j < FPX; ++j) Mma<T>::run(a1[i], b1[j], acc[i][j]);
    }

    const bool vec_ok = ((p.ldy & 3) == 0) && ((p.cout & 3) == 0);
#pragma unroll
    for (int j = 0; j < FPX; ++j) {
        const long pix = px0 + (wpx * FPX + j) * 16 + (lane & 15);
        if (pix >= M) continue;
#pragma unroll
        for (int i = 0; i < FCO; ++i) {
            const int co = co0 + (wco * FCO + i) * 16 + (lane >> 4) * 4;
            if (co >= p.cout) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (ksplit > 1) {                    // partial tile -> this split's slab [M][cout] (plain stores); conv_finalize_kernel
                                                 // sums the slabs and applies the epilogue
                float* wsp = reinterpret_cast<float*>(p.workspace) + ((long)blockIdx.z * M + pix) * p.cout + co;
                if ((p.cout & 3) == 0) {
                    *reinterpret_cast<float4*>(wsp) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.cout) wsp[r] = v[r];
                }
            } else {
                conv_epilogue4<T>(p, v, pix, co, vec_ok);
            }
        }
    }
}

template <typename T>
__global__ void conv_finalize_kernel(sp_conv_params p, int ksplit) {
    const long M = (long)p.n * p.h * p.w_;
    const int groups = (p.cout + 3) / 4;
    const bool vec_ok = ((p.ldy & 3) == 0) && ((p.cout & 3) == 0);
    const float* ws = reinterpret_cast<const float*>(p.workspace);
    const long slab = M * p.cout;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < M * groups; e += (long)gridDim.x * 256) {
        const long pix = e / groups;
        const int co = (int)(e - pix * groups) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const float* src = ws + pix * p.cout + co;
        if ((p.cout & 3) == 0) {
#pragma unroll 4
            for (int z = 0; z < ksplit; ++z) {
                const float4 t = *reinterpret_cast<const float4*>(src + z * slab);
                v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
            }
        } else {
            for (int z = 0; z < ksplit; ++z)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < p.cout) v[r] += src[z * slab + r];
        }
        conv_epilogue4<T>(p, v, pix, co, vec_ok);
    }
}

// ------------------------------------------------------------------------------------------------------------
// 3x3 convolution, "tall" halo kernel: 128 output channels x (16 x 32) output pixels per block, 8 waves, each wave
// 64 co x (4 rows x 32 cols).  Compared with conv3x3_halo_kernel above it
//   * doubles the per-wave tile (128 accumulator registers) and REUSES B fragments vertically: for a fixed tap
//     column ds, the 6 halo rows a wave needs serve all three tap rows dr (row h feeds output rows h-dr), so a stage
//     of 96 MFMAs needs 12 A + 12 B ds_read_b128 (0.25 reads / MFMA; the kernel above: 0.5 - it is LDS-bandwidth bound);
//   * stages both operands with LDS-DMA (global_load_lds, no VGPR staging, no ds_write): K is walked in 64-byte
//     channel chunks (one MFMA k-step), halo double buffered (2 x 45 KB, fetched a whole chunk ahead), the three
//     weight taps of one tap column double buffered (2 x 24 KB, fetched one stage ahead); one barrier per stage;
//   * L2 traffic per flop drops 1.7x (113 KB per 64-byte chunk of a 128 x 512 tile).
// LDS rows are 64 bytes (4 slots of 16 B).  Bank conflicts of ds_read_b128 are decided per 8 consecutive lanes over
// 32 banks (128 B) - measured with SQ_LDS_BANK_CONFLICT on three swizzles (profiles/README.md): 8 consecutive halo
// pixels at one logical slot are conflict free with slot ^ ((pixel >> 1) & 3) (pixel parity picks the 64-byte half, the
// key the slot; invariant under +40 pixels = one halo row); weight rows use key bits (row >> 1) & 1 and (row >> 4) & 1
// because of the permuted fragment rows (below).  The DMA writes lane-linear, so the swizzle is applied on the source
// side: lane (row, ps) fetches logical slot ps ^ key.
// ------------------------------------------------------------------------------------------------------------
constexpr int TL_TH = 16, TL_TW = 32, TL_HR = TL_TH + 2;
constexpr int g_num_cu = 256;                          // MI355X


// WCO = 2: 128 co per block, a wave = 64 co x (4 rows x 32 cols);  WCO = 1: 64 co per block, a wave = 64 co x (2 rows x 32).
// The kernel is PERSISTENT: a block walks work items (co-tile, 16x32 patch) with stride gridDim.x and the DMA pipeline
// runs across item boundaries (the next item's first halo chunk and weight stage are in flight while the current item
// finishes and its epilogue stores drain), so small-K layers (Cin <= 64: one or two chunks per item) no longer pay an
// exposed prologue per tile.
// TH = patch height: 16 (the tall tile) or 8 (WCO = 2 only: the halo kernel's 128 co x 8x32 tile on this kernel's LDS-DMA
// pipeline; a wave then owns 2 rows, 64 accumulator registers, and keeps all 12 A fragments of a stage live: IH = 1).
template <typename T, int WCO, int TH = 16, bool IDX = false>          // IDX: pool2 == 2 with sp_conv_params.pool_idx (its own instantiation)
__global__ __launch_bounds__(512) void conv3x3_tall_kernel(sp_conv_params p, int cotiles, int total, int stagger) {
    constexpr int TL_TH = TH, TL_HR = TH + 2;          // shadow the file-scope constants
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int KC = 4 * E;                    // channels per 64-byte chunk
    constexpr int CO_T = 64 * WCO;
    constexpr int WPX = 8 / WCO;                 // waves along the patch rows
    constexpr int RW = TL_TH / WPX;              // output rows per wave: 4 / 2
    constexpr int NB = RW + 2;                   // halo rows a wave reads
    constexpr int NFR = RW * 2;                  // pixel fragments per wave
    // WCO = 2: a stage = one tap column (3 taps) of a chunk, three barriers per chunk, halo pitch 40 pixels.
    // WCO = 1: a stage = the whole chunk (9 taps, 144 MFMAs per wave between barriers - these layers have only one or two
    //          chunks per item); the 2 x 36 KB of weights fit beside the halo ring only with a halo pitch of 36 pixels, which
    //          flips bit 1 of the swizzle key on odd halo rows (one row = +36 pixels = +18 in pixel >> 1).
    constexpr int TPS = WCO == 1 ? 9 : 3;        // taps per stage
    constexpr int SPC = 9 / TPS;                 // stages per chunk
    constexpr int HP = WCO == 1 ? 36 : 40;       // halo pitch in pixels
    constexpr int HALO_INSTR = (TL_HR * HP * 64 + 1023) / 1024;   // 41 / 45 wave-instructions of 1 KB
    constexpr int HALO_BUF = HALO_INSTR * 1024;
    constexpr int W_BYTES = TPS * CO_T * 64;     // one weight stage
    constexpr int W_INSTR = W_BYTES / 1024;      // 36 / 24 wave-instructions
    constexpr int W_PER = (W_INSTR + 7) / 8;     // per wave: 5 / 3
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / WPX, wpx = wave % WPX;
    const int H = p.h, W = p.w_, CIN = p.cin_p;
    const int tiles_x = W / TL_TW, tiles_y = H / TL_TH;
    const int kchunks = (CIN + KC - 1) / KC;
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
    const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    char* halo_l = smem;                                   // 2 x HALO_BUF
    char* wbuf_l = smem + 2 * HALO_BUF;                    // 2 x W_BYTES
    // XCD-aware order: hardware deals consecutive block ids round-robin to the 8 XCDs; blocks of one XCD get consecutive
    // work items (co-tiles of one patch adjacent), so a patch's halo is fetched into ONE L2.
    const int G = gridDim.x;
    int bid = blockIdx.x;
    if ((G & 7) == 0) bid = (bid & 7) * (G >> 3) + (bid >> 3);
    const int my_items = (total - bid + G - 1) / G;        // items bid, bid + G, ...
    const int nchunks = my_items * kchunks;

    // ---- DMA descriptors.  Lane l of a wave-instruction writes LDS row (l >> 2), physical slot (l & 3).  Sources are
    // addressed as raw buffers (base in SGPRs + a 32-bit byte offset per lane); zero fill (image border, padded
    // channels, rows past cout) = an offset beyond num_records, which the buffer unit answers with zeros.
    constexpr unsigned OOB = 0x80000000u, OOB_C = 0x40000000u;   // position / channel masks; any sum of them and a real offset
                                                                 // (< 2^30, checked by the launcher) stays >= num_records
    const int up = p.in_up2 ? 1 : 0;                         // x at half resolution, read through its nearest-neighbour x2 expansion
    const int HS = H >> up, WS = W >> up;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.n * HS * WS * CIN * (int)sizeof(T), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wg), 0, p.cout * 9 * CIN * (int)sizeof(T), 0x00020000);
    const int ls = ((lane & 3) ^ ((lane >> 3) & 3)) * E;   // halo: logical slot (in elements) this lane must fetch, key (hp >> 1) & 3
    // descriptors of the item whose chunks are being REQUESTED (one chunk / one stage ahead of the compute)
    unsigned h_off[6], w_off[W_PER];
    auto item_coords = [&](int item, int& n, int& ty0, int& tx0, int& co0) {
        const int wk = bid + item * G;
        co0 = (wk % cotiles) * CO_T;
        int t = wk / cotiles;
        tx0 = (t % tiles_x) * TL_TW;
        t /= tiles_x;
        ty0 = (t % tiles_y) * TL_TH;
        n = t / tiles_y;
    };
    int hyx[6];                                            // (halo row << 8) | halo column of this lane's pixel per instruction, -1 = never loaded
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int hp = (wave * 6 + i) * 16 + (lane >> 2);
        const int hy = hp / HP, hx = hp - hy * HP;
        hyx[i] = (hx < TL_TW + 2 && hy < TL_HR) ? (hy << 8) | hx : -1;
        asm volatile("" : "+v"(hyx[i]));                   // keep the packed form live (not the unpacked pair) across the main loop
    }
    auto set_halo_desc = [&](int item) {
        int n, ty0, tx0, co0;
        item_coords(item, n, ty0, tx0, co0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int yy = ty0 - 1 + (hyx[i] >> 8), xx = tx0 - 1 + (hyx[i] & 255);
            const bool ok = hyx[i] >= 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            h_off[i] = ok ? (unsigned)((((n * HS + (yy >> up)) * WS + (xx >> up)) * CIN + ls) * (int)sizeof(T)) : OOB;
        }
    };
    // weight rows (stage row = tap-in-stage * CO_T + co) are swizzled by key = ((co >> 1) & 1) | (((co >> 4) & 1) << 1),
    // row = q * 16 + (lane >> 2): see the fragment addresses below
    auto w_ls = [&](int i) { return ((lane & 3) ^ (((lane >> 3) & 1) | (((wave * W_PER + i) & 1) << 1))) * E; };
    auto set_w_desc = [&](int item) {
        int n, ty0, tx0, co0;
        item_coords(item, n, ty0, tx0, co0);
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            const int q = wave * W_PER + i;
            const int row = q * 16 + (lane >> 2);
            const int ts = row / CO_T, co = co0 + row % CO_T;            // tap inside the stage
            const int tap0 = TPS == 9 ? ts : ts * 3;                     // WCO = 2: tap (dr = ts, ds) - the stage adds ds
            w_off[i] = (q < W_INSTR && co < p.cout) ? (unsigned)(((co * 9 + tap0) * CIN + w_ls(i)) * (int)sizeof(T)) : OOB;
        }
    };
    auto issue_halo = [&](int gc) {                        // gc: block-global chunk index
        const int c0 = (gc % kchunks) * KC;
        const unsigned add = c0 + ls < CIN ? (unsigned)(c0 * (int)sizeof(T)) : OOB_C;
        char* dst = halo_l + (gc & 1) * HALO_BUF + wave * 6 * 1024;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (wave * 6 + i < HALO_INSTR)                 // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                                         (int)(h_off[i] + add), 0, 0, 0);
        }
    };
    auto issue_w = [&](int gc, int ds, int buf) {          // ds: tap column of the stage (WCO = 2), ignored for WCO = 1
        const int c0 = (gc % kchunks) * KC;
        const unsigned base = (unsigned)(((TPS == 9 ? 0 : ds) * CIN + c0) * (int)sizeof(T));
        char* dst = wbuf_l + buf * W_BYTES + wave * W_PER * 1024;
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            if (wave * W_PER + i < W_INSTR) {              // wave-uniform
                const unsigned add = c0 + w_ls(i) < CIN ? base : OOB_C;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                                         (int)(w_off[i] + add), 0, 0, 0);
            }
        }
    };

    // ---- fragment read addresses: one VGPR for A, one per tap column for B; everything else is an immediate ----
    const int frow = lane & 15, fslot = lane >> 4;
    // A-fragment rows are PERMUTED: fragment i, MFMA row rho -> weight row (rho >> 2) * 16 + i * 4 + (rho & 3), so the C/D
    // layout leaves lane (pixel, g) with the 16 CONSECUTIVE output channels g*16 .. g*16 + 15 of its pixel (32 / 64 bytes
    // per lane, 128 / 256 contiguous bytes per pixel and store instruction).  8 consecutive lanes then read rows
    // {b..b+3, b+16..b+19}: the weight-tile swizzle key takes its two bits from row bits 1 and 4.
    const unsigned a_addr = lds_base + 2 * HALO_BUF + (wco * 64 + (frow >> 2) * 16 + (frow & 3)) * 64 +
                            ((fslot ^ (((frow >> 1) & 1) | (((frow >> 2) & 1) << 1))) << 4);
    unsigned b_addr[3];
#pragma unroll
    for (int ds = 0; ds < 3; ++ds)
        b_addr[ds] = lds_base + ((RW * wpx) * HP + frow + ds) * 64 + ((fslot ^ (((frow + ds) >> 1) & 3)) << 4);

    // The accumulators of an item START at the bias of its output channels (lane (pixel, g): channels g*16 + i*4 + r): the
    // bias loads are issued right behind the previous item's stores, where the pipeline waits for memory anyway, instead of
    // in front of every fragment's stores - the epilogue of a bias-only layer then has no load in its dependency chain
    // (per-item overhead was ~5 us of a 20 us item).  Average / max pooling commute with the bias; the x1/4 of the pooled
    // input gradient (in_up2) would scale it, so those launches (which carry no bias anyway) keep the epilogue form.
    const bool bias_in_acc = p.bias != nullptr && !up && p.img_scale == nullptr;   // (a scaled accumulator takes its bias in the epilogue)
    f32x4_t acc[4][NFR];
    auto init_acc = [&](int it) {
        f32x4_t b4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) b4[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (bias_in_acc && it < my_items) {
            int n, ty0, tx0, co0;
            item_coords(it, n, ty0, tx0, co0);
            const int cb = co0 + wco * 64 + (lane >> 4) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = cb + i * 4;
                if (co + 3 < p.cout) {
                    const float4 t = *reinterpret_cast<const float4*>(p.bias + co);
                    b4[i] = f32x4_t{t.x, t.y, t.z, t.w};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) b4[i][r] = co + r < p.cout ? p.bias[co + r] : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NFR; ++j) acc[i][j] = b4[i];
    };
    init_acc(0);

    // one tap column ds of one chunk: 12 A + 2*NB B fragment reads, 24*RW MFMAs.  Halo row h (relative to the wave's
    // first output row) is read once and used by tap rows dr = 0..2 for output row h - dr.  Reads of row h+1 (and the A
    // fragments of tap row h+1) are in flight while row h is multiplied.
    // WCO = 2 walks the column twice, two co-fragments at a time: 24 fewer live fragment registers (128 accumulators leave
    // no room for 12 A fragments), at 36 instead of 24 LDS reads per 96 MFMAs.
    constexpr int IH = (WCO == 2 && RW == 4) ? 2 : 1, IW = 4 / IH;
    auto stage = [&](auto ds_c, unsigned ab, unsigned bb, auto&& mid) {
        constexpr int DS = decltype(ds_c)::value;
        // byte offset of tap (dr, DS) inside the weight stage
        constexpr int TAP_STRIDE = CO_T * 64;
        constexpr int A0 = (TPS == 9 ? DS : 0) * TAP_STRIDE, ADR = (TPS == 9 ? 3 : 1) * TAP_STRIDE;
        const unsigned bo = HP == 36 ? (bb ^ 32u) : bb;    // odd halo rows (pitch 36 only): swizzle key flipped in bit 1
        static_for<IH>([&](auto ihc) {
            constexpr int i0 = decltype(ihc)::value * IW;
            uint4 a[3][IW], bf[NB][2];
            static_for<IW>([&](auto i) { lds_rd128<A0 + (i0 + decltype(i)::value) * 256>(a[0][decltype(i)::value], ab); });
            lds_rd128<0>(bf[0][0], bb); lds_rd128<1024>(bf[0][1], bb);
            static_for<NB>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                if constexpr (h + 1 < NB) {                // request row h + 1 (+ the A fragments of tap row h + 1)
                    if constexpr (h + 1 < 3)
                        static_for<IW>([&](auto i) { lds_rd128<A0 + (h + 1) * ADR + (i0 + decltype(i)::value) * 256>(a[h + 1][decltype(i)::value], ab); });
                    lds_rd128<(h + 1) * (HP * 64)>(bf[h + 1][0], ((h + 1) & 1) ? bo : bb);
                    lds_rd128<(h + 1) * (HP * 64) + 1024>(bf[h + 1][1], ((h + 1) & 1) ? bo : bb);
                    wait_lgkm<(h + 1 < 3) ? IW + 2 : 2>(); // everything older than that request has landed
                } else {
                    wait_lgkm<0>();
                }
                static_for<3>([&](auto drc) {
                    constexpr int dr = decltype(drc)::value;
                    constexpr int rr = h - dr;
                    if constexpr (rr >= 0 && rr < RW) {
#pragma unroll
                        for (int i = 0; i < IW; ++i)
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh) Mma<T>::run(a[dr][i], bf[h][hh], acc[i0 + i][rr * 2 + hh]);
                    }
                });
                if constexpr (h == 1 && decltype(ihc)::value == 0) mid();      // behind the second halo row's MFMAs (see the main loop)
            });
        });
    };

    if (nchunks <= 0) return;
    const bool late = stagger != 0 && wave >= 4;           // wave-uniform (wave comes from readfirstlane)
    set_halo_desc(0);
    set_w_desc(0);
    issue_halo(0);
    issue_w(0, 0, 0);
    int g = 0;                                             // block-global stage counter: weight ring slot = g & 1
    int kc = 0, item = 0;                                  // chunk inside the item / item being computed
    const bool vec_ok = ((p.ldy & 3) == 0) && ((p.cout & 3) == 0);
    for (int gc = 0; gc < nchunks; ++gc) {
        const bool more_chunks = gc + 1 < nchunks;
        const bool item_ends = kc + 1 == kchunks;
        const unsigned hb = (unsigned)((gc & 1) * HALO_BUF);
        static_for<SPC>([&](auto sc) {
            constexpr int st = decltype(sc)::value;       // stage inside the chunk
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                  // stage g landed for everyone; everyone left stage g - 1
            // request the next stage (and, at the first stage of a chunk, the next chunk's halo).  The two waves of a SIMD
            // would otherwise both spend the head of the stage issuing LDS-DMA (3 - 9 wave-instructions of ~60+ issue cycles
            // each) and then both wait for their first fragments: the matrix pipe idles.  The second-dispatched half of the
            // block (waves 4 - 7: one per SIMD) therefore issues ITS requests a third of the way into its MFMA stream, when the
            // first half is multiplying (SP_TUNE_CONV_STAGGER; default: on for the 16-row 128-co tile only, see launch_tall).
            auto request = [&]() {
                if (st + 1 < SPC) {
                    issue_w(gc, st + 1, (g + 1) & 1);
                } else if (more_chunks) {
                    if (item_ends) set_w_desc(item + 1);      // first request for the next item
                    issue_w(gc + 1, 0, (g + 1) & 1);
                }
                if (st == 0 && more_chunks) {
                    if (item_ends) set_halo_desc(item + 1);
                    issue_halo(gc + 1);
                }
            };
            if (!late) request();
            const unsigned ab = a_addr + (unsigned)((g & 1) * W_BYTES);
            auto deferred = [&]() { if (late) request(); };
            auto nothing = []() {};
            if constexpr (SPC == 3) {
                stage(std::integral_constant<int, st>{}, ab, b_addr[st] + hb, deferred);
            } else {
                stage(std::integral_constant<int, 0>{}, ab, b_addr[0] + hb, deferred);
                stage(std::integral_constant<int, 1>{}, ab, b_addr[1] + hb, nothing);
                stage(std::integral_constant<int, 2>{}, ab, b_addr[2] + hb, nothing);
            }
            ++g;
        });
        if (item_ends) {
            int n, ty0, tx0, co0;
            item_coords(item, n, ty0, tx0, co0);
            const long pix0 = ((long)n * H + ty0 + RW * wpx) * W + tx0 + (lane & 15);
            const int co_b = co0 + wco * 64 + (lane >> 4) * 16;          // this lane's 16 consecutive channels
            const bool wide = vec_ok && (p.ldy & 7) == 0 && co_b + 16 <= p.cout;
            if (up) {                                                    // the 1/4 of the average-pooling gradient
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NFR; ++j) acc[i][j] *= 0.25f;
            }
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
                            float a[16], b[16];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    a[i * 4 + r] = pool2_combine(acc[i][j][r], acc[i][j + 2][r], p.pool2 == 2);
                                    b[i * 4 + r] = pool2_combine(acc[i][j + 1][r], acc[i][j + 3][r], p.pool2 == 2);
                                }
                            conv_epilogue_pool2<T>(p, a, b, lane, prow, tx0 >> 1, co_b, !bias_in_acc);
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
            kc = 0;
            ++item;
            init_acc(item);
        } else {
            ++kc;
        }
    }
}

template <typename T, int WCO, int TH = 16, bool IDX = false>
int launch_tall(const sp_conv_params& p, hipStream_t s) {
    constexpr int HP = WCO == 1 ? 36 : 40;
    constexpr int LDS = 2 * ((((TH + 2) * HP * 64 + 1023) / 1024) * 1024) + 2 * (WCO == 1 ? 9 : 3) * 64 * WCO * 64;
    static bool attr_set = false;
    auto kern = conv3x3_tall_kernel<T, WCO, TH, IDX>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        attr_set = true;
    }
    const int cotiles = (p.cout + 64 * WCO - 1) / (64 * WCO);
    const int total = p.n * (p.h / TH) * (p.w_ / TL_TW) * cotiles;
    int grid = total < g_num_cu ? total : g_num_cu;        // persistent: one block per CU
    if (grid >= 8) grid -= grid % 8;
    // measured (scratch/ab_conv.py, profiles/README.md): staggering helps the 16-row 128-co tile (+3-5 %) and costs the 8-row one 4 %
    sp_note_route(sizeof(T) == 4 ? (WCO == 1 ? "conv3x3_tall<f32,1,16>" : TH == 16 ? "conv3x3_tall<f32,2,16>" : "conv3x3_tall<f32,2,8>")
                                 : (WCO == 1 ? "conv3x3_tall<16bit,1,16>" : TH == 16 ? "conv3x3_tall<16bit,2,16>" : "conv3x3_tall<16bit,2,8>"));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, s, p, cotiles, total, sp_tune(SP_TUNE_CONV_STAGGER, (WCO == 2 && TH == 16) ? 1 : 0));
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// Output tile of the LDS-DMA igemm on small-spatial layers with Cout > 64 (SP_TUNE_IGEMM_TILE: 0 = 64 co x 64 px, 1 = 128 x 128,
// 2 = 128 co x 64 px, 3 = 64 co x 128 px).  64 x 64 is the fastest form almost everywhere (round 5, scratch/ab_small_tile.py: these
// launches are latency-bound, larger tiles do not pay by themselves) - except where its tile count lands between one and two rounds
// of the 256 CUs and K is long: 512 -> 512 on 8 x 8 maps at batch 40 has 320 tiles of 72 K-steps each (no split: 1.25 rounds, 36.5
// us); 160 tiles of 128 co x 64 px split K three ways (480 blocks of 24 steps: 28.3 us).
inline int igemm_small_tile(long M, int cout, int cin_p, int ksize, int dtype) {
    const int forced = sp_tune(SP_TUNE_IGEMM_TILE, -1);
    if (forced >= 0) return forced;
    const long tiles64 = ((M + 63) / 64) * ((cout + 63) / 64);
    const int e = dtype == SP_F32 ? 4 : 8;
    const int nk = ksize * ksize * ((cin_p + 8 * e - 1) / (8 * e));
    return (tiles64 > 256 && tiles64 <= 512 && nk >= 48 && cout % 128 == 0) ? 2 : 0;
}

// number of K splits for `tiles` output tiles and nk K-steps: aim at ~1.5 blocks per CU, at least 6 K-steps per split
inline int split_k_plan(int tiles, int nk) {
    const int target = sp_tune(SP_TUNE_SPLITK_TARGET, 384);          // (round 5: 640 -> 384, 5 - 10 % on the 4 x 4 / 8 x 8 layers; scratch/ab_small_tile.py)
    int min_steps = sp_tune(SP_TUNE_SPLITK_MINSTEPS, 6);
    if (min_steps < 1) min_steps = 1;
    if (tiles > 256 || nk < 16) return 1;           // more tiles than CUs: a split only adds the finalize pass (measured)
    int ksplit = (target + tiles - 1) / tiles;
    if (ksplit > nk / min_steps) ksplit = nk / min_steps;
    if (ksplit > 16) ksplit = 16;
    return ksplit < 1 ? 1 : ksplit;
}

template <typename T, int WCO, int WPX, int FCO, int FPX>
int launch_dma(const sp_conv_params& p, hipStream_t s) {
    constexpr int CO_T = WCO * FCO * 16, PX_T = WPX * FPX * 16;
    constexpr int LDS = 3 * (CO_T + PX_T) * 128;
    static bool attr_set = false;
    auto kern = conv_igemm_dma_kernel<T, WCO, WPX, FCO, FPX>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        attr_set = true;
    }
    const long M = (long)p.n * p.h * p.w_;
    const int tiles = (int)((M + PX_T - 1) / PX_T) * ((p.cout + CO_T - 1) / CO_T);
    const int e = p.dtype == SP_F32 ? 4 : 8;
    const int nk = p.ksize * p.ksize * ((p.cin_p + 8 * e - 1) / (8 * e));
    // split-K where the output tiles cannot fill the chip and the caller lent an fp32 workspace of ksplit slabs [M][cout]:
    // every split stores its partial tile with plain stores, conv_finalize_kernel sums the slabs (no fill, no atomics -
    // the first version met in one slab through fp32 atomics and spent most of its time there: 4x4 layers 41 -> 15 us)
    int ksplit = split_k_plan(tiles, nk);
    if (p.workspace == nullptr) ksplit = 1;
    while (ksplit > 1 && p.workspace_bytes < (int64_t)ksplit * M * p.cout * 4) --ksplit;
    if (ksplit > 1) {
        const int per = (nk + ksplit - 1) / ksplit;
        ksplit = (nk + per - 1) / per;                 // every split owns at least one K-step, so every slab is fully written
    }
    dim3 grid((unsigned)((M + PX_T - 1) / PX_T), (unsigned)((p.cout + CO_T - 1) / CO_T), (unsigned)ksplit);
    sp_note_route(ksplit > 1 ? "conv_igemm_dma+finalize (split-K)" : "conv_igemm_dma");
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, p, ksplit);
    SP_LAUNCH_CHECK();
    if (ksplit > 1) {
        long blocks = (M * ((p.cout + 3) / 4) + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(conv_finalize_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, p, ksplit);
        SP_LAUNCH_CHECK();
    }
    return SP_OK;
}

template <typename T, int WCO, int WPX, int FCO, int FPX>
int launch_cfg(const sp_conv_params& p, hipStream_t s) {
    constexpr int CO_T = WCO * FCO * 16, PX_T = WPX * FPX * 16;
    constexpr int LDS = 2 * (CO_T + PX_T) * 128;
    static bool attr_set = false;
    auto kern = conv_igemm_kernel<T, WCO, WPX, FCO, FPX>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        attr_set = true;
    }
    const long M = (long)p.n * p.h * p.w_;
    dim3 grid((unsigned)((M + PX_T - 1) / PX_T), (unsigned)((p.cout + CO_T - 1) / CO_T));
    sp_note_route("conv_igemm (register-staged)");
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, p);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// 1x1 convolution, direct (bf16): y[px][co] = sum_ci W[co][ci] x[px][ci] is a plain GEMM whose B operand (16 pixels x 8
// consecutive channels per lane) can be loaded straight from the NHWC tensor in fragment order - no LDS staging, no
// barriers in the loop.  The 64 x Cin weight tile sits in LDS (loaded once per block), a wave walks over groups of 32
// pixels: per 32 channels one 16-byte global load per pixel fragment, four ds_read_b128 and eight MFMAs.  These layers
// (residual mappings, attention projections, the RGB head; Cin 8 .. 520, 20 us each in the tiled kernel) are latency
// bound: what matters is that nothing serialises.  Fragment rows are permuted as in the 3x3 kernels (16 consecutive
// output channels per lane); LDS rows are padded to a multiple of 128 bytes (+16) and the 64-byte halves swapped for
// rows with bit 4 set, so the 8 rows a lane group reads ({b..b+3, b+16..b+19}) fall on distinct banks.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1x1_direct_kernel(sp_conv_params p, int row_bytes) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];          // [64][row_bytes]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int CIN = p.cin_p;
    const long M = (long)p.n * p.h * p.w_;
    const int co0 = blockIdx.y * 64;
    const bf16* __restrict__ xg = reinterpret_cast<const bf16*>(p.x);
    const bf16* __restrict__ wg = reinterpret_cast<const bf16*>(p.w);
    const int ksteps = (CIN + 31) / 32;
    // ---- weight tile -> LDS, zero-padded to ksteps * 32 channels and 64 rows
    const int cpr = ksteps * 4;                                          // 16-byte chunks per row
    for (int e = tid; e < 64 * cpr; e += 256) {
        const int row = e / cpr, c = e - row * cpr;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (co0 + row < p.cout && c * 8 < CIN) v = *reinterpret_cast<const uint4*>(wg + (long)(co0 + row) * CIN + c * 8);
        *reinterpret_cast<uint4*>(wsm + row * row_bytes + ((c ^ (((row >> 4) & 1) << 2)) << 4)) = v;
    }
    __syncthreads();
    const int i16 = lane & 15, g = lane >> 4;
    const int arow = (i16 >> 2) * 16 + (i16 & 3);                        // + i * 4: permuted fragment rows
    const int aswz = ((arow >> 4) & 1) << 2;
    const bool vec_ok = ((p.ldy & 3) == 0) && ((p.cout & 3) == 0);
    const int co_b = co0 + g * 16;
    const bool wide = vec_ok && (p.ldy & 7) == 0 && co_b + 16 <= p.cout;
    const long ngroups = (M + 31) / 32;
    for (long grp = (long)blockIdx.x * 4 + wave; grp < ngroups; grp += (long)gridDim.x * 4) {
        const long px0 = grp * 32 + i16, px1 = px0 + 16;
        const bf16* x0 = xg + (px0 < M ? px0 : M - 1) * CIN + g * 8;
        const bf16* x1 = xg + (px1 < M ? px1 : M - 1) * CIN + g * 8;
        f32x4_t acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        for (int m = 0; m < ksteps; ++m) {
            uint4 b0 = make_uint4(0, 0, 0, 0), b1 = make_uint4(0, 0, 0, 0);
            if (m * 32 + g * 8 < CIN) {
                b0 = *reinterpret_cast<const uint4*>(x0 + m * 32);
                b1 = *reinterpret_cast<const uint4*>(x1 + m * 32);
            }
            const int slot = (m * 4 + g) ^ aswz;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint4 a = *reinterpret_cast<const uint4*>(wsm + (arow + i * 4) * row_bytes + (slot << 4));
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b0), acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b1), acc[i][1], 0, 0, 0);
            }
        }
        static_for<2>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const long pix = j == 0 ? px0 : px1;
            if (pix < M) {
                if (wide) {
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r];
                    conv_epilogue16<bf16>(p, v, pix, co_b);
                } else {
                    static_for<4>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        const int co = co_b + i * 4;
                        if (co < p.cout) {
                            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                            conv_epilogue4<bf16>(p, v, pix, co, vec_ok);
                        }
                    });
                }
            }
        });
    }
}

int launch_1x1_direct(const sp_conv_params& p, hipStream_t s) {
    const int ksteps = (p.cin_p + 31) / 32;
    const int row_bytes = ((ksteps * 64 + 127) / 128) * 128 + 16;
    const int lds = 64 * row_bytes;
    static int attr = 0;
    if (attr < lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_direct_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", lds, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        attr = lds;
    }
    const long M = (long)p.n * p.h * p.w_;
    const int cotiles = (p.cout + 63) / 64;
    long gx = (M + 127) / 128;                                           // one 32-pixel group per wave and pass at most
    const long cap = 1024 / cotiles > 0 ? 1024 / cotiles : 1;
    if (gx > cap) gx = cap;
    sp_note_route("conv1x1_direct");
    hipLaunchKernelGGL(conv1x1_direct_kernel, dim3((unsigned)gx, (unsigned)cotiles), dim3(256), lds, s, p, row_bytes);
    SP_LAUNCH_CHECK();
    return SP_OK;
}


// ------------------------------------------------------------------------------------------------------------
// 1x1 convolution on a small feature map (bf16; 2x2 .. 16x16 at batch 20: the residual mappings and attention projections of
// the deep stages).  There are only a few hundred 32-pixel groups, so the direct kernel above leaves most SIMDs idle while
// each busy wave walks Cin/32 dependent steps (load -> LDS read -> 8 MFMAs, ~700 cycles each) behind a weight-tile fill it
// uses exactly once.  Here a block owns ONE pixel group x 64 output channels and its four waves split K: every wave requests
// all its operands at once straight from global memory in fragment order (no LDS staging, no barrier before the math), the
// four partial tiles meet in LDS, and wave w finishes fragment row w (bias, residuals, activation, store).  The summation
// order is fixed, so results do not depend on scheduling.   512 -> 512 @4x4: 9.9 -> 5.5 us, 768 -> 512 @2x2: 12.8 -> 6.8 us.
// ------------------------------------------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(256) void conv1x1_splitk_kernel(sp_conv_params p) {
    __shared__ float4 red[4 * 4 * 2 * 64];                              // [source wave][fragment row i][pixel half j][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int CIN = p.cin_p;
    const long M = (long)p.n * p.h * p.w_;
    const int co0 = blockIdx.y * 64;
    const bf16* __restrict__ xg = reinterpret_cast<const bf16*>(p.x);
    const bf16* __restrict__ wg = reinterpret_cast<const bf16*>(p.w);
    const int ksteps = (CIN + 31) / 32;
    const int per = (ksteps + 3) / 4;
    const int m0 = wave * per;
    const int cnt = ksteps - m0 < per ? ksteps - m0 : per;               // K-steps of this wave (<= 0: none)
    const int i16 = lane & 15, g = lane >> 4;
    const int arow = (i16 >> 2) * 16 + (i16 & 3);                        // + i * 4: permuted fragment rows (16 consecutive co per lane)
    const long px0 = (long)blockIdx.x * 32 + i16, px1 = px0 + 16;
    const bf16* x0 = xg + (px0 < M ? px0 : M - 1) * CIN + g * 8;
    const bf16* x1 = xg + (px1 < M ? px1 : M - 1) * CIN + g * 8;
    const bf16* wr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = co0 + arow + i * 4;
        wr[i] = wg + (long)(row < p.cout ? row : p.cout - 1) * CIN + g * 8;   // rows past Cout are computed on a copy and never stored
    }
    f32x4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    // a wave's K-steps in chunks of KMAX (one chunk for Cin <= 32 * 4 * KMAX; Cin > 1024 - channel_factor 0.5's 1536-channel input
    // gradient on its 2 x 2 map - walks two or more, in a fixed order)
    for (int c0 = 0; c0 < cnt; c0 += KMAX) {
        uint4 a[KMAX][4], b[KMAX][2];
        static_for<KMAX>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const int m = m0 + c0 + u;
#pragma unroll
            for (int i = 0; i < 4; ++i) a[u][i] = make_uint4(0, 0, 0, 0);
            b[u][0] = make_uint4(0, 0, 0, 0);
            b[u][1] = make_uint4(0, 0, 0, 0);
            if (c0 + u < cnt && m * 32 + g * 8 < CIN) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[u][i] = *reinterpret_cast<const uint4*>(wr[i] + m * 32);
                b[u][0] = *reinterpret_cast<const uint4*>(x0 + m * 32);
                b[u][1] = *reinterpret_cast<const uint4*>(x1 + m * 32);
            }
        });
        static_for<KMAX>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if (c0 + u < cnt) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[u][i]), __builtin_bit_cast(bf16x8_t, b[u][0]), acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[u][i]), __builtin_bit_cast(bf16x8_t, b[u][1]), acc[i][1], 0, 0, 0);
                }
            }
        });
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            red[((wave * 4 + i) * 2 + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    __syncthreads();
    const bool vec_ok = ((p.ldy & 3) == 0) && ((p.cout & 3) == 0);
    const int co = co0 + g * 16 + wave * 4;
    if (co >= p.cout) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long pix = j == 0 ? px0 : px1;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int src = 0; src < 4; ++src) {
            const float4 t = red[((src * 4 + wave) * 2 + j) * 64 + lane];
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if (pix < M) conv_epilogue4<bf16>(p, v, pix, co, vec_ok);
    }
}

int launch_1x1_splitk(const sp_conv_params& p, hipStream_t s) {
    const int ksteps = (p.cin_p + 31) / 32;
    const long M = (long)p.n * p.h * p.w_;
    dim3 grid((unsigned)((M + 31) / 32), (unsigned)((p.cout + 63) / 64));
    sp_note_route("conv1x1_splitk");
    if (ksteps <= 8) hipLaunchKernelGGL(conv1x1_splitk_kernel<2>, grid, dim3(256), 0, s, p);
    else if (ksteps <= 16) hipLaunchKernelGGL(conv1x1_splitk_kernel<4>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv1x1_splitk_kernel<6>, grid, dim3(256), 0, s, p);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// 3x3 convolution of an 8-channel input (bf16): the RGB images, padded 3 -> 8, entering the discriminator and the VGG-16
// (64 outputs, 256 x 256, batch 20: 1.3 M pixels).  The generic kernels spend one K-step of 32 channels per tap on these 8
// channels - nine MFMAs of which three quarters multiply zero padding - and are MFMA-bound at 73 us where HBM needs 24 us
// (21 MB in, 168 MB out).  Here one pixel's 8 channels are exactly the 8 k-values ONE lane feeds a 16x16x32 MFMA, so the four
// lane groups of a wave take FOUR TAPS of the same pixel: K = 9 taps x 8 channels is covered by three MFMAs.
//   * a block owns 16 rows x 32 pixels; the 18 x 34 halo (16 B per pixel, 9.6 KB) is staged in LDS with zero borders;
//   * lane (px, g) reads for MFMA m the pixel shifted by tap m * 4 + g: one ds_read_b128 at a lane-dependent offset;
//   * the weights [co][tap][8] lie in fragment order already: 12 x 16-byte loads per lane, once, kept in registers;
//   * a wave computes 64 channels x 32 pixels per row (fragment rows permuted: 16 consecutive channels per lane, 32-byte stores).
// ------------------------------------------------------------------------------------------------------------
constexpr int C8_TH = 16, C8_TW = 32, C8_HW = C8_TW + 2, C8_HH = C8_TH + 2;

__global__ __launch_bounds__(256) void conv3x3_cin8_kernel(sp_conv_params p) {
    __shared__ __attribute__((aligned(16))) uint4 halo[C8_HH * C8_HW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.h, W = p.w_;
    const int tiles_x = W / C8_TW, tiles_y = H / C8_TH;
    int t = blockIdx.x;
    const int tx0 = (t % tiles_x) * C8_TW; t /= tiles_x;
    const int ty0 = (t % tiles_y) * C8_TH;
    const int n = t / tiles_y;
    const uint4* __restrict__ xg = reinterpret_cast<const uint4*>(p.x);
    const uint4* __restrict__ wg = reinterpret_cast<const uint4*>(p.w);
    // ---- halo tile (rows ty0 - 1 .. ty0 + 16, columns tx0 - 1 .. tx0 + 32), zero outside the image
    uint4 hv[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int e = tid + q * 256;
        const int r = e / C8_HW, c = e - r * C8_HW;
        const int y = ty0 - 1 + r, x = tx0 - 1 + c;
        hv[q] = make_uint4(0, 0, 0, 0);
        if (e < C8_HH * C8_HW && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) hv[q] = xg[((long)n * H + y) * W + x];
    }
    // ---- weights: fragment i, MFMA m: row co = (i16 >> 2) * 16 + i * 4 + (i16 & 3), k-group g = tap m * 4 + g
    const int i16 = lane & 15, g = lane >> 4;
    uint4 a[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = (i16 >> 2) * 16 + i * 4 + (i16 & 3);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int tap = m * 4 + g;
            a[i][m] = make_uint4(0, 0, 0, 0);
            if (tap < 9 && co < p.cout) a[i][m] = wg[co * 9 + tap];
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int e = tid + q * 256;
        if (e < C8_HH * C8_HW) halo[e] = hv[q];
    }
    __syncthreads();
    // lane's three pixel offsets (in halo pixels, relative to (row, column) of the output pixel); taps 9..11 re-read tap 8
    int toff[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int tap = m * 4 + g < 9 ? m * 4 + g : 8;
        toff[m] = (tap / 3) * C8_HW + tap % 3;
    }
    const int co_b = g * 16;
    const bool co_ok = co_b + 16 <= p.cout;
    float bias[16];                                                            // the lane's 16 channels, once
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = 0.f;
    if (p.bias && co_ok) Wide16<float>::ld(p.bias + co_b, bias);
#pragma unroll 2
    for (int rr = 0; rr < C8_TH / 4; ++rr) {
        const int row = wave * (C8_TH / 4) + rr;                              // output row of the tile
        f32x4_t acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const uint4 b0 = halo[row * C8_HW + i16 + toff[m]];
            const uint4 b1 = halo[row * C8_HW + 16 + i16 + toff[m]];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i][m]), __builtin_bit_cast(bf16x8_t, b0), acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i][m]), __builtin_bit_cast(bf16x8_t, b1), acc[i][1], 0, 0, 0);
            }
        }
        if (!co_ok) continue;
        const long pix0 = ((long)n * H + ty0 + row) * W + tx0 + i16;
        const float sc = conv_img_scale(p, pix0);                            // (a tile row lies inside one image)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r] * sc + bias[i * 4 + r];
            conv_epilogue16<bf16>(p, v, pix0 + j * 16, co_b, false);
        }
    }
}

int launch_cin8(const sp_conv_params& p, hipStream_t s) {
    const long blocks = (long)p.n * (p.h / C8_TH) * (p.w_ / C8_TW);
    sp_note_route("conv3x3_cin8");
    hipLaunchKernelGGL(conv3x3_cin8_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// 3x3 convolution with at most 4 output channels on a big map (bf16): the input gradients of the layers that read the RGB
// images (64 -> 3 @256^2: 168 MB in, 21 MB out, ~24 us of HBM time).  The 64-channel tile of the tall kernel multiplies 61
// idle rows (108 us, MFMA-bound); here a wave's MFMA carries ONE 16-row fragment (rows 0 .. Cout-1 live), so the layer costs
// 18 MFMAs per 16 pixels and is bound by the input stream.
//   * a block owns 8 rows x 32 pixels (three blocks per CU overlap each other's load and multiply phases; a persistent form that
//     prefetched the next tile into registers needed 243 VGPRs, lost the third block and measured 76 instead of 62 us): the
//     10 x 34 halo goes to LDS [pixel][Cin] with a pitch of Cin * 2 + 16 bytes (the 16
//     lanes of a fragment read 16 consecutive pixels: conflict-free), zero outside the image;
//   * the weights [co][tap][ci] are fragment-ordered: 9 * Cin / 32 loads of 16 bytes per lane, once, kept in registers;
//   * lanes of group 0 hold (pixel, co = 0 .. 3): generic scalar epilogue (bias, mask, residuals, activation).
// ------------------------------------------------------------------------------------------------------------
constexpr int TN_TH = 8, TN_TW = 32, TN_HW = TN_TW + 2, TN_HH = TN_TH + 2;

template <int KC>                                                             // Cin / 32
__global__ __launch_bounds__(256) void conv3x3_thinco_kernel(sp_conv_params p) {
    extern __shared__ __attribute__((aligned(16))) char tn_smem[];
    constexpr int CIN = KC * 32, PITCH = CIN * 2 + 16, CPP = KC * 4;           // 16-byte chunks per pixel
    constexpr int NCH = TN_HH * TN_HW * CPP, PER = (NCH + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.h, W = p.w_;
    const int tiles_x = W / TN_TW, tiles_y = H / TN_TH;
    int t = blockIdx.x;
    const int tx0 = (t % tiles_x) * TN_TW; t /= tiles_x;
    const int ty0 = (t % tiles_y) * TN_TH;
    const int n = t / tiles_y;
    const uint4* __restrict__ xg = reinterpret_cast<const uint4*>(p.x);
    const uint4* __restrict__ wg = reinterpret_cast<const uint4*>(p.w);
    uint4 hv[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = tid + q * 256;
        const int px = e / CPP, c = e - px * CPP;
        const int r = px / TN_HW, cc = px - r * TN_HW;
        const int y = ty0 - 1 + r, x = tx0 - 1 + cc;
        hv[q] = make_uint4(0, 0, 0, 0);
        if (e < NCH && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) hv[q] = xg[(((long)n * H + y) * W + x) * CPP + c];
    }
    const int i16 = lane & 15, g = lane >> 4;
    uint4 a[9][KC];                                                            // row i16 = output channel (zero rows past Cout)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            a[tap][kc] = make_uint4(0, 0, 0, 0);
            if (i16 < p.cout) a[tap][kc] = wg[(i16 * 9 + tap) * CPP + kc * 4 + g];
        }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = tid + q * 256;
        const int px = e / CPP, c = e - px * CPP;
        if (e < NCH) *reinterpret_cast<uint4*>(tn_smem + px * PITCH + c * 16) = hv[q];
    }
    __syncthreads();
    const bool vec_ok = ((p.ldy & 3) == 0) && ((p.cout & 3) == 0);
#pragma unroll
    for (int rr = 0; rr < TN_TH / 4; ++rr) {
        const int row = wave * (TN_TH / 4) + rr;
        f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const char* base = tn_smem + ((row + tap / 3) * TN_HW + i16 + tap % 3) * PITCH + g * 16;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const uint4 b0 = *reinterpret_cast<const uint4*>(base + kc * 64);
                const uint4 b1 = *reinterpret_cast<const uint4*>(base + 16 * PITCH + kc * 64);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[tap][kc]), __builtin_bit_cast(bf16x8_t, b0), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[tap][kc]), __builtin_bit_cast(bf16x8_t, b1), acc[1], 0, 0, 0);
            }
        }
        if (g != 0) continue;                                                  // rows 4 .. 15 of the fragment are idle
        const long pix0 = ((long)n * H + ty0 + row) * W + tx0 + i16;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
            conv_epilogue4<bf16>(p, v, pix0 + j * 16, 0, vec_ok);
        }
    }
}

template <int KC>
int launch_thinco(const sp_conv_params& p, hipStream_t s) {
    constexpr int LDS = TN_HH * TN_HW * (KC * 64 + 16);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_thinco_kernel<KC>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { sp_set_error("hipFuncSetAttribute(LDS=%d) failed: %s", LDS, hipGetErrorString(e)); return SP_ERR_LAUNCH; }
        attr_set = true;
    }
    const long blocks = (long)p.n * (p.h / TN_TH) * (p.w_ / TN_TW);
    sp_note_route("conv3x3_thinco");
    hipLaunchKernelGGL(conv3x3_thinco_kernel<KC>, dim3((unsigned)blocks), dim3(256), LDS, s, p);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

template <typename T>
int dispatch(const sp_conv_params& p, hipStream_t s) {
    const long M = (long)p.n * p.h * p.w_;
    if (sizeof(T) == 2 && p.ksize == 3 && p.cout <= 4 && (p.cin_p == 32 || p.cin_p == 64) && !p.pool2 && !p.in_up2 && p.h % TN_TH == 0 &&
        p.w_ % TN_TW == 0 && sp_tune(SP_TUNE_CONV_THINCO, 1) && sp_tune(SP_TUNE_CONV_TALL, 1) <= 1)
        return p.cin_p == 64 ? launch_thinco<2>(p, s) : launch_thinco<1>(p, s);
    if (sizeof(T) == 2 && p.ksize == 3 && p.cin_p == 8 && p.cout % 16 == 0 && p.cout <= 64 && (p.ldy & 7) == 0 && !p.pool2 && !p.in_up2 &&
        p.h % C8_TH == 0 && p.w_ % C8_TW == 0 && sp_tune(SP_TUNE_CONV_CIN8, 1) && sp_tune(SP_TUNE_CONV_TALL, 1) <= 1)   // (forced tall modes: tests)
        return launch_cin8(p, s);
    if (sizeof(T) == 2 && p.ksize == 1 && p.cin_p > 1024 && ((M + 31) / 32) * ((p.cout + 63) / 64) <= sp_tune(SP_TUNE_CONV1X1_SPLITK, 320))
        return launch_1x1_splitk(p, s);                 // (small maps only: the wide networks' deepest 1x1 input gradients; larger maps fall through)
    if (sizeof(T) == 2 && p.ksize == 1 && p.cin_p <= 1024) {
        // small maps: K split over the waves of a block, as long as the blocks (32 pixels x 64 channels each, every one
        // streaming its whole 64 x Cin weight tile from L2) stay few: beyond ~320 the LDS-staged tile of the direct kernel wins
        const int ks = (p.cin_p + 31) / 32;
        const long blocks = ((M + 31) / 32) * ((p.cout + 63) / 64);
        if (ks >= 4 && ks <= 24 && blocks <= sp_tune(SP_TUNE_CONV1X1_SPLITK, 320)) return launch_1x1_splitk(p, s);
        if (sp_tune(SP_TUNE_CONV1X1_DIRECT, 1)) return launch_1x1_direct(p, s);
    }
    // cout <= 16 on a big feature map (the generator's RGB head, 64 -> 3 @256^2): memory-bound; the halo-reuse kernels read
    // the input once instead of once per tap, which outweighs the idle MFMA rows (158 -> ~85 us)
    if (p.pool_idx != nullptr) {
        // ReLU + MaxPool with recorded window positions (pool2 == 2, the API checked h % 8 == 0, w % 32 == 0, cout > 32, cout % 16 == 0):
        // its own instantiations of a fixed set of kernels, so that the hot ones keep their register allocation - 16-bit: the
        // ping-pong kernel's general epilogue (64 co x 16x32 / 128 co x 8x32 tiles); fp32, and what the ping-pong launcher declines:
        // the LDS-DMA tall kernel on the same tiles; Cout <= 64 with h % 16 != 0: the register-staged halo kernel
        const long esz = p.dtype == SP_F32 ? 4 : 2;
        const bool fits30 = (long)p.n * p.h * p.w_ * p.cin_p * esz < (1L << 30) && (long)p.cout * 9 * p.cin_p * esz < (1L << 30);
        if (p.cout <= 64 && p.h % TL_TH != 0) return launch_halo<T, 64, 1, true>(p, s);
        if (!fits30) { sp_set_error("sp_conv2d_igemm: pool_idx needs operands below 1 GiB (n*h*w*cin_p, cout*9*cin_p)"); return SP_ERR_UNSUPPORTED; }
        if (sizeof(T) == 2 && sp_tune(SP_TUNE_CONV_PP, 1)) {
            const int rc = sp_conv_pp_launch(p, p.cout <= 64 ? 16 : 8, s);
            if (rc != 1) return rc;
        }
        return p.cout <= 64 ? launch_tall<T, 1, 16, true>(p, s) : launch_tall<T, 2, 8, true>(p, s);
    }
    const bool thin_big = p.cout <= 16 && p.cin_p >= 32 && M >= (1L << 18);
    if (p.ksize == 3 && (p.cout > 32 || thin_big) && p.h % HALO_TH == 0 && p.w_ % HALO_TW == 0) {
        // persistent tall kernel (half the LDS reads per MFMA, LDS-DMA pipeline across tiles); SP_CONV_TALL=0 disables, 2 forces
        const int tall_mode = sp_tune(SP_TUNE_CONV_TALL, 1);
        const long esz = p.dtype == SP_F32 ? 4 : 2;
        const bool fits30 = (long)p.n * p.h * p.w_ * p.cin_p * esz < (1L << 30) && (long)p.cout * 9 * p.cin_p * esz < (1L << 30);
        const bool tall_ok = tall_mode && fits30 && p.h % TL_TH == 0 && p.w_ % TL_TW == 0;
        if (p.cout <= 64) {
            // bf16, 16 < Cout <= 64 on 16-row patches: the ping-pong schedule with 8 row-pair waves (conv_pp.hip, WCO = 1)
            if (tall_ok && sizeof(T) == 2 && p.cout > 16 && tall_mode <= 1 && sp_tune(SP_TUNE_CONV_PP, 1) && !(sp_tune(SP_TUNE_CONV_PP, 1) & 32)) {
                const int rc = sp_conv_pp_launch(p, 16, s);
                if (rc != 1) return rc;
            }
            if (tall_ok) return launch_tall<T, 1>(p, s);
            return launch_halo<T, 64, 1>(p, s);
        }
        // bf16: the ping-pong schedule (conv_pp.hip) on the same two tiles; tile height by the same round count
        const int pp_mode = sizeof(T) == 2 ? sp_tune(SP_TUNE_CONV_PP, 1) : 0;
        if (pp_mode && fits30 && tall_mode <= 1) {
            const long bt = (long)p.n * (p.h / TL_TH) * (p.w_ / TL_TW) * ((p.cout + 127) / 128);
            const long rt = (bt + 255) / 256, rh = (2 * bt + 255) / 256;
            int th = (tall_ok && 19 * rt < 10 * rh) ? 16 : 8;
            if (pp_mode == 8 || (pp_mode == 16 && tall_ok)) th = pp_mode;
            // 16-row patches: conv_ppw.hip (64 co x 4 rows per wave, 0.25 LDS reads per MFMA under the ping-pong schedule) where its
            // epilogue covers the launch.  One of its items takes 1.72 - 2.04 x the time of an 8-row item (the longer K, the better: its
            // 128-value epilogue amortises; scratch/test_ppw.py, profiles/README.md) - it wins where the round count says so, and
            // everywhere the lockstep tall<2,16> kernel used to (8 - 15 % faster on the same tiles).  SP_TUNE_CONV_PPW: 0 off, 2 forced
            const int ppw_mode = sp_tune(SP_TUNE_CONV_PPW, 1);
            if (tall_ok && ppw_mode && pp_mode == 1 && sp_conv_ppw_covers(p)) {
                const int kch = (p.cin_p + 31) / 32;
                const long ratio = kch >= 12 ? 172 : kch >= 6 ? 185 : kch >= 3 ? 194 : 204;
                // (the 8-row form splits the items of a last, partial round along K - conv_pp.hip: its cost is no longer whole rounds)
                const long wsb = (p.workspace != nullptr && p.split_sync != nullptr) ? p.workspace_bytes : 0;
                const long rh100 = sp_conv_pp_rounds100(2 * bt, p.cin_p, wsb);
                // the 16-row form splits its last round too (conv_ppw.hip), but is CHOSEN on whole rounds: priced with its split it
                // takes a handful of launches the split 8-row form runs within 0 - 6 % of it (256 -> 256 @64^2 x 40: 138 vs 147 us, the
                // others level) - step-neutral (profiles/round6_ab_ppw_tail_split_step.txt), and ties go to the kernel that is
                // measured, profiled and tuned as the dominant one.  SP_TUNE_CONV_PPW = 3 prices it with the split.
                const long rt100 = ppw_mode == 3 ? sp_conv_ppw_rounds100(bt, p.cin_p, wsb) : 100 * rt;
                if (ppw_mode == 2 || ratio * rt100 < 100 * rh100) {
                    const int rc = sp_conv_ppw_launch(p, s);
                    if (rc != 1) return rc;
                }
                th = 8;             // where the four-row form loses on rounds, the 8-row ping-pong form beats the lockstep 16-row kernel too
            }
            if (!(pp_mode == 1 && th == 16)) {             // (the 16-row form is reached only when forced: see conv_pp.hip)
                const int rc = sp_conv_pp_launch(p, th, s);
                if (rc != 1) return rc;
            }
        }
        if (tall_ok) {
            // one block per CU for both kernels, so time ~ rounds over the 256 CUs x time per block; a tall block does twice
            // the work of a halo block in ~1.9x the time (scratch/bench_tall.py, profiles/README.md): it wins where the
            // round quantisation favours it (e.g. 160 instead of 320 blocks).
            const long bt = (long)p.n * (p.h / TL_TH) * (p.w_ / TL_TW) * ((p.cout + 127) / 128);
            const long rt = (bt + 255) / 256, rh = (2 * bt + 255) / 256;
            if (tall_mode == 2 || 19 * rt < 10 * rh) return launch_tall<T, 2>(p, s);
        }
        // remaining Cout > 64 layers: the 128 co x 8x32 tile on the LDS-DMA pipeline of the tall kernel (TH = 8; measured 74 ->
        // 67 us per launch in the step, 902 -> 914 img/s) or, with SP_CONV_SHORT=0 / tall_mode 0, on the register-staged halo kernel
        const int short_env = sp_tune(SP_TUNE_CONV_SHORT, 1);
        if (fits30 && (tall_mode == 3 || (tall_mode == 1 && short_env))) return launch_tall<T, 2, 8>(p, s);
        return launch_halo<T, 128, 3>(p, s);
    }
    // 16-wide maps, Cout > 64, bf16: the ping-pong kernel on 16 x 16-pixel tiles (conv_pp.hip; SP_TUNE_CONV_PP = 0 or 2 keeps the
    // LDS-DMA igemm below)
    if (sizeof(T) == 2 && p.ksize == 3 && p.w_ == 16 && p.cout > 64) {
        const int ppm = sp_tune(SP_TUNE_CONV_PP, 1);
        if (ppm == 1 || ppm == 3) {
            const int rc = sp_conv_pp_launch(p, 1616, s);
            if (rc != 1) return rc;
        }
    }
    // LDS-DMA kernel: measured faster for the small-spatial 3x3 layers (latency-bound), slower for 1x1 (profiles/README.md);
    // SP_IGEMM_DMA=2 forces it everywhere, 0 disables it
    const int dma_mode = sp_tune(SP_TUNE_IGEMM_DMA, 1);
    const long esz_ = p.dtype == SP_F32 ? 4 : 2;
    const bool dma_fits = M * p.cin_p * esz_ < (1L << 30) && (long)p.cout * p.ksize * p.ksize * p.cin_p * esz_ < (1L << 30);
    if (p.cout > 16 && dma_fits && (dma_mode == 2 || (dma_mode == 1 && p.ksize == 3 && M <= 8192))) {
        if (p.cout <= 32) return launch_dma<T, 1, 4, 2, 4>(p, s);        //  32 co x 256 px
        if (p.cout <= 64) return launch_dma<T, 1, 4, 4, 4>(p, s);        //  64 co x 256 px
        if (M <= 8192) {
            switch (igemm_small_tile(M, p.cout, p.cin_p, p.ksize, p.dtype)) {
                case 1: return launch_dma<T, 2, 2, 4, 4>(p, s);          // 128 co x 128 px
                case 2: return launch_dma<T, 2, 2, 4, 2>(p, s);          // 128 co x  64 px
                case 3: return launch_dma<T, 2, 2, 2, 4>(p, s);          //  64 co x 128 px
                default: return launch_dma<T, 2, 2, 2, 2>(p, s);         //  64 co x  64 px
            }
        }
        return launch_dma<T, 2, 2, 4, 4>(p, s);                          // 128 co x 128 px
    }
    if (p.cout <= 16) return launch_cfg<T, 1, 4, 1, 4>(p, s);            //  16 co x 256 px
    if (p.cout <= 32) return launch_cfg<T, 1, 4, 2, 4>(p, s);            //  32 co x 256 px
    if (p.cout <= 64) return launch_cfg<T, 1, 4, 4, 4>(p, s);            //  64 co x 256 px
    if (M <= 8192) return launch_cfg<T, 2, 2, 2, 2>(p, s);               //  64 co x  64 px (tiny spatial: more blocks)
    return launch_cfg<T, 2, 2, 4, 4>(p, s);                              // 128 co x 128 px
}

}  // namespace

extern "C" int sp_conv2d_workspace(int32_t n, int32_t h, int32_t w_, int32_t cin_p, int32_t cout, int32_t ksize, int32_t dtype,
                                   int64_t* bytes_out) {
    SP_CHECK_ARG(bytes_out && n > 0 && h > 0 && w_ > 0 && cin_p > 0 && cout > 0 && (ksize == 1 || ksize == 3), "sp_conv2d_workspace: bad args");
    *bytes_out = 0;
    // mirrors dispatch(): only the LDS-DMA implicit GEMM (3x3, small spatial extent) splits K
    const long M = (long)n * h * w_;
    const int dma_mode = sp_tune(SP_TUNE_IGEMM_DMA, 1);
    const bool halo_path = ksize == 3 && cout > 32 && h % HALO_TH == 0 && w_ % HALO_TW == 0;
    if (halo_path && dtype != SP_F32 && sp_tune(SP_TUNE_CONV_PP, 1)) {
        // the ping-pong kernel's K-split of its last partial round (conv_pp.hip): partial tiles of the tail items
        const long b8 = sp_conv_pp_split_workspace(n, h, w_, cin_p, cout), b16 = sp_conv_ppw_split_workspace(n, h, w_, cin_p, cout);
        *bytes_out = b8 > b16 ? b8 : b16;
        return SP_OK;
    }
    if (ksize == 3 && w_ == 16 && cout > 64 && dtype != SP_F32 && (sp_tune(SP_TUNE_CONV_PP, 1) == 1 || sp_tune(SP_TUNE_CONV_PP, 1) == 3)) {
        const long b16 = sp_conv_pp_split_workspace_w16(n, h, cin_p, cout);       // the same kernel on 16 x 16-pixel tiles
        if (b16 > 0) { *bytes_out = b16; return SP_OK; }
    }
    if (ksize != 3 || halo_path || cout <= 16 || M > 8192 || dma_mode == 0) return SP_OK;
    int co_t = cout <= 32 ? 32 : 64, px_t = cout <= 64 ? 256 : 64;
    if (cout > 64) {                                                     // (the tile dispatch() picks for these layers)
        const int tile = igemm_small_tile(M, cout, cin_p, ksize, dtype);
        if (tile == 1 || tile == 2) co_t = 128;
        if (tile == 1 || tile == 3) px_t = 128;
    }
    const int tiles = (int)((M + px_t - 1) / px_t) * ((cout + co_t - 1) / co_t);
    const int e = dtype == SP_F32 ? 4 : 8;
    const int nk = 9 * ((cin_p + 8 * e - 1) / (8 * e));
    const int ksplit = split_k_plan(tiles, nk);
    if (ksplit > 1) *bytes_out = (int64_t)ksplit * M * cout * 4;
    return SP_OK;
}

extern "C" int sp_conv2d_igemm(const sp_conv_params* pp, sp_stream_t stream) {
    SP_CHECK_ARG(pp != nullptr, "sp_conv2d_igemm: null params");
    const sp_conv_params& p = *pp;
    SP_CHECK_ARG(p.x && p.w && (p.y || (p.dtype == SP_F8 && p.y8) || (p.tail_w && p.tail_y)), "sp_conv2d_igemm: null tensor pointer");
    SP_CHECK_ARG(p.ksize == 1 || p.ksize == 3, "sp_conv2d_igemm: ksize %d unsupported (1 or 3)", p.ksize);
    SP_CHECK_ARG(p.n > 0 && p.h > 0 && p.w_ > 0 && p.cin_p > 0 && p.cout > 0, "sp_conv2d_igemm: bad dims");
    SP_CHECK_ARG(p.dtype == SP_F32 || p.dtype == SP_BF16 || p.dtype == SP_F8, "sp_conv2d_igemm: bad dtype %d", p.dtype);
    if (p.dtype == SP_F8) {
        // BASELINE.json config 5: e4m3 operands on the fp8 MFMA, the ping-pong 3x3 kernel only (conv_pp.hip)
        SP_CHECK_ARG(p.x_scale && p.w_scale && (!p.y8 || p.y8_inv_scale), "sp_conv2d_igemm: SP_F8 needs x_scale, w_scale (and y8_inv_scale with y8)");
        SP_CHECK_ARG(p.cin_p % 16 == 0 && p.cout % 16 == 0 && p.ldy % 16 == 0 && p.ldy >= p.cout, "sp_conv2d_igemm: SP_F8 needs cin_p, cout, ldy multiples of 16");
        SP_CHECK_ARG(p.img_scale == nullptr, "sp_conv2d_igemm: SP_F8 does not take img_scale");
        SP_CHECK_ARG((p.act == SP_ACT_NONE || p.act == SP_ACT_RELU) && (p.pool2 == 0 || p.pool2 == 2) && !p.res1 && !p.res2 && !p.mask_src && !p.in_up2,
                     "sp_conv2d_igemm: SP_F8 supports act NONE / ReLU, pool2 0 / 2, no residuals, no mask_src, no in_up2");
        const int rc = sp_conv_pp_launch(p, 8, reinterpret_cast<hipStream_t>(stream));
        if (rc == 1) { sp_set_error("sp_conv2d_igemm: SP_F8 covers 3x3 layers with cout > 64, h %% 8 == 0, w %% 32 == 0 only"); return SP_ERR_UNSUPPORTED; }
        return rc;
    }
    const int e = p.dtype == SP_F32 ? 4 : 8;
    SP_CHECK_ARG(p.cin_p % e == 0, "sp_conv2d_igemm: cin_p=%d must be a multiple of %d (16 bytes)", p.cin_p, e);
    SP_CHECK_ARG(p.ldy >= p.cout, "sp_conv2d_igemm: ldy < cout");
    if (p.in_up2)
        SP_CHECK_ARG(p.ksize == 3 && p.cout > 32 && p.h % HALO_TH == 0 && p.w_ % HALO_TW == 0,
                     "sp_conv2d_igemm: in_up2 needs a 3x3 layer with cout > 32, h %% 8 == 0, w %% 32 == 0");
    SP_CHECK_ARG(p.pool2 >= 0 && p.pool2 <= 2, "sp_conv2d_igemm: pool2 must be 0, 1 (average) or 2 (maximum)");
    SP_CHECK_ARG(p.pool_idx == nullptr || (p.pool2 == 2 && p.img_scale == nullptr && p.dtype != SP_F8),
                 "sp_conv2d_igemm: pool_idx goes with pool2 == 2 (maximum), without img_scale");
    if (p.pool2 == 2)
        SP_CHECK_ARG(p.res1 == nullptr && p.res2 == nullptr && (p.act == SP_ACT_NONE || p.act == SP_ACT_RELU || p.act == SP_ACT_LRELU),
                     "sp_conv2d_igemm: max pooling in the epilogue needs a monotonic activation and no residuals");
    if (p.pool2)
        SP_CHECK_ARG(p.ksize == 3 && p.cout > 32 && p.cout % 16 == 0 && p.h % HALO_TH == 0 && p.w_ % HALO_TW == 0 && p.ldy % 8 == 0 &&
                         p.mask_src == nullptr,
                     "sp_conv2d_igemm: pool2 needs a 3x3 layer with cout > 32, cout %% 16 == 0, h %% 8 == 0, w %% 32 == 0, ldy %% 8 == 0 and no mask_src");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (p.tail_w != nullptr) {
        // fused 1x1 tail: the 64-channel FAST form of the ping-pong kernel only - no silent fall-back to a kernel that would ignore it
        SP_CHECK_ARG(p.tail_y && p.tail_cout >= 1 && p.tail_cout <= 4 && p.tail_ld >= p.tail_cout && (p.tail_act == SP_ACT_NONE || p.tail_act == SP_ACT_TANH),
                     "sp_conv2d_igemm: bad tail arguments");
        const bool ok = p.dtype == SP_BF16 && p.ksize == 3 && p.cout == 64 && p.h % 16 == 0 && p.w_ % 32 == 0 && (p.ldy & 7) == 0 && !p.pool2 &&
                        !p.in_up2 && p.act != SP_ACT_TANH && p.img_scale == nullptr && sp_tune(SP_TUNE_CONV_PP, 1) && sp_tune(SP_TUNE_CONV_TALL, 1) == 1 &&
                        !(sp_tune(SP_TUNE_CONV_PP_PRIO, 1) & 16);
        if (!ok || sp_conv_pp_launch(p, 16, s) != SP_OK) {
            sp_set_error("sp_conv2d_igemm: the fused 1x1 tail needs 16-bit storage, a 3x3 layer with cout == 64, h %% 16 == 0, w %% 32 == 0, ldy %% 8 == 0, no pooling");
            return SP_ERR_UNSUPPORTED;
        }
        return SP_OK;
    }
    if (p.img_scale != nullptr) {
        SP_CHECK_ARG(p.img_split >= 0 && p.img_split <= p.n, "sp_conv2d_igemm: img_split %d outside [0, n]", p.img_split);
        sp_conv_params q = p;                              // first output pixel of the second group (pooled geometry with pool2)
        q.split_pix_ = (int64_t)p.img_split * (p.pool2 ? (long)(p.h / 2) * (p.w_ / 2) : (long)p.h * p.w_);
        return q.dtype == SP_F32 ? dispatch<float>(q, s) : dispatch<bf16>(q, s);
    }
    return p.dtype == SP_F32 ? dispatch<float>(p, s) : dispatch<bf16>(p, s);
}
