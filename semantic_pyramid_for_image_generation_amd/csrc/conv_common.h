// Pieces shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_pp.hip): MFMA wrappers, compile-time loops,
// the epilogues (bias / activation-derivative mask / residuals / activation / 2x2 pooling) and the inline-asm LDS / wait helpers.
#pragma once
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

struct f8 { uint8_t v; };                        // OCP e4m3 storage (SP_F8): 16 elements per 16-byte fragment read
template <> struct Mma<f8> {
    // a / b: 16 consecutive k of one row (bytes 0-7 feed the first MFMA, 8-15 the second; A and B use the same split, so the
    // sum over the 64-byte chunk is complete)
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
        const long a0 = (long)(((unsigned long long)a.y << 32) | a.x), a1 = (long)(((unsigned long long)a.w << 32) | a.z);
        const long b0 = (long)(((unsigned long long)b.y << 32) | b.x), b1 = (long)(((unsigned long long)b.w << 32) | b.z);
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, c, 0, 0, 0);
    }
};

template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// two-group batches (sp_conv_params.img_scale): the scale of the group output pixel `pix` belongs to (pix counts OUTPUT pixels)
__device__ __forceinline__ float conv_img_scale(const sp_conv_params& p, long pix) {
    return p.img_scale != nullptr ? p.img_scale[pix >= p.split_pix_ ? 1 : 0] : 1.f;
}

template <typename T>
__device__ __forceinline__ void conv_epilogue4(const sp_conv_params& p, float (&v)[4], long pix, int co, bool vec_ok, bool add_bias = true) {
    if (p.img_scale != nullptr && add_bias) {      // (add_bias == false: the caller pre-added the bias and has applied the scale itself)
        const float sc = conv_img_scale(p, pix);
        v[0] *= sc; v[1] *= sc; v[2] *= sc; v[3] *= sc;
    }
    T* __restrict__ yg = reinterpret_cast<T*>(p.y);
    const T* r1 = reinterpret_cast<const T*>(p.res1);
    const T* r2 = reinterpret_cast<const T*>(p.res2);
    const T* ms = reinterpret_cast<const T*>(p.mask_src);
    const long off = pix * p.ldy + co;
    if (vec_ok) {
        if (p.bias && add_bias) {
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
            float sv = v[r];
            if (p.bias && add_bias) sv += p.bias[co + r];
            if (ms) sv *= (Elem<T>::ld(ms + off + r) > 0.f ? 1.f : p.mask_neg_slope);
            if (r1) sv += Elem<T>::ld(r1 + off + r);
            if (r2) sv += Elem<T>::ld(r2 + off + r);
            Elem<T>::st(yg + off + r, apply_act(sv, p.act));
        }
    }
}

// 16 consecutive output channels of one pixel (the tall kernel's permuted fragment rows): 16-byte loads / stores.
template <typename T> struct Wide16;
template <> struct Wide16<bf16> {
    static __device__ __forceinline__ void ld(const bf16* p, float (&o)[16]) {
        const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 8);
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) { o[2 * k] = bf16_bits_to_f32(w[k] & 0xffffu); o[2 * k + 1] = bf16_bits_to_f32(w[k] >> 16); }
    }
    static __device__ __forceinline__ void st(bf16* p, const float (&o)[16]) {
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = f32x2_to_bf16x2(o[2 * k], o[2 * k + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4*>(p + 8) = make_uint4(w[4], w[5], w[6], w[7]);
    }
};
template <> struct Wide16<float> {
    static __device__ __forceinline__ void ld(const float* p, float (&o)[16]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float4 t = *reinterpret_cast<const float4*>(p + 4 * k); o[4 * k] = t.x; o[4 * k + 1] = t.y; o[4 * k + 2] = t.z; o[4 * k + 3] = t.w; }
    }
    static __device__ __forceinline__ void st(float* p, const float (&o)[16]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(p + 4 * k) = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    }
};
// bias / mask / residuals / activation of 16 consecutive channels of one pixel, in place (everything of the epilogue but the store)
template <typename T, bool TANH_OK = true>
__device__ __forceinline__ void conv_epilogue16_values(const sp_conv_params& p, float (&v)[16], long off, int co, bool add_bias = true) {
    const T* r1 = reinterpret_cast<const T*>(p.res1);
    const T* r2 = reinterpret_cast<const T*>(p.res2);
    const T* ms = reinterpret_cast<const T*>(p.mask_src);
    float t[16];
    if (p.bias && add_bias) {
        Wide16<float>::ld(p.bias + co, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += t[r];
    }
    if (ms) {
        Wide16<T>::ld(ms + off, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] *= (t[r] > 0.f ? 1.f : p.mask_neg_slope);
    }
    if (r1) {
        Wide16<T>::ld(r1 + off, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += t[r];
    }
    if (r2) {
        Wide16<T>::ld(r2 + off, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += t[r];
    }
    apply_act_vec<16, TANH_OK>(v, p.act);
}
template <typename T>
__device__ __forceinline__ void conv_epilogue16(const sp_conv_params& p, float (&v)[16], long pix, int co, bool add_bias = true) {
    const long off = pix * p.ldy + co;
    if (p.img_scale != nullptr && add_bias) {
        const float sc = conv_img_scale(p, pix);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] *= sc;
    }
    conv_epilogue16_values<T>(p, v, off, co, add_bias);
    Wide16<T>::st(reinterpret_cast<T*>(p.y) + off, v);
}

// 2x2 average pooling fused into the epilogue (pool2): the two rows of a pair are two fragments of the SAME lane, the two
// columns sit in lanes l and l ^ 1 (DPP quad_perm [1,0,3,2]).  a / b: vertical sums of the column halves 0..15 / 16..31 of a
// 32-pixel row pair.  Even lanes finish the pooled pixel of half a, odd lanes the one of half b, so every lane stores one
// pooled pixel x 16 channels; bias / residuals / activation apply at the pooled resolution (conv_epilogue16 on pooled pixels).
__device__ __forceinline__ float dpp_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
// pool2 == 2: 2x2 MAX pooling (the frozen VGG-16 stages: conv -> ReLU -> MaxPool; bias and a monotonic activation commute
// with the maximum, and so does the bf16 rounding: bit-identical to the separate pooling kernel).
__device__ __forceinline__ float pool2_combine(float x, float y, bool is_max) { return is_max ? fmaxf(x, y) : x + y; }
template <typename T>
__device__ __forceinline__ void conv_epilogue_pool2(const sp_conv_params& p, const float (&a)[16], const float (&b)[16], int lane,
                                                    long ppix_row, int pcol0, int co, bool add_bias = true) {
    // a co-tile past Cout (Cout % 16 == 0 is all the API asks of a pooled layer) has whole 16-channel groups outside the tensor: not
    // stored (they would land on the next pooled pixel; lanes l and l ^ 1 share their group, so the exchange below stays paired)
    if (co + 16 > p.cout) return;
    const bool odd = lane & 1;
    const bool is_max = p.pool2 == 2;
    float v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float recv = dpp_xor1(odd ? a[c] : b[c]);
        const float mine = odd ? b[c] : a[c];
        v[c] = is_max ? fmaxf(mine, recv) : (mine + recv) * 0.25f;
    }
    conv_epilogue16<T>(p, v, ppix_row + pcol0 + (odd ? 8 : 0) + ((lane & 15) >> 1), co, add_bias);
}

// pool2 == 2 with sp_conv_params.pool_idx: the maximum AND its window position.  a0 / a1: the lane's column of half a in the upper /
// lower row of the pair, b0 / b1 likewise for half b (raw accumulators).  The position is the FIRST maximum in scan order
// (row 0: columns 0, 1; row 1: columns 0, 1 - a later element wins only if strictly greater, as in sp_maxpool2_bwd and torch) over
// the values the separate path would have stored and compared: accumulator (+ bias) rounded to the storage type (ReLU is monotonic
// and decides nothing where the maximum is positive; where it is not, the gradient is zero whatever the position).
template <typename T> __device__ __forceinline__ float round_to_storage(float v);
template <> __device__ __forceinline__ float round_to_storage<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to_storage<bf16>(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
// two values at once (one packed conversion instead of two)
template <typename T> __device__ __forceinline__ void round_pair_to_storage(float& a, float& b);
template <> __device__ __forceinline__ void round_pair_to_storage<float>(float&, float&) {}
template <> __device__ __forceinline__ void round_pair_to_storage<bf16>(float& a, float& b) {
    const uint32_t w = f32x2_to_bf16x2(a, b);
    a = h16_lo_to_f32(w);
    b = h16_hi_to_f32(w);
}
template <typename T>
__device__ __forceinline__ void conv_epilogue_pool2_idx(const sp_conv_params& p, const float (&a0)[16], const float (&a1)[16],
                                                        const float (&b0)[16], const float (&b1)[16], int lane, long ppix_row, int pcol0,
                                                        int co, bool add_bias = true) {
    if (co + 16 > p.cout) return;
    const bool odd = lane & 1;
    float bias[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) bias[c] = 0.f;
    if (p.bias && add_bias) Wide16<float>::ld(p.bias + co, bias);
    // even lanes finish the pooled pixel of half a (they hold its even column, lane ^ 1 the odd one), odd lanes the one of half b.
    // Channel by channel (value and row flag travel through one DPP move each): nothing but v[] and idx lives across the loop -
    // these kernels have no register to spare beside their accumulators, and a spill would put scratch traffic on the counted
    // vmcnt waits of their LDS-DMA pipelines
    float v[16];
    unsigned idx = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float ra0 = a0[c] + bias[c], ra1 = a1[c] + bias[c], rb0 = b0[c] + bias[c], rb1 = b1[c] + bias[c];
        round_pair_to_storage<T>(ra0, ra1);
        round_pair_to_storage<T>(rb0, rb1);
        const bool fa = ra1 > ra0, fb = rb1 > rb0;                            // the vertical maximum sits in row 1 (only if strictly greater)
        const float ma = fa ? ra1 : ra0, mb = fb ? rb1 : rb0;
        const float mine = odd ? mb : ma;
        const int mine_row = (odd ? fb : fa) ? 1 : 0;
        const float recv = dpp_xor1(odd ? ma : mb);
        const int recv_row = __builtin_amdgcn_update_dpp(0, (odd ? fa : fb) ? 1 : 0, 0xB1, 0xF, 0xF, true);
        // column 0 of the window is the even lane's, column 1 the odd lane's
        const float m0 = odd ? recv : mine, m1 = odd ? mine : recv;
        const int r0 = odd ? recv_row : mine_row, r1 = odd ? mine_row : recv_row;
        const bool col1 = m1 > m0 || (m1 == m0 && r1 < r0);
        v[c] = col1 ? m1 : m0;
        idx |= (unsigned)(col1 ? (2 * r1 + 1) : (2 * r0)) << (2 * c);
    }
    const long ppix = ppix_row + pcol0 + (odd ? 8 : 0) + ((lane & 15) >> 1);
    p.pool_idx[ppix * (p.cout >> 4) + (co >> 4)] = idx;
    conv_epilogue16<T>(p, v, ppix, co, false);                            // (bias already added; no scale, mask or residuals with pool2 == 2)
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int OFF> __device__ __forceinline__ void lds_rd128(uint4& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

}  // namespace
