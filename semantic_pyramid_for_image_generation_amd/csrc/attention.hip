// SAGAN self-attention core (models.py:262-270): P = softmax(Q K^T) over the pooled keys (no 1/sqrt(d)),
// O = P V, never materialising the (B, HW, HW/4) score tensor in HBM.  q/k/v/o are NHWC, i.e. already
// [batch][position][channel]; the four 1x1 convolutions around this core go through sp_conv2d_igemm.
// One block = TQ queries of one image against all Nk keys: K (fp32, +1-padded rows) and the TQ x Nk score
// tile live in LDS; V streams through L2.  0.2 GFLOP per image - VALU fp32 is ample here, the point of
// the fusion is the 1 MB/image score round trip it removes.
// Backward recomputes P from the saved log-sum-exp; every query block stores its dK / dV contribution in its own fp32
// slab and a reduce pass adds the slabs in block order (no atomics: bit-reproducible).
#include "common.h"

namespace {


template <typename T, int TQ>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                       T* __restrict__ o, float* __restrict__ lse, int N, int NK, int D, int DV) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* ks = sm;                       // [NK][D+1]
    float* qs = ks + NK * (D + 1);        // [TQ][D]
    float* S = qs + TQ * D;               // [TQ][NK]
    float* vs = S + TQ * NK;              // [32][DV]  V chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, q0 = blockIdx.x * TQ;
    const T* kb = k + (long)b * NK * D;
    const T* vb = v + (long)b * NK * DV;
    const T* qb = q + ((long)b * N + q0) * D;
    for (int e = tid; e < NK * D; e += 256) ks[(e / D) * (D + 1) + e % D] = Elem<T>::ld(kb + e);
    for (int e = tid; e < TQ * D; e += 256) qs[e] = (q0 + e / D < N) ? Elem<T>::ld(qb + e) : 0.f;
    __syncthreads();
    for (int j = tid; j < NK; j += 256) {
        const float* kr = ks + j * (D + 1);
        for (int qi = 0; qi < TQ; ++qi) {
            float s = 0.f;
            for (int d = 0; d < D; ++d) s += qs[qi * D + d] * kr[d];
            S[qi * NK + j] = s;
        }
    }
    __syncthreads();
    for (int qi = wave; qi < TQ; qi += 4) {
        float m = -INFINITY;
        for (int j = lane; j < NK; j += 64) m = fmaxf(m, S[qi * NK + j]);
        m = wave_max(m);
        float sum = 0.f;
        for (int j = lane; j < NK; j += 64) { const float e = expf(S[qi * NK + j] - m); S[qi * NK + j] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int j = lane; j < NK; j += 64) S[qi * NK + j] *= inv;
        if (lane == 0 && q0 + qi < N) lse[(long)b * N + q0 + qi] = m + logf(sum);
    }
    // O = P V : thread = (channel c, query phase); V streams through LDS in 32-key chunks (coalesced block-wide reads
    // instead of one dependent 2-byte global load per key and lane)
    const int cpar = DV < 256 ? DV : 256;
    const int qpar = 256 / cpar;
    const int qph = tid / cpar;
    const int c = tid % cpar;
    const int nq = (TQ + qpar - 1) / qpar;
    float acc[TQ];
#pragma unroll
    for (int i = 0; i < TQ; ++i) acc[i] = 0.f;
    for (int j0 = 0; j0 < NK; j0 += 32) {
        __syncthreads();
        for (int e = tid; e < 32 * DV; e += 256) vs[e] = (j0 + e / DV < NK) ? Elem<T>::ld(vb + (long)j0 * DV + e) : 0.f;
        __syncthreads();
        if (qph < qpar) {
#pragma unroll 4
            for (int jj = 0; jj < 32; ++jj) {
                const float vv = vs[jj * DV + c];
#pragma unroll
                for (int i = 0; i < TQ; ++i)
                    if (i < nq) acc[i] += S[(qph + i * qpar) * NK + j0 + jj] * vv;
            }
        }
    }
    if (qph < qpar) {
#pragma unroll
        for (int i = 0; i < TQ; ++i) {
            const int qi = qph + i * qpar;
            if (i < nq && qi < TQ && q0 + qi < N) Elem<T>::st(o + ((long)b * N + q0 + qi) * DV + c, acc[i]);
        }
    }
}

template <typename T, int TQ>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                       const T* __restrict__ dout, const float* __restrict__ lse,
                                                       T* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                       int N, int NK, int D, int DV) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* ks = sm;                       // [NK][D+1]
    float* qs = ks + NK * (D + 1);        // [TQ][D]
    float* dos = qs + TQ * D;             // [TQ][DV]
    float* P = dos + TQ * DV;             // [TQ][NK]  (P, later dS)
    float* red = P + TQ * NK;             // [TQ][4]
    float* vt = red + TQ * 4;             // [32][NK+1] transposed V chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, q0 = blockIdx.x * TQ;
    const T* kb = k + (long)b * NK * D;
    const T* vb = v + (long)b * NK * DV;
    for (int e = tid; e < NK * D; e += 256) ks[(e / D) * (D + 1) + e % D] = Elem<T>::ld(kb + e);
    for (int e = tid; e < TQ * D; e += 256) qs[e] = (q0 + e / D < N) ? Elem<T>::ld(q + ((long)b * N + q0) * D + e) : 0.f;
    for (int e = tid; e < TQ * DV; e += 256) dos[e] = (q0 + e / DV < N) ? Elem<T>::ld(dout + ((long)b * N + q0) * DV + e) : 0.f;
    __syncthreads();
    // requires NK <= 256: thread j owns key column j
    const int j = tid;
    const bool jl = j < NK;
    float p[TQ], dp[TQ];
#pragma unroll
    for (int qi = 0; qi < TQ; ++qi) { p[qi] = 0.f; dp[qi] = 0.f; }
    if (jl) {
        const float* kr = ks + j * (D + 1);
#pragma unroll
        for (int qi = 0; qi < TQ; ++qi) {
            float s = 0.f;
            for (int d = 0; d < D; ++d) s += qs[qi * D + d] * kr[d];
            p[qi] = (q0 + qi < N) ? expf(s - lse[(long)b * N + q0 + qi]) : 0.f;
        }
    }
    // dP[qi][j] = sum_c dO[qi][c] V[j][c]: V goes through LDS transposed, 32 channels at a time
    for (int c0 = 0; c0 < DV; c0 += 32) {
        __syncthreads();
        for (int e = tid; e < NK * 32; e += 256) {
            const int jj = e >> 5, cc = e & 31;
            vt[cc * (NK + 1) + jj] = (c0 + cc < DV) ? Elem<T>::ld(vb + (long)jj * DV + c0 + cc) : 0.f;
        }
        __syncthreads();
        if (jl) {
#pragma unroll 4
            for (int cc = 0; cc < 32; ++cc) {
                if (c0 + cc >= DV) break;
                const float vv = vt[cc * (NK + 1) + j];
#pragma unroll
                for (int qi = 0; qi < TQ; ++qi) dp[qi] += dos[qi * DV + c0 + cc] * vv;
            }
        }
    }
    if (jl) {
#pragma unroll
        for (int qi = 0; qi < TQ; ++qi) P[qi * NK + j] = p[qi];
    }
    __syncthreads();
    // dV[j][c] = sum_qi P[qi][j] dO[qi][c] of this query block: channel-major threads -> coalesced stores into its slab
    {
        const int cpar = DV < 256 ? DV : 256, jpar = 256 / cpar;
        const int c = tid % cpar, jph = tid / cpar;
        if (jph < jpar) {
            float dcol[TQ];
#pragma unroll
            for (int qi = 0; qi < TQ; ++qi) dcol[qi] = dos[qi * DV + c];
            for (int jj = jph; jj < NK; jj += jpar) {
                float a = 0.f;
#pragma unroll
                for (int qi = 0; qi < TQ; ++qi) a += P[qi * NK + jj] * dcol[qi];
                dv[((long)blockIdx.x * gridDim.y + b) * NK * DV + (long)jj * DV + c] = a;      // slab of this query block
            }
        }
    }
    // Drow[qi] = sum_j P dP
#pragma unroll
    for (int qi = 0; qi < TQ; ++qi) {
        const float w = wave_sum(p[qi] * dp[qi]);
        if (lane == 0) red[qi * 4 + wave] = w;
    }
    __syncthreads();
    if (jl) {
#pragma unroll
        for (int qi = 0; qi < TQ; ++qi) {
            const float dr = red[qi * 4] + red[qi * 4 + 1] + red[qi * 4 + 2] + red[qi * 4 + 3];
            const float ds = p[qi] * (dp[qi] - dr);
            P[qi * NK + j] = ds;
            p[qi] = ds;
        }
        // dK[j][d] += sum_qi dS[qi][j] * Q[qi][d]
        for (int d = 0; d < D; ++d) {
            float a = 0.f;
#pragma unroll
            for (int qi = 0; qi < TQ; ++qi) a += p[qi] * qs[qi * D + d];
            dk[((long)blockIdx.x * gridDim.y + b) * NK * D + (long)j * D + d] = a;
        }
    }
    __syncthreads();
    // dQ[qi][d] = sum_j dS[qi][j] * K[j][d]
    for (int e = tid; e < TQ * D; e += 256) {
        const int qi = e / D, d = e - qi * D;
        float a = 0.f;
        for (int jj = 0; jj < NK; ++jj) a += P[qi * NK + jj] * ks[jj * (D + 1) + d];
        if (q0 + qi < N) Elem<T>::st(dq + ((long)b * N + q0 + qi) * D + d, a);
    }
}


// ---- bf16 MFMA forward (D % 32 == 0, DV % 16 == 0, NK % 32 == 0, NK <= 256): one wave = 16 queries --------------
// S^T = K Q^T on MFMA (A = 16 keys x 32 d from L2, B = the wave's 16 queries), so every lane holds, for ITS query
// (lane & 15), 4 keys per 16-key fragment: the softmax is in-register plus two cross-lane steps (xor 16, 32).
// The normalised P goes straight into MFMA B operands for O^T = V^T P^T; V^T fragments come from an LDS copy of V
// ([key][channel], pitch + 32 B) through ds_read_b64_tr_b16, with the same key permutation inside each 32-key step
// as the P registers have (keys {4g..4g+3} U {16+4g..}), exactly as in the weight-gradient kernel.
constexpr int AM_NKMAX = 256;
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                            const bf16* __restrict__ v, bf16* __restrict__ o, float* __restrict__ lse,
                                                            int N, int NK, int D, int DV) {
    extern __shared__ __attribute__((aligned(16))) char vsm[];        // [NK][DV*2 + 32]
    const int PV = DV * 2 + 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const bf16* kb = k + (long)b * NK * D;
    const bf16* vb = v + (long)b * NK * DV;
    const int cpr = DV / 8;                                            // 16-byte chunks per V row
    for (int e = tid; e < NK * cpr; e += 256) {
        const int row = e / cpr, c = e - row * cpr;
        *reinterpret_cast<uint4*>(vsm + row * PV + c * 16) = *reinterpret_cast<const uint4*>(vb + (long)row * DV + c * 8);
    }
    const int i16 = lane & 15, g = lane >> 4;
    const int qrow = min(q0 + i16, N - 1);
    const bf16* qp = q + ((long)b * N + qrow) * D + g * 8;
    // ---- S^T fragments
    f32x4_t sfr[AM_NKMAX / 16];
    const int nf = NK / 16;
#pragma unroll
    for (int f = 0; f < AM_NKMAX / 16; ++f) sfr[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < D; d0 += 32) {
        const uint4 qv = *reinterpret_cast<const uint4*>(qp + d0);
#pragma unroll
        for (int f = 0; f < AM_NKMAX / 16; ++f) {
            if (f < nf) {
                const uint4 kv = *reinterpret_cast<const uint4*>(kb + (long)(f * 16 + i16) * D + d0 + g * 8);
                sfr[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kv), __builtin_bit_cast(bf16x8_t, qv), sfr[f], 0, 0, 0);
            }
        }
    }
    // ---- softmax over keys for query (lane & 15): in-lane over (f, r), then across the 4 lane groups
    float m = -INFINITY;
#pragma unroll
    for (int f = 0; f < AM_NKMAX / 16; ++f)
        if (f < nf) m = fmaxf(fmaxf(fmaxf(m, sfr[f][0]), fmaxf(sfr[f][1], sfr[f][2])), sfr[f][3]);
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int f = 0; f < AM_NKMAX / 16; ++f)
        if (f < nf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __expf(sfr[f][r] - m); sfr[f][r] = e; sum += e; }
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    if (g == 0 && q0 + i16 < N) lse[(long)b * N + q0 + i16] = m + __logf(sum);
    __syncthreads();                                                   // V is in LDS
    // ---- O^T = V^T P^T
    f32x4_t oacc[16];
    const int ncb = DV / 16;
#pragma unroll
    for (int cb = 0; cb < 16; ++cb) oacc[cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < AM_NKMAX / 32; ++t) {
        if (t * 32 < NK) {
            uint4 pb;
            pb.x = f32x2_to_bf16x2(sfr[2 * t][0] * inv, sfr[2 * t][1] * inv);
            pb.y = f32x2_to_bf16x2(sfr[2 * t][2] * inv, sfr[2 * t][3] * inv);
            pb.z = f32x2_to_bf16x2(sfr[2 * t + 1][0] * inv, sfr[2 * t + 1][1] * inv);
            pb.w = f32x2_to_bf16x2(sfr[2 * t + 1][2] * inv, sfr[2 * t + 1][3] * inv);
            const char* vrow = vsm + (t * 32 + g * 4 + (i16 >> 2)) * PV + (i16 & 3) * 8;
#pragma unroll
            for (int cb = 0; cb < 16; ++cb) {
                if (cb < ncb) {
                    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vrow + cb * 32));
                    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vrow + cb * 32 + 16 * PV));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    const uint4 va = make_uint4(l2.x, l2.y, h2.x, h2.y);
                    oacc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, va), __builtin_bit_cast(bf16x8_t, pb), oacc[cb], 0, 0, 0);
                }
            }
        }
    }
    if (q0 + i16 < N) {
        bf16* op = o + ((long)b * N + q0 + i16) * DV + g * 4;
#pragma unroll
        for (int cb = 0; cb < 16; ++cb)
            if (cb < ncb) {
                const float t4[4] = {oacc[cb][0], oacc[cb][1], oacc[cb][2], oacc[cb][3]};
                Elem<bf16>::st4(op + cb * 16, t4);
            }
    }
}

// ---- bf16 MFMA backward (D in {32, 64}, DV % 32 == 0 and <= 256, NK % 32 == 0 and <= 256) -------------------------
// Block = 64 queries (4 waves x 16) of one image against all NK keys.
//  phase 1 (per wave, registers; same [key][query] orientation as the forward): P^T = exp(K Q^T - lse),
//          dP^T = V dO^T, Drow = sum_key P dP, dS^T = P (dP - Drow), dQ^T = K^T dS^T (K^T through ds_read_tr from LDS).
//  phase 2: P and dS go to LDS as bf16 [query][key].
//  phase 3: the contraction over the block's 64 queries runs on MFMA as well: dV^T[c][key] = dO^T P and
//          dK^T[d][key] = Q^T dS, all four operands through ds_read_tr (rows = queries), so A and B see the same
//          query permutation; a wave owns every 4th 16-key fragment.  Partial sums over the N/64 query blocks meet in
//          per-query-block slabs of the fp32 scratch; attn_reduce_kernel sums them and converts.
constexpr int AB_QB = 64;
__device__ __forceinline__ uint4 tr_pair(const char* p, int hi_off) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + hi_off));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}
__device__ __forceinline__ uint4 pack8(const f32x4_t& a, const f32x4_t& b) {
    uint4 r;
    r.x = f32x2_to_bf16x2(a[0], a[1]);
    r.y = f32x2_to_bf16x2(a[2], a[3]);
    r.z = f32x2_to_bf16x2(b[0], b[1]);
    r.w = f32x2_to_bf16x2(b[2], b[3]);
    return r;
}

__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(const bf16* __restrict__ q, const bf16* __restrict__ k,
                                                            const bf16* __restrict__ v, const bf16* __restrict__ dout,
                                                            const float* __restrict__ lse, bf16* __restrict__ dq,
                                                            float* __restrict__ dk, float* __restrict__ dv, int N, int NK, int D, int DV) {
    extern __shared__ __attribute__((aligned(16))) char bsm[];
    const int PK = D * 2 + 32, PDO = DV * 2 + 32, PP = NK * 2 + 32;
    char* Ksm = bsm;                               // [NK][PK]
    char* Qsm = Ksm + NK * PK;                    // [64][PK]
    char* dOsm = Qsm + AB_QB * PK;                // [64][PDO]
    char* Psm = dOsm + AB_QB * PDO;               // [64][PP]
    char* dSsm = Psm + AB_QB * PP;                // [64][PP]
    const int PVL = DV * 2 + 16;                  // V rows while they borrow the P / dS region
    const bool v_in_lds = NK * PVL <= 2 * AB_QB * PP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, qb0 = blockIdx.x * AB_QB;
    const bf16* kb = k + (long)b * NK * D;
    const bf16* vb = v + (long)b * NK * DV;
    // every query block writes its own partial dK / dV slab (plain 16-byte stores; summed by attn_reduce_kernel): fp32
    // atomics from 16 query blocks onto the same rows cost 170 of this kernel's 223 us
    float* dk_slab = dk + (long)blockIdx.x * gridDim.y * NK * D;
    float* dv_slab = dv + (long)blockIdx.x * gridDim.y * NK * DV;
    // ---- stage K, Q, dO (16-byte chunks; query rows past N are zero)
    {
        const int cpr = D / 8;
        for (int e = tid; e < NK * cpr; e += 256) {
            const int row = e / cpr, c = e - row * cpr;
            *reinterpret_cast<uint4*>(Ksm + row * PK + c * 16) = *reinterpret_cast<const uint4*>(kb + (long)row * D + c * 8);
        }
        for (int e = tid; e < AB_QB * cpr; e += 256) {
            const int row = e / cpr, c = e - row * cpr;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (qb0 + row < N) val = *reinterpret_cast<const uint4*>(q + ((long)b * N + qb0 + row) * D + c * 8);
            *reinterpret_cast<uint4*>(Qsm + row * PK + c * 16) = val;
        }
        const int cpo = DV / 8;
        for (int e = tid; e < AB_QB * cpo; e += 256) {
            const int row = e / cpo, c = e - row * cpo;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (qb0 + row < N) val = *reinterpret_cast<const uint4*>(dout + ((long)b * N + qb0 + row) * DV + c * 8);
            *reinterpret_cast<uint4*>(dOsm + row * PDO + c * 16) = val;
        }
        // V (phase 1 only) borrows the P / dS region when it fits: 16 scattered 16-byte global loads per fragment become
        // conflict-free ds_read_b128 (row pitch DV*2 + 16 bytes: 8 consecutive rows cover all 32 banks)
        if (v_in_lds) {
            for (int e = tid; e < NK * cpo; e += 256) {
                const int row = e / cpo, c = e - row * cpo;
                *reinterpret_cast<uint4*>(Psm + row * PVL + c * 16) = *reinterpret_cast<const uint4*>(vb + (long)row * DV + c * 8);
            }
        }
    }
    __syncthreads();
    const int i16 = lane & 15, g = lane >> 4;
    const int ql = wave * 16 + i16;                                  // this lane's query inside the block
    const bool q_ok = qb0 + ql < N;
    const int nf = NK / 16;
    // ---- phase 1: P^T and dP^T fragments
    f32x4_t pfr[AM_NKMAX / 16], dpf[AM_NKMAX / 16];
#pragma unroll
    for (int f = 0; f < AM_NKMAX / 16; ++f) { pfr[f] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dpf[f] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    for (int d0 = 0; d0 < D; d0 += 32) {
        const uint4 qv = *reinterpret_cast<const uint4*>(Qsm + ql * PK + (d0 + g * 8) * 2);
#pragma unroll
        for (int f = 0; f < AM_NKMAX / 16; ++f)
            if (f < nf) {
                const uint4 kv = *reinterpret_cast<const uint4*>(Ksm + (f * 16 + i16) * PK + (d0 + g * 8) * 2);
                pfr[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kv), __builtin_bit_cast(bf16x8_t, qv), pfr[f], 0, 0, 0);
            }
    }
    for (int c0 = 0; c0 < DV; c0 += 32) {
        const uint4 dov = *reinterpret_cast<const uint4*>(dOsm + ql * PDO + (c0 + g * 8) * 2);
#pragma unroll
        for (int f = 0; f < AM_NKMAX / 16; ++f)
            if (f < nf) {
                const uint4 vv = v_in_lds ? *reinterpret_cast<const uint4*>(Psm + (f * 16 + i16) * PVL + (c0 + g * 8) * 2)
                                          : *reinterpret_cast<const uint4*>(vb + (long)(f * 16 + i16) * DV + c0 + g * 8);
                dpf[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), __builtin_bit_cast(bf16x8_t, dov), dpf[f], 0, 0, 0);
            }
    }
    const float l = q_ok ? lse[(long)b * N + qb0 + ql] : 0.f;
    float drow = 0.f;
#pragma unroll
    for (int f = 0; f < AM_NKMAX / 16; ++f)
        if (f < nf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = q_ok ? __expf(pfr[f][r] - l) : 0.f;
                pfr[f][r] = pv;
                drow += pv * dpf[f][r];
            }
        }
    drow += __shfl_xor(drow, 16, 64);
    drow += __shfl_xor(drow, 32, 64);
#pragma unroll
    for (int f = 0; f < AM_NKMAX / 16; ++f)
        if (f < nf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dpf[f][r] = pfr[f][r] * (dpf[f][r] - drow);      // dS^T
        }
    // ---- dQ^T = K^T dS^T
    {
        f32x4_t dqa[4];
        const int ndb = D / 16;
#pragma unroll
        for (int db = 0; db < 4; ++db) dqa[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < AM_NKMAX / 32; ++t)
            if (t * 32 < NK) {
                const uint4 sb = pack8(dpf[2 * t], dpf[2 * t + 1]);
                const char* krow = Ksm + (t * 32 + g * 4 + (i16 >> 2)) * PK + (i16 & 3) * 8;
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    if (db < ndb) {
                        const uint4 ka = tr_pair(krow + db * 32, 16 * PK);
                        dqa[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ka), __builtin_bit_cast(bf16x8_t, sb), dqa[db], 0, 0, 0);
                    }
            }
        if (q_ok) {
            bf16* qp = dq + ((long)b * N + qb0 + ql) * D + g * 4;
#pragma unroll
            for (int db = 0; db < 4; ++db)
                if (db < ndb) {
                    const float t4[4] = {dqa[db][0], dqa[db][1], dqa[db][2], dqa[db][3]};
                    Elem<bf16>::st4(qp + db * 16, t4);
                }
        }
    }
    // ---- phase 2: P, dS -> LDS [query][key] (lane: query ql, keys f*16 + g*4 .. +3)
    if (v_in_lds) __syncthreads();                 // every wave is done with V before its rows are overwritten
#pragma unroll
    for (int f = 0; f < AM_NKMAX / 16; ++f)
        if (f < nf) {
            uint2 pw, sw;
            pw.x = f32x2_to_bf16x2(pfr[f][0], pfr[f][1]);
            pw.y = f32x2_to_bf16x2(pfr[f][2], pfr[f][3]);
            sw.x = f32x2_to_bf16x2(dpf[f][0], dpf[f][1]);
            sw.y = f32x2_to_bf16x2(dpf[f][2], dpf[f][3]);
            *reinterpret_cast<uint2*>(Psm + ql * PP + (f * 16 + g * 4) * 2) = pw;
            *reinterpret_cast<uint2*>(dSsm + ql * PP + (f * 16 + g * 4) * 2) = sw;
        }
    __syncthreads();
    // ---- phase 3: dV^T = dO^T P, dK^T = Q^T dS over the block's 64 queries; wave w owns key fragments w, w+4, ...
    const int rowsel = g * 4 + (i16 >> 2), colsel = (i16 & 3) * 8;
    for (int cb0 = 0; cb0 < DV / 16; cb0 += 8) {           // 128 channels of dV per pass (DV = 256: two passes)
        f32x4_t acc[4][8];
        const int ncb = min(8, DV / 16 - cb0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) acc[a][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t2 = 0; t2 < AB_QB / 32; ++t2) {
            uint4 pb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int kf = wave + 4 * a;
                pb[a] = tr_pair(Psm + (t2 * 32 + rowsel) * PP + colsel + (kf < nf ? kf : 0) * 32, 16 * PP);
            }
#pragma unroll
            for (int cb = 0; cb < 8; ++cb)
                if (cb < ncb) {
                    const uint4 da = tr_pair(dOsm + (t2 * 32 + rowsel) * PDO + colsel + (cb0 + cb) * 32, 16 * PDO);
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[a][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, da), __builtin_bit_cast(bf16x8_t, pb[a]), acc[a][cb], 0, 0, 0);
                }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int kf = wave + 4 * a;
            if (kf < nf) {
                float* dst = dv_slab + ((long)b * NK + kf * 16 + i16) * DV + cb0 * 16 + g * 4;
#pragma unroll
                for (int cb = 0; cb < 8; ++cb)
                    if (cb < ncb) *reinterpret_cast<float4*>(dst + cb * 16) = make_float4(acc[a][cb][0], acc[a][cb][1], acc[a][cb][2], acc[a][cb][3]);
            }
        }
    }
    {
        f32x4_t acc[4][4];
        const int ndb = D / 16;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int db = 0; db < 4; ++db) acc[a][db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t2 = 0; t2 < AB_QB / 32; ++t2) {
            uint4 sb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int kf = wave + 4 * a;
                sb[a] = tr_pair(dSsm + (t2 * 32 + rowsel) * PP + colsel + (kf < nf ? kf : 0) * 32, 16 * PP);
            }
#pragma unroll
            for (int db = 0; db < 4; ++db)
                if (db < ndb) {
                    const uint4 qa = tr_pair(Qsm + (t2 * 32 + rowsel) * PK + colsel + db * 32, 16 * PK);
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[a][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, qa), __builtin_bit_cast(bf16x8_t, sb[a]), acc[a][db], 0, 0, 0);
                }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int kf = wave + 4 * a;
            if (kf < nf) {
                float* dst = dk_slab + ((long)b * NK + kf * 16 + i16) * D + g * 4;
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    if (db < ndb) *reinterpret_cast<float4*>(dst + db * 16) = make_float4(acc[a][db][0], acc[a][db][1], acc[a][db][2], acc[a][db][3]);
            }
        }
    }
}

// dst[i] = sum over slabs of src[slab][i]
template <typename T>
__global__ void attn_reduce_kernel(const float* __restrict__ src, int nslabs, long n, T* __restrict__ dst) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (long)gridDim.x * 256) {
        float4 a = reinterpret_cast<const float4*>(src)[i];
        for (int s = 1; s < nslabs; ++s) {
            const float4 t = reinterpret_cast<const float4*>(src + (long)s * n)[i];
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        const float o[4] = {a.x, a.y, a.z, a.w};
        Elem<T>::st4(dst + i * 4, o);
    }
}

template <typename T>
__global__ void cast_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) Elem<T>::st(dst + i, src[i]);
}

template <typename T, int TQ>
int launch_attn_tq(bool fwd, const void* q, const void* k, const void* v, void* o_or_dq, const void* dout, float* lse, float* dk,
                   float* dv, int B, int N, int NK, int D, int DV, hipStream_t s) {
    const int lds_f = (NK * (D + 1) + TQ * D + TQ * NK + 32 * DV) * 4;
    const int lds_b = (NK * (D + 1) + TQ * D + TQ * DV + TQ * NK + TQ * 4 + 32 * (NK + 1)) * 4;
    dim3 grid(sp_div_up(N, TQ), B);
    if (fwd) {
        { static int done = 0; if (done < lds_f) { hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<T, TQ>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_f); done = lds_f; } }
        hipLaunchKernelGGL((attn_fwd_kernel<T, TQ>), grid, dim3(256), lds_f, s, (const T*)q, (const T*)k, (const T*)v, (T*)o_or_dq, lse, N, NK, D, DV);
    } else {
        { static int done = 0; if (done < lds_b) { hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<T, TQ>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b); done = lds_b; } }
        hipLaunchKernelGGL((attn_bwd_kernel<T, TQ>), grid, dim3(256), lds_b, s, (const T*)q, (const T*)k, (const T*)v, (const T*)dout, lse,
                           (T*)o_or_dq, dk, dv, N, NK, D, DV);
    }
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// 32 queries per block unless the backward's LDS tiles (keys + scores + dO + transposed V chunk) would not fit 160 KB
inline int attn_valu_tq(int NK, int D, int DV) {
    const long lds_b32 = (long)(NK * (D + 1) + 32 * D + 32 * DV + 32 * NK + 32 * 4 + 32 * (NK + 1)) * 4;
    return lds_b32 <= 160 * 1024 ? 32 : 16;
}
inline bool attn_bwd_mfma_ok(int dtype, int nk, int d, int dv) {
    return dtype == SP_BF16 && (d == 32 || d == 64) && dv % 32 == 0 && dv <= 256 && nk % 32 == 0;
}
template <typename T>
int launch_attn(bool fwd, const void* q, const void* k, const void* v, void* o_or_dq, const void* dout, float* lse, float* dk,
                float* dv, int B, int N, int NK, int D, int DV, hipStream_t s) {
    if (attn_valu_tq(NK, D, DV) == 32) return launch_attn_tq<T, 32>(fwd, q, k, v, o_or_dq, dout, lse, dk, dv, B, N, NK, D, DV, s);
    return launch_attn_tq<T, 16>(fwd, q, k, v, o_or_dq, dout, lse, dk, dv, B, N, NK, D, DV, s);
}

}  // namespace

extern "C" int sp_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int32_t batch, int32_t n,
                                int32_t nk, int32_t d, int32_t dv, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(q && k && v && o && lse, "sp_attention_fwd: null pointer");
    SP_CHECK_ARG(nk > 0 && nk <= 256 && d > 0 && d <= 64 && dv > 0 && dv <= 256, "sp_attention_fwd: unsupported extents nk=%d d=%d dv=%d", nk, d, dv);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SP_BF16 && d % 32 == 0 && dv % 16 == 0 && dv <= 256 && nk % 32 == 0) {
        const int lds = nk * (dv * 2 + 32);
        { static int done3 = 0; if (done3 < lds) { hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds); done3 = lds; } }
        hipLaunchKernelGGL(attn_fwd_mfma_kernel, dim3(sp_div_up(n, 64), batch), dim3(256), lds, s, (const bf16*)q, (const bf16*)k, (const bf16*)v,
                           (bf16*)o, lse, n, nk, d, dv);
        SP_LAUNCH_CHECK();
        return SP_OK;
    }
    return dtype == SP_F32 ? launch_attn<float>(true, q, k, v, o, nullptr, lse, nullptr, nullptr, batch, n, nk, d, dv, s)
                           : launch_attn<bf16>(true, q, k, v, o, nullptr, lse, nullptr, nullptr, batch, n, nk, d, dv, s);
}

extern "C" int sp_attention_bwd(const void* q, const void* k, const void* v, const void* dout, const float* lse, void* dq,
                                float* dk_f32, float* dv_f32, void* dk, void* dv_out, int32_t batch, int32_t n, int32_t nk,
                                int32_t d, int32_t dv, int32_t dtype, sp_stream_t stream) {
    SP_CHECK_ARG(q && k && v && dout && lse && dq && dk_f32 && dv_f32 && dk && dv_out, "sp_attention_bwd: null pointer");
    SP_CHECK_ARG(nk > 0 && nk <= 256 && d > 0 && d <= 64 && dv > 0 && dv <= 256, "sp_attention_bwd: unsupported extents nk=%d d=%d dv=%d", nk, d, dv);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long nkd = (long)batch * nk * d, nkv = (long)batch * nk * dv;
    if (attn_bwd_mfma_ok(dtype, nk, d, dv)) {
        const int nqb = sp_div_up(n, AB_QB);           // one partial slab per query block (scratch: nqb x the gradient size)
        const int lds = nk * (d * 2 + 32) + AB_QB * (d * 2 + 32) + AB_QB * (dv * 2 + 32) + 2 * AB_QB * (nk * 2 + 32);
        { static int done4 = 0; if (done4 < lds) { hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds); done4 = lds; } }
        hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3(nqb, batch), dim3(256), lds, s, (const bf16*)q, (const bf16*)k,
                           (const bf16*)v, (const bf16*)dout, lse, (bf16*)dq, dk_f32, dv_f32, n, nk, d, dv);
        hipLaunchKernelGGL(attn_reduce_kernel<bf16>, dim3(sp_div_up(nkd / 4, 256)), dim3(256), 0, s, dk_f32, nqb, nkd, (bf16*)dk);
        hipLaunchKernelGGL(attn_reduce_kernel<bf16>, dim3(sp_div_up(nkv / 4, 256)), dim3(256), 0, s, dv_f32, nqb, nkv, (bf16*)dv_out);
        SP_LAUNCH_CHECK();
        return SP_OK;
    }
    // VALU path (fp32 storage, odd extents): one slab per block of TQ queries, summed in block order
    SP_CHECK_ARG(nkd % 4 == 0 && nkv % 4 == 0, "sp_attention_bwd: batch*nk*d and batch*nk*dv must be multiples of 4");
    const int nqb = sp_div_up(n, attn_valu_tq(nk, d, dv));
    int rc = dtype == SP_F32 ? launch_attn<float>(false, q, k, v, dq, dout, const_cast<float*>(lse), dk_f32, dv_f32, batch, n, nk, d, dv, s)
                             : launch_attn<bf16>(false, q, k, v, dq, dout, const_cast<float*>(lse), dk_f32, dv_f32, batch, n, nk, d, dv, s);
    if (rc != SP_OK) return rc;
    if (dtype == SP_F32) {
        hipLaunchKernelGGL(attn_reduce_kernel<float>, dim3(sp_div_up(nkd / 4, 256)), dim3(256), 0, s, dk_f32, nqb, nkd, (float*)dk);
        hipLaunchKernelGGL(attn_reduce_kernel<float>, dim3(sp_div_up(nkv / 4, 256)), dim3(256), 0, s, dv_f32, nqb, nkv, (float*)dv_out);
    } else {
        hipLaunchKernelGGL(attn_reduce_kernel<bf16>, dim3(sp_div_up(nkd / 4, 256)), dim3(256), 0, s, dk_f32, nqb, nkd, (bf16*)dk);
        hipLaunchKernelGGL(attn_reduce_kernel<bf16>, dim3(sp_div_up(nkv / 4, 256)), dim3(256), 0, s, dv_f32, nqb, nkv, (bf16*)dv_out);
    }
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_attention_bwd_slabs(int32_t n, int32_t nk, int32_t d, int32_t dv, int32_t dtype, int64_t* slabs_out) {
    SP_CHECK_ARG(slabs_out && n > 0 && nk > 0 && d > 0 && dv > 0, "sp_attention_bwd_slabs: bad args");
    *slabs_out = attn_bwd_mfma_ok(dtype, nk, d, dv) ? sp_div_up(n, AB_QB) : sp_div_up(n, attn_valu_tq(nk, d, dv));
    return SP_OK;
}
