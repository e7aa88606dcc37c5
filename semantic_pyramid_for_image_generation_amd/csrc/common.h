// Shared device helpers for the sempyr HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sempyr.h"

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef float sp_f32x2_t __attribute__((ext_vector_type(2)));

// Storage element types.  Arithmetic is always fp32; T only selects the HBM format.
//
// THE 16-BIT FLAVOUR IS A COMPILE-TIME CHOICE OF THE TRANSLATION UNIT.  Every kernel source is compiled twice (build.sh): as is -
// `bf16` = bfloat16, the MFMA is v_mfma_f32_16x16x32_bf16 (SP_BF16) - and with -DSP_H16_FP16, where the SAME 16-bit code paths
// store IEEE half precision and multiply on v_mfma_f32_16x16x32_f16 (SP_F16: BASELINE.json config 5's "fp16 activations"; 10
// mantissa bits instead of 7).  Layouts, LDS images, transposed reads and schedules are 16-bit agnostic; what differs is exactly
// what is defined in this block: the two conversions, the MFMA, the constant 1.0.  The second compilation's entry points carry the
// suffix __h16 (build/rename_h16.h) and are reached through the dispatcher generated from include/sempyr.h (tools/gen_h16.py).
struct bf16 { uint16_t v; };
#ifdef SP_H16_FP16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8_t;
typedef _Float16 sp_h16x2_t __attribute__((ext_vector_type(2)));
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define SP_H16_ONE_PAIR 0x3C003C00u                       // two 1.0 values
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (uint16_t)h); }
// fp32 -> fp16, round-to-nearest-even (v_cvt_pk_f16_f32 on gfx950); values beyond 65504 become inf - the host keeps the
// activation gradients in range with a static loss scale (ops.py)
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
    const sp_f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, sp_h16x2_t));
}
// the two halves of a packed pair as fp32
__device__ __forceinline__ float h16_lo_to_f32(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float h16_hi_to_f32(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __bf16 sp_bf16x2_t __attribute__((ext_vector_type(2)));
#define SP_H16_ONE_PAIR 0x3F803F80u
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
// fp32 -> bf16, round-to-nearest-even, NaN kept quiet: gfx950 has the packed conversion in hardware (v_cvt_pk_bf16_f32, one
// instruction per two values; the integer sequence it replaces cost 7 VALU per value - measurable in every bf16 epilogue)
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
    const sp_f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, sp_bf16x2_t));
}
__device__ __forceinline__ float h16_lo_to_f32(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h16_hi_to_f32(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) { return f32x2_to_bf16x2(f, 0.f) & 0xffffu; }

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int PER16 = 4;          // elements per 16-byte chunk
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ void ld4(const float* p, float o[4]) {
        float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    static __device__ __forceinline__ void st4(float* p, const float o[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
    static __device__ __forceinline__ void st2(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
};
template <> struct Elem<bf16> {
    static constexpr int PER16 = 8;
    static __device__ __forceinline__ float ld(const bf16* p) { return bf16_bits_to_f32(p->v); }
    static __device__ __forceinline__ void st(bf16* p, float v) { p->v = (uint16_t)f32_to_bf16_bits(v); }
    static __device__ __forceinline__ void ld4(const bf16* p, float o[4]) {
        uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = bf16_bits_to_f32(v.x & 0xffffu); o[1] = bf16_bits_to_f32(v.x >> 16);
        o[2] = bf16_bits_to_f32(v.y & 0xffffu); o[3] = bf16_bits_to_f32(v.y >> 16);
    }
    static __device__ __forceinline__ void st4(bf16* p, const float o[4]) {
        uint2 v;
        v.x = f32x2_to_bf16x2(o[0], o[1]);
        v.y = f32x2_to_bf16x2(o[2], o[3]);
        *reinterpret_cast<uint2*>(p) = v;
    }
    static __device__ __forceinline__ void st2(bf16* p, float a, float b) {
        *reinterpret_cast<uint32_t*>(p) = f32x2_to_bf16x2(a, b);
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* scratch /* >= 4 floats, LDS */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == SP_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == SP_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == SP_ACT_TANH) return tanhf(v);
    return v;
}

// the same on N values with ONE (uniform) branch on `act`: the per-element form above expands to a branch - and an inlined
// tanhf - per value when it is called in an unrolled loop (7.8 K scalar instructions in the tall convolution kernel)
template <int N, bool TANH_OK = true>
__device__ __forceinline__ void apply_act_vec(float (&v)[N], int act) {
    if (act == SP_ACT_NONE) return;
    if (act == SP_ACT_LRELU) {
        // max(v, 0.2 v) is LeakyReLU(0.2) bit for bit (v > 0: v > 0.2 v; v < 0: 0.2 v > v; +-0 and NaN likewise), in one multiply
        // (packed by the compiler) and one v_max instead of multiply + compare + select; plain asm: fmaxf() adds a canonicalising
        // v_max in front of every operand
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const float s = 0.2f * v[r];
            asm("v_max_f32 %0, %1, %2" : "=v"(v[r]) : "v"(v[r]), "v"(s));
        }
    } else if (act == SP_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] = fmaxf(v[r], 0.f);
    } else if (TANH_OK && act == SP_ACT_TANH) {
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] = tanhf(v[r]);
    }
}

// ---- host-side error plumbing (defined in api.cpp) -----------------------------------------
extern "C" void sp_set_error(const char* fmt, ...);
// the kernel a conv / weight-gradient entry point chose for its last launch on this thread (sp_last_route, bench.py's table)
extern "C" void sp_note_route(const char* name);
#define SP_CHECK_ARG(cond, ...) do { if (!(cond)) { sp_set_error(__VA_ARGS__); return SP_ERR_INVALID; } } while (0)
#define SP_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    sp_set_error("%s:%d HIP launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return SP_ERR_LAUNCH; } } while (0)

static inline int sp_div_up(long a, long b) { return (int)((a + b - 1) / b); }

// V elements (16 bytes when possible) per lane
template <typename T, int V> struct VecIO;
template <> struct VecIO<float, 4> {
    static __device__ __forceinline__ void ld(const float* p, float (&o)[4]) { Elem<float>::ld4(p, o); }
    static __device__ __forceinline__ void st(float* p, const float (&o)[4]) { Elem<float>::st4(p, o); }
};
template <> struct VecIO<bf16, 4> {
    static __device__ __forceinline__ void ld(const bf16* p, float (&o)[4]) { Elem<bf16>::ld4(p, o); }
    static __device__ __forceinline__ void st(bf16* p, const float (&o)[4]) { Elem<bf16>::st4(p, o); }
};
template <> struct VecIO<bf16, 8> {
    static __device__ __forceinline__ void ld(const bf16* p, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        o[0] = bf16_bits_to_f32(v.x & 0xffffu); o[1] = bf16_bits_to_f32(v.x >> 16);
        o[2] = bf16_bits_to_f32(v.y & 0xffffu); o[3] = bf16_bits_to_f32(v.y >> 16);
        o[4] = bf16_bits_to_f32(v.z & 0xffffu); o[5] = bf16_bits_to_f32(v.z >> 16);
        o[6] = bf16_bits_to_f32(v.w & 0xffffu); o[7] = bf16_bits_to_f32(v.w >> 16);
    }
    static __device__ __forceinline__ void st(bf16* p, const float (&o)[8]) {
        uint4 v;
        v.x = f32x2_to_bf16x2(o[0], o[1]);
        v.y = f32x2_to_bf16x2(o[2], o[3]);
        v.z = f32x2_to_bf16x2(o[4], o[5]);
        v.w = f32x2_to_bf16x2(o[6], o[7]);
        *reinterpret_cast<uint4*>(p) = v;
    }
};

// knobs (sp_set_tuning, api.cpp); -1 = default
extern int* const sp_g_tune;
static inline int sp_tune(int key, int dflt) { return sp_g_tune[key] >= 0 ? sp_g_tune[key] : dflt; }
// fixed-order reductions (no fp32 atomics): always in the fp32 parity mode, on request in the bf16 throughput mode
static inline bool sp_deterministic(int dtype) { return sp_g_tune[SP_TUNE_DETERMINISTIC] >= 0 ? sp_g_tune[SP_TUNE_DETERMINISTIC] != 0 : dtype == SP_F32; }
// conv_pp.hip: SP_OK after launching, 1 if the shape is not covered (bf16 3x3, Cout > 64, th x 32 patches with th = 8 / 16)
int sp_conv_pp_launch(const sp_conv_params& p, int th, hipStream_t s);
long sp_conv_pp_split_workspace(int n, int h, int w, int cin_p, int cout);   // conv_pp.hip: scratch for the K-split of the last partial round
long sp_conv_pp_split_workspace_w16(int n, int h, int cin_p, int cout);      // ... of its 16-pixel-wide tiles
long sp_conv_pp_rounds100(long total, int cin_p, long workspace_bytes);      // ... and what a launch of `total` 8-row items costs with it
int sp_conv_ppw_launch(const sp_conv_params& p, hipStream_t s);      // conv_ppw.hip: 128 co x 16 x 32 px, 64 co x 4 rows per wave
long sp_conv_ppw_split_workspace(int n, int h, int w, int cin_p, int cout);   // conv_ppw.hip: scratch of ITS K-split of the last partial round
long sp_conv_ppw_rounds100(long total, int cin_p, long workspace_bytes);      // ... and the cost of `total` 16-row items with it
int sp_conv_ppw_covers(const sp_conv_params& p);                    // ... whether it takes the launch at all (shape, epilogue)
// reduce_queue.hip (compiled once, shared by both flavours): true = the slab reduction was queued for sp_wgrad_reduce_flush
bool spq_push_reduce(const float* slabs, int nsplit, long n_dw, float* dw, const float* bias_slabs, int bias_ld, int cout, float* dbias);
// conv_wgrad_rows.hip: SP_OK after launching, 1 if the shape is not covered
int sp_wgrad_rows_launch(const void* x, const void* dy, float* dw, float* dbias, int n, int h, int w, int cin, int cout,
                         int ld_dy, float* ws, long ws_floats, int dy_up2, hipStream_t s);
long sp_wgrad_rows_workspace(int n, int h, int w, int cin, int cout);
int sp_wgrad_rows_launch_pair(const void* x, const void* dy, float* dw_a, float* dbias_a, float* dw_b, float* dbias_b, int n, int split,
                              int h, int w, int cin, int cout, int ld_dy, float* ws, long ws_floats, int dy_up2, hipStream_t s);
// conv_wgrad_1x1.hip: same contract for the bf16 1x1 layers
int sp_wgrad1x1_launch(const void* x, const void* dy, float* dw, float* dbias, int n, int h, int w, int cin, int cout, int ld_dy,
                       float* ws, long ws_floats, hipStream_t s);
long sp_wgrad1x1_workspace(int n, int h, int w, int cin, int cout, int ld_dy);
int sp_wgrad1x1_launch_pair(const void* x, const void* dy, float* dw_a, float* dbias_a, float* dw_b, float* dbias_b, int n, int split, int h, int w,
                            int cin, int cout, int ld_dy, float* ws, long ws_floats, hipStream_t s);
int sp_wgrad3x3_cin8_launch_pair(const void* x, const void* dy, float* dw_a, float* dbias_a, float* dw_b, float* dbias_b, int n, int split, int h,
                                 int w, int cout, int ld_dy, float* ws, long ws_floats, hipStream_t s);
// ... and for the 3x3 layers with an 8-channel input
int sp_wgrad3x3_cin8_launch(const void* x, const void* dy, float* dw, float* dbias, int n, int h, int w, int cout, int ld_dy, float* ws,
                            long ws_floats, hipStream_t s);
long sp_wgrad3x3_cin8_workspace(int n, int h, int w, int cout, int ld_dy);
