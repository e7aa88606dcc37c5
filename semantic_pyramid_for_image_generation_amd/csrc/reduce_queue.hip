// Deferred slab reductions of the streaming weight-gradient kernels (conv_wgrad_1x1.hip), compiled ONCE (the slabs and dW are fp32 in
// both 16-bit flavours of the library).
//
// A pixel-split weight-gradient launch leaves nsplit partial tiles in its workspace and a second, tiny kernel adds them to dW in a fixed
// order.  31 of those per training step took 4.9 us each for 2.4 MB of slabs - launch tail, not traffic - and nothing reads dW before
// the batched spectral-norm backward at the end of the pass.  With sp_wgrad_reduce_defer(1) the launchers queue the reduction instead
// (spq_push_reduce) and sp_wgrad_reduce_flush(stream) runs everything queued in ONE launch per 48 entries (descriptors by value:
// capturable in a HIP graph).  Every entry is summed exactly as the separate kernel would have (same blocks, same order): bit-identical.
// The caller keeps the workspaces alive until the flush and flushes on the stream the weight-gradient kernels ran on.
#include <vector>
#include "common.h"

namespace {

struct RedEntry {
    const float* slabs; float* dw; const float* bias_slabs; float* dbias;
    long n_dw;
    int nsplit, bias_ld, cout, first_block;
};
constexpr int RED_MAX = 48;
struct RedBatch { RedEntry e[RED_MAX]; int n; };

std::vector<RedEntry> g_pending;
int g_defer = 0;

// the body of conv_wgrad_1x1.hip's wgrad1x1_reduce_kernel for block `blk` of entry r: a block owns 16 float4 columns (dW columns,
// then bias columns); its 16 thread rows take the slabs s = g, g + 16, ... and meet in LDS in a fixed order
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(RedBatch b) {
    __shared__ float4 red[16][16];
    int ei = 0;
#pragma unroll 1
    for (int k = 1; k < b.n; ++k)
        if ((int)blockIdx.x >= b.e[k].first_block) ei = k;
    const RedEntry& r = b.e[ei];
    const int blk = (int)blockIdx.x - r.first_block;
    const int c16 = threadIdx.x & 15, g = threadIdx.x >> 4;
    const long cols4 = r.n_dw / 4;
    const long bias4 = r.bias_slabs != nullptr ? r.bias_ld / 4 : 0;
    const long c = (long)blk * 16 + c16;
    const float* src = nullptr;
    long stride = 0;
    if (c < cols4) { src = r.slabs + c * 4; stride = r.n_dw; }
    else if (c < cols4 + bias4) { src = r.bias_slabs + (c - cols4) * 4; stride = r.bias_ld; }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (src != nullptr) {
#pragma unroll 4
        for (int k = g; k < r.nsplit; k += 16) {
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
        float4 d = *reinterpret_cast<float4*>(r.dw + c * 4);
        d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
        *reinterpret_cast<float4*>(r.dw + c * 4) = d;
    } else {
        const int co = (int)(c - cols4) * 4;
        const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (co + q < r.cout) r.dbias[co + q] += v[q];
    }
}

}  // namespace

// conv_wgrad_1x1.hip (either flavour): true = queued, the caller does not launch its reduce kernel
bool spq_push_reduce(const float* slabs, int nsplit, long n_dw, float* dw, const float* bias_slabs, int bias_ld, int cout, float* dbias) {
    if (!g_defer) return false;
    // two queued entries must not add to the same dW: the second would race with the first inside one launch (the separate kernels
    // were ordered by the stream) - such a launch keeps its own reduce kernel
    for (const RedEntry& e : g_pending)
        if (e.dw == dw || (dbias != nullptr && e.dbias == dbias)) return false;
    RedEntry r;
    r.slabs = slabs; r.dw = dw; r.bias_slabs = bias_slabs; r.dbias = dbias;
    r.n_dw = n_dw; r.nsplit = nsplit; r.bias_ld = bias_ld; r.cout = cout; r.first_block = 0;
    g_pending.push_back(r);
    return true;
}

extern "C" int sp_wgrad_reduce_defer(int32_t on) {
    g_defer = on ? 1 : 0;               // (what is queued stays queued until the flush)
    return SP_OK;
}

extern "C" int sp_wgrad_reduce_pending(void) { return (int)g_pending.size(); }

// on != 0: launch what is queued on `stream`; on == 0: drop it (a pass that was abandoned - its workspaces are gone)
extern "C" int sp_wgrad_reduce_flush(int32_t run, sp_stream_t stream) {
    if (!run) { g_pending.clear(); return SP_OK; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    size_t i = 0;
    while (i < g_pending.size()) {
        RedBatch b;
        b.n = 0;
        int blocks = 0;
        for (; i < g_pending.size() && b.n < RED_MAX; ++i) {
            RedEntry r = g_pending[i];
            const long cols = r.n_dw / 4 + (r.bias_slabs != nullptr ? r.bias_ld / 4 : 0);
            r.first_block = blocks;
            blocks += (int)((cols + 15) / 16);
            b.e[b.n++] = r;
        }
        hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, s, b);
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) { g_pending.clear(); sp_set_error("sp_wgrad_reduce_flush: HIP launch failed: %s", hipGetErrorString(e_)); return SP_ERR_LAUNCH; }
    }
    g_pending.clear();
    return SP_OK;
}
