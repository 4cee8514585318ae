// Shared host/device helpers for libspectrum_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/spectrum_hip.h"

namespace sv {

// Text of the calling thread's last failure (returned by sv_last_error()).
extern thread_local char g_last_error[512];

inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SV_E_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return SV_OK;
}

#define SV_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return sv::fail(SV_E_INVALID, __VA_ARGS__); \
    } while (0)

#define SV_F32_ONLY(dtype, name)                                                             \
    do {                                                                                     \
        if ((dtype) != SV_F32) return sv::fail(SV_E_UNSUPPORTED, name ": dtype %d unsupported", (int)(dtype)); \
    } while (0)

#define SV_F32_OR_BF16(dtype, name)                                                          \
    do {                                                                                     \
        if ((dtype) != SV_F32 && (dtype) != SV_BF16)                                         \
            return sv::fail(SV_E_UNSUPPORTED, name ": dtype %d unsupported", (int)(dtype));  \
    } while (0)

inline hipStream_t to_stream(sv_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Plan-time options (sv_set_option): which of several equivalent kernel bodies or launch shapes a planner picks.  None
// changes what an entry point computes; tests hold the bodies against each other through them and measurements sweep
// launch shapes.  -1 (the initial value of every option) leaves the choice to the planner.  The environment is never
// read on a launch path.
enum SvOpt {
    OPT_DGRAD_NHWC,      // 0: input gradients of the 3x3 family through the planar kernels
    OPT_DGRAD_PHASE,     // 0: upsampled layers in the direct form (result pixels of the upsampled image, 2x2 sums)
    OPT_DGRAD_PAIR,      // 0: one pixel per result column in the 8-channel layer
    OPT_DGRAD_GRID,      // > 0: workgroups of an input-gradient launch
    OPT_DGRAD_VALU,      // 1: planar input-gradient kernels on the vector pipe only
    OPT_WGRAD_NHWC,      // 0: kernel gradients of the 3x3 family through the planar kernels
    OPT_WGRAD_VALU,      // 1: planar kernel-gradient kernels on the vector pipe only
    OPT_NETFWD_GENERIC,  // 1: the run-time-table forward kernel for every layer table
    OPT_NETFWD_PAIR,     // 0: one sample per thread in the first layer of the fused forward
    OPT_NETFWD_PHASE,    // 0: upsampled layers of the fused forward in the direct form
    OPT_NETFWD_VALU,     // 1: every layer of the fused forward on the vector pipe
    OPT_NETFWD_S,        // > 0: cap on the samples a workgroup of the fused forward carries
    OPT_NETFWD_GRID,     // > 0: cap on the workgroups of the fused forward
    OPT_SMALL_CONVT,     // 0: few-channel transposed convs through the implicit GEMM; 2: on the vector pipe only; 3: f32 ones on the matrix cores at any size
    OPT_S1_MFMA,         // 0: stride-1 SAME convs outside the few-channel family through the planar kernels
    OPT_WGRAD_STORED,    // 0: kernel gradients of upsampled layers over the pixels of the upsampled image
    OPT_DGRAD_CHAIN,     // 0: sv_conv2d_dgrad_chain declines every list (one launch per layer); > 1: its workgroups
    OPT_COUNT
};
int opt(int id);  // api.hip

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8, never more than the work.
inline int ew_grid(int64_t n_items, int block = 256, int max_blocks = 2048) {
    int64_t b = ceil_div(n_items, block);
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

// ---------------------------------------------------------------- device helpers
// exp(z) - 1 for z <= 0 in ~10 instructions: a degree-6 Taylor polynomial near zero (error < 1e-8 of the result for
// |z| < 0.25, where exp(z) - 1 would cancel) and the hardware exponential elsewhere (absolute error < 1.2e-7 on a
// result of magnitude >= 0.22).
__device__ __forceinline__ float fast_expm1_neg(float z) {
    const float p = z * fmaf(z, fmaf(z, fmaf(z, fmaf(z, fmaf(z, 1.f / 720.f, 1.f / 120.f), 1.f / 24.f), 1.f / 6.f), 0.5f), 1.f);
    const float e = __expf(z) - 1.f;
    return z > -0.25f ? p : e;
}
// tanh: odd Taylor polynomial for |z| < 0.125 (error < 3e-9), 1 - 2 / (exp(2|z|) + 1) with the hardware exponential and
// reciprocal elsewhere (absolute error < 2e-7).
__device__ __forceinline__ float fast_tanh(float z) {
    const float a = fabsf(z);
    const float z2 = z * z;
    const float p = z * fmaf(z2, fmaf(z2, fmaf(z2, -17.f / 315.f, 2.f / 15.f), -1.f / 3.f), 1.f);
    const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * a) + 1.f);
    return a < 0.125f ? p : copysignf(t, z);
}

__device__ __forceinline__ float act_apply(float z, int act, float alpha) {
    switch (act) {
        case SV_ACT_RELU: return z > 0.f ? z : 0.f;
        case SV_ACT_ELU: return z > 0.f ? z : fast_expm1_neg(z);
        case SV_ACT_LRELU: return z >= 0.f ? z : alpha * z;
        case SV_ACT_TANH: return fast_tanh(z);
        case SV_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.f + __expf(-z));
        default: return z;
    }
}

// d act / d z written in terms of the activation output y.  At y == 0 the one-sided choice is TensorFlow's
// (ReluGrad / LeakyReluGrad test `features > 0`).
__device__ __forceinline__ float act_grad_from_y(float y, int act, float alpha) {
    switch (act) {
        case SV_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case SV_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
        case SV_ACT_LRELU: return y > 0.f ? 1.f : alpha;
        case SV_ACT_TANH: return 1.f - y * y;
        case SV_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// The same derivative as a branch-free quadratic in y with one coefficient set per sign of y, so a staging loop
// does not carry a switch: d = c0 + c1*y + c2*y^2, coefficients chosen by (y > 0).
struct ActGradCoef {
    float p0, p1, p2;  // y > 0
    float n0, n1, n2;  // y <= 0
};
inline ActGradCoef act_grad_coef(int act, float alpha) {
    switch (act) {
        case SV_ACT_RELU: return {1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        case SV_ACT_ELU: return {1.f, 0.f, 0.f, 1.f, 1.f, 0.f};
        case SV_ACT_LRELU: return {1.f, 0.f, 0.f, alpha, 0.f, 0.f};
        case SV_ACT_TANH: return {1.f, 0.f, -1.f, 1.f, 0.f, -1.f};
        case SV_ACT_SIGMOID: return {0.f, 1.f, -1.f, 0.f, 1.f, -1.f};
        default: return {1.f, 0.f, 0.f, 1.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ ActGradCoef act_grad_coef_dev(int act, float alpha) {
    switch (act) {
        case SV_ACT_RELU: return {1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        case SV_ACT_ELU: return {1.f, 0.f, 0.f, 1.f, 1.f, 0.f};
        case SV_ACT_LRELU: return {1.f, 0.f, 0.f, alpha, 0.f, 0.f};
        case SV_ACT_TANH: return {1.f, 0.f, -1.f, 1.f, 0.f, -1.f};
        case SV_ACT_SIGMOID: return {0.f, 1.f, -1.f, 0.f, 1.f, -1.f};
        default: return {1.f, 0.f, 0.f, 1.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ float act_grad_q(float y, const ActGradCoef& k) {
    const bool pos = y > 0.f;
    const float c0 = pos ? k.p0 : k.n0, c1 = pos ? k.p1 : k.n1, c2 = pos ? k.p2 : k.n2;
    return fmaf(fmaf(c2, y, c1), y, c0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a 256-thread block; result valid in every thread. `red` is 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wid] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// splitmix64-style counter hash -> uniform in [0,1)
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float u01(uint64_t seed, uint64_t ctr) {
    return (float)(mix64(seed ^ mix64(ctr)) >> 40) * (1.0f / 16777216.0f);
}

// conv_small.hip: input gradient / transposed forward with an image side of <= 4 channels (1: geometry not covered)
int launch_convt_small(const sv_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act,
                       float alpha, hipStream_t st);
int launch_convt_small_bf16(const sv_conv_geom& g, const void* x, const void* w, const float* bias, void* y, int act,
                            float alpha, hipStream_t st);

// Deferred reductions (conv_direct.hip): true when the "dst (+)= sum of `slabs` rows of n floats at src" step was queued
// for the flush instead of being launched now.
bool defer_reduce(const float* src, float* dst, int64_t n, int slabs, int accumulate);
void deferred_begin();
// For a call whose slabs live in memory the caller does not own until the flush: the calling thread's bracket is set
// aside (returns whether one was open) so that the reduction launches at once, then put back.
bool deferred_suspend();
void deferred_resume(bool was);
int deferred_flush(hipStream_t st, bool launch);

}  // namespace sv
