// Generic direct convolution kernels (any kernel size, stride, padding, channel count; f32).
//
// These are the shape-agnostic fallbacks: one lane per output (or input) pixel, a register tile of
// channels per lane, filter taps read through wave-uniform addresses so the compiler keeps them on the
// scalar path.  Shapes on the benchmarked path are routed to the tiled kernels in conv_tiled.hip first.
#include <stdarg.h>

#include <vector>

#include "common.h"

namespace sv {

// conv_gemm.hip: implicit GEMM on the matrix cores for the GEMM-shaped geometries
bool conv_gemm_worthwhile(const sv_conv_geom& g, int mode);
int conv_gemm_wgrad_splits(const sv_conv_geom& g);
size_t conv_gemm_wgrad_workspace(const sv_conv_geom& g);
size_t conv_gemm_splitk_workspace(const sv_conv_geom& g, int mode);
int launch_conv_gemm_fwd(const sv_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act,
                         float alpha, void* ws, size_t ws_bytes, hipStream_t st, const float* xmask = nullptr);
int launch_conv_gemm_dgrad(const sv_conv_geom& g, const float* dz, const float* w, const float* bias, float* dx, int act,
                           float alpha, void* ws, size_t ws_bytes, hipStream_t st);
int launch_conv_gemm_wgrad(const sv_conv_geom& g, const float* x, const float* dz, float* slabs, int splits, hipStream_t st,
                           int accumulate, const float* xmask = nullptr);
// conv_gemm_bf16.hip: the same three products with bf16 operands
bool conv_gemm_bf16_ok(const sv_conv_geom& g, int mode);
int conv_gemm_bf16_wgrad_splits(const sv_conv_geom& g);
int launch_conv_gemm_bf16_fwd(const sv_conv_geom& g, const void* x, const void* w, const float* bias, void* y, int act,
                              float alpha, void* ws, size_t ws_bytes, hipStream_t st);
int launch_conv_gemm_bf16_dgrad(const sv_conv_geom& g, const void* dz, const void* w, const float* bias, void* dx, int act,
                                float alpha, void* ws, size_t ws_bytes, hipStream_t st);
int launch_conv_gemm_bf16_wgrad(const sv_conv_geom& g, const void* x, const void* dz, float* slabs, int splits,
                                hipStream_t st, int accumulate);

// ------------------------------------------------------------------------------------------------ forward
// y[n,oh,ow,co] = act(bias[co] + sum_{kh,kw,ci} x[n, (oh*SH+kh-pt)/up_h, (ow*SW+kw-pl)/up_w, ci] * w[kh,kw,ci,co])
template <int CO_T, typename T>
__global__ __launch_bounds__(256) void conv2d_fwd_direct(sv_conv_geom g, const T* __restrict__ x,
                                                         const T* __restrict__ w,
                                                         const float* __restrict__ bias, T* __restrict__ y,
                                                         int act, float alpha) {
    const int64_t npix = (int64_t)g.N * g.Ho * g.Wo;
    const int co0 = blockIdx.y * CO_T;
    const int Hp = g.H / g.up_h, Wp = g.W / g.up_w;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const int ow = (int)(p % g.Wo);
        const int oh = (int)((p / g.Wo) % g.Ho);
        const int n = (int)(p / ((int64_t)g.Wo * g.Ho));
        float acc[CO_T];
#pragma unroll
        for (int c = 0; c < CO_T; ++c) acc[c] = (bias != nullptr && co0 + c < g.Cout) ? bias[co0 + c] : 0.f;
        for (int kh = 0; kh < g.KH; ++kh) {
            const int ihl = oh * g.SH + kh - g.pad_t;
            const bool okh = ihl >= 0 && ihl < g.H;
            const int ih = okh ? ihl / g.up_h : 0;
            for (int kw = 0; kw < g.KW; ++kw) {
                const int iwl = ow * g.SW + kw - g.pad_l;
                const bool ok = okh && iwl >= 0 && iwl < g.W;
                const int iw = ok ? iwl / g.up_w : 0;
                const T* xp = x + (((int64_t)n * Hp + ih) * Wp + iw) * g.Cin;
                const T* wp = w + (int64_t)(kh * g.KW + kw) * g.Cin * g.Cout + co0;
                if (ok) {
                    for (int ci = 0; ci < g.Cin; ++ci) {
                        const float xv = (float)xp[ci];
#pragma unroll
                        for (int c = 0; c < CO_T; ++c)
                            if (co0 + c < g.Cout) acc[c] = fmaf(xv, (float)wp[(int64_t)ci * g.Cout + c], acc[c]);
                    }
                }
            }
        }
        T* yp = y + p * g.Cout + co0;
#pragma unroll
        for (int c = 0; c < CO_T; ++c)
            if (co0 + c < g.Cout) yp[c] = (T)act_apply(acc[c], act, alpha);
    }
}

// ------------------------------------------------------------------------------------------------ input gradient
// out[n,ihp,iwp,ci] = epi( sum over logical (ih,iw) mapped to this stored pixel, kh, kw, co with
//                         (ih+pt-kh) % SH == 0:  dz[n,(ih+pt-kh)/SH,(iw+pl-kw)/SW,co] * w[kh,kw,ci,co] )
// epi = optional bias[ci] + activation: that is the forward of Conv2DTranspose.
template <int CI_T, typename T>
__global__ __launch_bounds__(256) void conv2d_dgrad_direct(sv_conv_geom g, const T* __restrict__ dz,
                                                           const T* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ dx,
                                                           int act, float alpha) {
    const int Hp = g.H / g.up_h, Wp = g.W / g.up_w;
    const int64_t npix = (int64_t)g.N * Hp * Wp;
    const int ci0 = blockIdx.y * CI_T;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
        const int iwp = (int)(p % Wp);
        const int ihp = (int)((p / Wp) % Hp);
        const int n = (int)(p / ((int64_t)Wp * Hp));
        float acc[CI_T];
#pragma unroll
        for (int c = 0; c < CI_T; ++c) acc[c] = 0.f;
        for (int a = 0; a < g.up_h; ++a) {
            const int ih = ihp * g.up_h + a;
            for (int kh = 0; kh < g.KH; ++kh) {
                const int t = ih + g.pad_t - kh;
                const bool okh = t >= 0 && (t % g.SH) == 0 && (t / g.SH) < g.Ho;
                const int oh = okh ? t / g.SH : 0;
                for (int b = 0; b < g.up_w; ++b) {
                    const int iw = iwp * g.up_w + b;
                    for (int kw = 0; kw < g.KW; ++kw) {
                        const int u = iw + g.pad_l - kw;
                        const bool ok = okh && u >= 0 && (u % g.SW) == 0 && (u / g.SW) < g.Wo;
                        const int ow = ok ? u / g.SW : 0;
                        const T* zp = dz + (((int64_t)n * g.Ho + oh) * g.Wo + ow) * g.Cout;
                        const T* wp = w + ((int64_t)(kh * g.KW + kw) * g.Cin + ci0) * g.Cout;
                        if (ok) {
                            for (int co = 0; co < g.Cout; ++co) {
                                const float zv = (float)zp[co];
#pragma unroll
                                for (int c = 0; c < CI_T; ++c)
                                    if (ci0 + c < g.Cin) acc[c] = fmaf(zv, (float)wp[(int64_t)c * g.Cout + co], acc[c]);
                            }
                        }
                    }
                }
            }
        }
        T* op = dx + p * g.Cin + ci0;
#pragma unroll
        for (int c = 0; c < CI_T; ++c)
            if (ci0 + c < g.Cin) {
                float v = acc[c] + (bias != nullptr ? bias[ci0 + c] : 0.f);
                op[c] = (T)act_apply(v, act, alpha);
            }
    }
}

// ------------------------------------------------------------------------------------------------ kernel gradient
// part[split][k][co] = sum over the split's output pixels of x[n, ih, iw, ci] * dz[n, oh, ow, co], k = (kh,kw,ci).
// grid = (K_total, cout tiles, n_split); the slab of partial sums is folded in split order by wgrad_reduce so
// the result does not depend on scheduling.  SWAP exchanges the roles of the two tensors' channels in the
// OUTPUT layout (dw[kh,kw,ci,co] vs the Conv2DTranspose layout, which is the same memory order) - unused here,
// the transposed wrapper swaps its operands instead.
template <int CO_T, typename T>
__global__ __launch_bounds__(256) void conv2d_wgrad_direct(sv_conv_geom g, const T* __restrict__ x,
                                                           const T* __restrict__ dz, float* __restrict__ part,
                                                           int64_t pix_per_split) {
    __shared__ float red[4 * CO_T];
    const int k = blockIdx.x;
    const int ci = k % g.Cin;
    const int kw = (k / g.Cin) % g.KW;
    const int kh = k / (g.Cin * g.KW);
    const int co0 = blockIdx.y * CO_T;
    const int Hp = g.H / g.up_h, Wp = g.W / g.up_w;
    const int64_t npix = (int64_t)g.N * g.Ho * g.Wo;
    const int64_t p_begin = (int64_t)blockIdx.z * pix_per_split;
    int64_t p_end = p_begin + pix_per_split;
    if (p_end > npix) p_end = npix;
    float acc[CO_T];
#pragma unroll
    for (int c = 0; c < CO_T; ++c) acc[c] = 0.f;
    for (int64_t p = p_begin + threadIdx.x; p < p_end; p += blockDim.x) {
        const int ow = (int)(p % g.Wo);
        const int oh = (int)((p / g.Wo) % g.Ho);
        const int n = (int)(p / ((int64_t)g.Wo * g.Ho));
        const int ihl = oh * g.SH + kh - g.pad_t;
        const int iwl = ow * g.SW + kw - g.pad_l;
        if (ihl < 0 || ihl >= g.H || iwl < 0 || iwl >= g.W) continue;
        const float xv = (float)x[(((int64_t)n * Hp + ihl / g.up_h) * Wp + iwl / g.up_w) * g.Cin + ci];
        const T* zp = dz + p * g.Cout + co0;
#pragma unroll
        for (int c = 0; c < CO_T; ++c)
            if (co0 + c < g.Cout) acc[c] = fmaf(xv, (float)zp[c], acc[c]);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < CO_T; ++c) {
        const float s = wave_sum(acc[c]);
        if (lane == 0) red[wid * CO_T + c] = s;
    }
    __syncthreads();
    if (threadIdx.x < CO_T && co0 + threadIdx.x < g.Cout) {
        const int c = threadIdx.x;
        const float s = red[c] + red[CO_T + c] + red[2 * CO_T + c] + red[3 * CO_T + c];
        const int64_t kc = (int64_t)g.KH * g.KW * g.Cin * g.Cout;
        part[(int64_t)blockIdx.z * kc + (int64_t)k * g.Cout + co0 + c] = s;
    }
}

// dw[i] (+)= sum_sp part[sp][i].  64 consecutive outputs per block (one coalesced 256-byte row per slab),
// 16 waves each summing every 16th slab with 8 loads in flight, then a fixed-order fold through LDS:
// latency-parallel and bitwise reproducible.
__device__ __forceinline__ void wgrad_reduce_rows(const float* __restrict__ part, float* __restrict__ dw, int64_t kc,
                                                  int n_split, int accumulate, int block, float (&red)[16][64]) {
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)block * 64 + lane;
    float s = 0.f;
    if (i < kc) {
        int sp = grp;
        for (; sp + 7 * 16 < n_split; sp += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(sp + u * 16) * kc + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; sp < n_split; sp += 16) s += part[(int64_t)sp * kc + i];
    }
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && i < kc) {
        float t = accumulate ? dw[i] : 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][lane];
        dw[i] = t;
    }
}

__global__ __launch_bounds__(1024) void wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw,
                                                      int64_t kc, int n_split, int accumulate) {
    __shared__ float red[16][64];
    wgrad_reduce_rows(part, dw, kc, n_split, accumulate, blockIdx.x, red);
}

// ---- deferred reductions: between sv_deferred_reduce_begin and _flush (same host thread) every "sum the slabs into the
// gradient" step of the library is queued instead of launched; the flush runs all of them in ONE launch (a job table, one
// range of workgroups per job).  The caller keeps every queued job's slabs untouched until the flush.
constexpr int kMaxDeferredJobs = 48;
struct DeferredJob {
    const float* src;
    float* dst;
    int64_t n;
    int slabs, accumulate;
};
struct DeferredArgs {
    int n_jobs;
    int first[kMaxDeferredJobs + 1];
    DeferredJob job[kMaxDeferredJobs];
};
__global__ __launch_bounds__(1024) void reduce_multi_k(DeferredArgs m) {
    __shared__ float red[16][64];
    int j = 0;
    while (j + 1 < m.n_jobs && (int)blockIdx.x >= m.first[j + 1]) ++j;
    const DeferredJob& q = m.job[j];
    wgrad_reduce_rows(q.src, q.dst, q.n, q.slabs, q.accumulate, blockIdx.x - m.first[j], red);
}

static thread_local bool t_defer_on = false;
static thread_local std::vector<DeferredJob> t_deferred;

bool defer_reduce(const float* src, float* dst, int64_t n, int slabs, int accumulate) {
    if (!t_defer_on) return false;
    t_deferred.push_back(DeferredJob{src, dst, n, slabs, accumulate});
    return true;
}

bool deferred_suspend() {
    const bool was = t_defer_on;
    t_defer_on = false;
    return was;
}

void deferred_resume(bool was) { t_defer_on = was; }

void deferred_begin() {
    t_deferred.clear();
    t_defer_on = true;
}

int deferred_flush(hipStream_t st, bool launch) {
    t_defer_on = false;
    if (!launch) {
        t_deferred.clear();
        return SV_OK;
    }
    size_t done = 0;
    while (done < t_deferred.size()) {
        DeferredArgs m{};
        int blocks = 0;
        while (done < t_deferred.size() && m.n_jobs < kMaxDeferredJobs) {
            m.job[m.n_jobs] = t_deferred[done++];
            m.first[m.n_jobs] = blocks;
            blocks += (int)ceil_div(m.job[m.n_jobs].n, 64);
            ++m.n_jobs;
        }
        m.first[m.n_jobs] = blocks;
        hipLaunchKernelGGL(reduce_multi_k, dim3((unsigned)blocks), dim3(1024), 0, st, m);
        int rc = check_launch("reduce_multi");
        if (rc) {
            t_deferred.clear();
            return rc;
        }
    }
    t_deferred.clear();
    return SV_OK;
}

// dst (+)= the fixed-order sum of `slabs` rows of n floats at part: queued when a deferred bracket is open, else launched
int launch_slab_sum(const float* part, float* dst, int64_t n, int slabs, int accumulate, hipStream_t st) {
    if (defer_reduce(part, dst, n, slabs, accumulate)) return SV_OK;
    hipLaunchKernelGGL(wgrad_reduce, dim3((unsigned)ceil_div(n, 64)), dim3(1024), 0, st, part, dst, n, slabs, accumulate);
    return check_launch("wgrad_reduce");
}

// ------------------------------------------------------------------------------------------------ host side
int validate_geom(const sv_conv_geom* g, const char* who) {
    if (g == nullptr) return fail(SV_E_INVALID, "%s: null geometry", who);
    if (g->N <= 0 || g->H <= 0 || g->W <= 0 || g->Cin <= 0 || g->Cout <= 0 || g->KH <= 0 || g->KW <= 0 ||
        g->SH <= 0 || g->SW <= 0 || g->Ho <= 0 || g->Wo <= 0 || g->up_h <= 0 || g->up_w <= 0 || g->pad_t < 0 ||
        g->pad_l < 0)
        return fail(SV_E_INVALID, "%s: non-positive extent in geometry", who);
    if (g->H % g->up_h != 0 || g->W % g->up_w != 0)
        return fail(SV_E_INVALID, "%s: logical extent %dx%d is not a multiple of the upsample factor %dx%d", who,
                    g->H, g->W, g->up_h, g->up_w);
    // every output pixel's window must start inside the padded input: (Ho-1)*SH - pad_t <= H-1
    if ((int64_t)(g->Ho - 1) * g->SH - g->pad_t > g->H - 1 || (int64_t)(g->Wo - 1) * g->SW - g->pad_l > g->W - 1)
        return fail(SV_E_INVALID, "%s: output extent %dx%d does not fit input %dx%d", who, g->Ho, g->Wo, g->H, g->W);
    return SV_OK;
}

// Optional scratch of the forward (mode 0) / input-gradient (mode 1) launches: lets a result with few tiles cut its
// reduction over more workgroups.  0 when the geometry gains nothing from it.
size_t conv_direct_scratch(const sv_conv_geom& g, int mode) { return conv_gemm_splitk_workspace(g, mode); }

template <typename T>
static void fwd_direct_any(const sv_conv_geom& g, const T* x, const T* w, const float* bias, T* y, int act, float alpha,
                           dim3 grid, int cot, hipStream_t st) {
    if (cot == 16)
        hipLaunchKernelGGL((conv2d_fwd_direct<16, T>), grid, dim3(256), 0, st, g, x, w, bias, y, act, alpha);
    else if (cot == 8)
        hipLaunchKernelGGL((conv2d_fwd_direct<8, T>), grid, dim3(256), 0, st, g, x, w, bias, y, act, alpha);
    else
        hipLaunchKernelGGL((conv2d_fwd_direct<4, T>), grid, dim3(256), 0, st, g, x, w, bias, y, act, alpha);
}
template <typename T>
static void dgrad_direct_any(const sv_conv_geom& g, const T* dz, const T* w, const float* bias, T* dx, int act,
                             float alpha, dim3 grid, int cit, hipStream_t st) {
    if (cit == 16)
        hipLaunchKernelGGL((conv2d_dgrad_direct<16, T>), grid, dim3(256), 0, st, g, dz, w, bias, dx, act, alpha);
    else if (cit == 8)
        hipLaunchKernelGGL((conv2d_dgrad_direct<8, T>), grid, dim3(256), 0, st, g, dz, w, bias, dx, act, alpha);
    else
        hipLaunchKernelGGL((conv2d_dgrad_direct<4, T>), grid, dim3(256), 0, st, g, dz, w, bias, dx, act, alpha);
}
template <typename T>
static void wgrad_direct_any(const sv_conv_geom& g, const T* x, const T* dz, float* part, int64_t pps, dim3 grid, int cot,
                             hipStream_t st) {
    if (cot == 16)
        hipLaunchKernelGGL((conv2d_wgrad_direct<16, T>), grid, dim3(256), 0, st, g, x, dz, part, pps);
    else if (cot == 8)
        hipLaunchKernelGGL((conv2d_wgrad_direct<8, T>), grid, dim3(256), 0, st, g, x, dz, part, pps);
    else
        hipLaunchKernelGGL((conv2d_wgrad_direct<4, T>), grid, dim3(256), 0, st, g, x, dz, part, pps);
}

// The masked forms (x read as x * (1 - mask)) exist on the implicit-GEMM kernel only.
bool conv_masked_supported(const sv_conv_geom& g) { return conv_gemm_worthwhile(g, 0) && conv_gemm_worthwhile(g, 2); }

int launch_fwd_masked(const sv_conv_geom& g, const float* x, const float* mask, const float* w, const float* bias, float* y,
                      int act, float alpha, hipStream_t st, void* ws, size_t ws_bytes) {
    if (!conv_masked_supported(g)) return fail(SV_E_UNSUPPORTED, "sv_conv2d_fwd_masked: geometry outside the implicit-GEMM family");
    return launch_conv_gemm_fwd(g, x, w, bias, y, act, alpha, ws, ws_bytes, st, mask);
}

int launch_fwd_direct(const sv_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act,
                      float alpha, hipStream_t st, void* ws, size_t ws_bytes) {
    if (conv_gemm_worthwhile(g, 0)) return launch_conv_gemm_fwd(g, x, w, bias, y, act, alpha, ws, ws_bytes, st);
    const int64_t npix = (int64_t)g.N * g.Ho * g.Wo;
    const int cot = g.Cout >= 16 ? 16 : (g.Cout > 4 ? 8 : 4);
    dim3 grid((unsigned)ew_grid(npix, 256, 1 << 16), (unsigned)ceil_div(g.Cout, cot));
    fwd_direct_any<float>(g, x, w, bias, y, act, alpha, grid, cot, st);
    return check_launch("conv2d_fwd_direct");
}

int launch_dgrad_direct(const sv_conv_geom& g, const float* dz, const float* w, const float* bias, float* dx,
                        int act, float alpha, hipStream_t st, void* ws, size_t ws_bytes) {
    {  // image side of <= 4 channels: LDS-tiled kernel (as a GEMM of <= 4 columns the gathers are re-read per pixel)
        const int rc = launch_convt_small(g, dz, w, bias, dx, act, alpha, st);
        if (rc <= 0) return rc;
    }
    if (conv_gemm_worthwhile(g, 1)) return launch_conv_gemm_dgrad(g, dz, w, bias, dx, act, alpha, ws, ws_bytes, st);
    const int64_t npix = (int64_t)g.N * (g.H / g.up_h) * (g.W / g.up_w);
    const int cit = g.Cin >= 16 ? 16 : (g.Cin > 4 ? 8 : 4);
    dim3 grid((unsigned)ew_grid(npix, 256, 1 << 16), (unsigned)ceil_div(g.Cin, cit));
    dgrad_direct_any<float>(g, dz, w, bias, dx, act, alpha, grid, cit, st);
    return check_launch("conv2d_dgrad_direct");
}

// Split of the pixel reduction: aim at >= 1024 blocks in total, at least 256 pixels per block.
int wgrad_direct_splits(const sv_conv_geom& g) {
    const int64_t npix = (int64_t)g.N * g.Ho * g.Wo;
    const int cot = g.Cout >= 16 ? 16 : (g.Cout > 4 ? 8 : 4);
    const int64_t base = (int64_t)g.KH * g.KW * g.Cin * ceil_div(g.Cout, cot);
    int64_t want = ceil_div(2048, base);
    int64_t maxs = ceil_div(npix, 512);
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    return (int)want;
}

size_t wgrad_direct_workspace(const sv_conv_geom& g) {
    const size_t direct = (size_t)wgrad_direct_splits(g) * g.KH * g.KW * g.Cin * g.Cout * sizeof(float);
    const size_t gemm = conv_gemm_wgrad_workspace(g);
    return direct > gemm ? direct : gemm;
}

int launch_wgrad_direct(const sv_conv_geom& g, const float* x, const float* dz, float* dw, int accumulate,
                        void* ws, size_t ws_bytes, hipStream_t st, const float* xmask) {
    const int64_t kc = (int64_t)g.KH * g.KW * g.Cin * g.Cout;
    if (ws == nullptr || ws_bytes < wgrad_direct_workspace(g))
        return fail(SV_E_WORKSPACE, "conv2d_wgrad: workspace %zu bytes < required %zu", ws_bytes,
                    wgrad_direct_workspace(g));
    if (xmask != nullptr && !conv_masked_supported(g))
        return fail(SV_E_UNSUPPORTED, "sv_conv2d_wgrad_masked: geometry outside the implicit-GEMM family");
    if (conv_gemm_worthwhile(g, 2)) {
        const int splits = conv_gemm_wgrad_splits(g);
        if (splits == 1) return launch_conv_gemm_wgrad(g, x, dz, dw, 1, st, accumulate, xmask);  // tiles go straight into dw
        int rc = launch_conv_gemm_wgrad(g, x, dz, (float*)ws, splits, st, 0, xmask);
        if (rc) return rc;
        if (defer_reduce((const float*)ws, dw, kc, splits, accumulate)) return SV_OK;
        hipLaunchKernelGGL(wgrad_reduce, dim3((unsigned)ceil_div(kc, 64)), dim3(1024), 0, st, (const float*)ws, dw, kc, splits,
                           accumulate);
        return check_launch("wgrad_reduce");
    }
    const int ns = wgrad_direct_splits(g);
    const int64_t npix = (int64_t)g.N * g.Ho * g.Wo;
    const int64_t pps = ceil_div(npix, ns);
    const int cot = g.Cout >= 16 ? 16 : (g.Cout > 4 ? 8 : 4);
    dim3 grid((unsigned)(g.KH * g.KW * g.Cin), (unsigned)ceil_div(g.Cout, cot), (unsigned)ns);
    float* part = (float*)ws;
    wgrad_direct_any<float>(g, x, dz, part, pps, grid, cot, st);
    int rc = check_launch("conv2d_wgrad_direct");
    if (rc) return rc;
    if (defer_reduce(part, dw, kc, ns, accumulate)) return SV_OK;
    hipLaunchKernelGGL(wgrad_reduce, dim3((unsigned)ceil_div(kc, 64)), dim3(1024), 0, st, part, dw, kc, ns, accumulate);
    return check_launch("wgrad_reduce");
}

// ---- bf16 storage (activations, filters, activation gradients as bf16; sums, bias and kernel gradients in f32):
// the matrix-core kernel of conv_gemm_bf16.hip where the shape is GEMM-like, the direct kernels above otherwise.
typedef __bf16 bf16_t;

int launch_fwd_bf16(const sv_conv_geom& g, const void* x, const void* w, const float* bias, void* y, int act, float alpha,
                    hipStream_t st, void* ws, size_t ws_bytes) {
    if (conv_gemm_bf16_ok(g, 0)) return launch_conv_gemm_bf16_fwd(g, x, w, bias, y, act, alpha, ws, ws_bytes, st);
    const int64_t npix = (int64_t)g.N * g.Ho * g.Wo;
    const int cot = g.Cout >= 16 ? 16 : (g.Cout > 4 ? 8 : 4);
    dim3 grid((unsigned)ew_grid(npix, 256, 1 << 16), (unsigned)ceil_div(g.Cout, cot));
    fwd_direct_any<bf16_t>(g, (const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, act, alpha, grid, cot, st);
    return check_launch("conv2d_fwd_direct<bf16>");
}

int launch_dgrad_bf16(const sv_conv_geom& g, const void* dz, const void* w, const float* bias, void* dx, int act,
                      float alpha, hipStream_t st, void* ws, size_t ws_bytes) {
    {
        const int rc = launch_convt_small_bf16(g, dz, w, bias, dx, act, alpha, st);
        if (rc <= 0) return rc;
    }
    if (conv_gemm_bf16_ok(g, 1))
        return launch_conv_gemm_bf16_dgrad(g, dz, w, bias, dx, act, alpha, ws, ws_bytes, st);
    const int64_t npix = (int64_t)g.N * (g.H / g.up_h) * (g.W / g.up_w);
    const int cit = g.Cin >= 16 ? 16 : (g.Cin > 4 ? 8 : 4);
    dim3 grid((unsigned)ew_grid(npix, 256, 1 << 16), (unsigned)ceil_div(g.Cin, cit));
    dgrad_direct_any<bf16_t>(g, (const bf16_t*)dz, (const bf16_t*)w, bias, (bf16_t*)dx, act, alpha, grid, cit, st);
    return check_launch("conv2d_dgrad_direct<bf16>");
}

int launch_wgrad_bf16(const sv_conv_geom& g, const void* x, const void* dz, float* dw, int accumulate, void* ws,
                      size_t ws_bytes, hipStream_t st) {
    const int64_t kc = (int64_t)g.KH * g.KW * g.Cin * g.Cout;
    if (ws == nullptr || ws_bytes < wgrad_direct_workspace(g))
        return fail(SV_E_WORKSPACE, "conv2d_wgrad: workspace %zu bytes < required %zu", ws_bytes,
                    wgrad_direct_workspace(g));
    if (conv_gemm_bf16_ok(g, 2)) {
        const int splits = conv_gemm_bf16_wgrad_splits(g);
        if (splits == 1) return launch_conv_gemm_bf16_wgrad(g, x, dz, dw, 1, st, accumulate);
        int rc = launch_conv_gemm_bf16_wgrad(g, x, dz, (float*)ws, splits, st, 0);
        if (rc) return rc;
        if (defer_reduce((const float*)ws, dw, kc, splits, accumulate)) return SV_OK;
        hipLaunchKernelGGL(wgrad_reduce, dim3((unsigned)ceil_div(kc, 64)), dim3(1024), 0, st, (const float*)ws, dw, kc, splits,
                           accumulate);
        return check_launch("wgrad_reduce");
    }
    const int ns = wgrad_direct_splits(g);
    const int64_t npix = (int64_t)g.N * g.Ho * g.Wo;
    const int64_t pps = ceil_div(npix, ns);
    const int cot = g.Cout >= 16 ? 16 : (g.Cout > 4 ? 8 : 4);
    dim3 grid((unsigned)(g.KH * g.KW * g.Cin), (unsigned)ceil_div(g.Cout, cot), (unsigned)ns);
    float* part = (float*)ws;
    wgrad_direct_any<bf16_t>(g, (const bf16_t*)x, (const bf16_t*)dz, part, pps, grid, cot, st);
    int rc = check_launch("conv2d_wgrad_direct<bf16>");
    if (rc) return rc;
    if (defer_reduce(part, dw, kc, ns, accumulate)) return SV_OK;
    hipLaunchKernelGGL(wgrad_reduce, dim3((unsigned)ceil_div(kc, 64)), dim3(1024), 0, st, part, dw, kc, ns, accumulate);
    return check_launch("wgrad_reduce");
}

}  // namespace sv
