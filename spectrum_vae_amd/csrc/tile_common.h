// Device-side building blocks shared by the LDS-tiled conv kernels (conv_tiled.hip) and the fused network kernels
// (net_fused.hip): planar LDS images with a zero halo, their staging from NHWC global tensors, and the register-tile
// FMA loop.  gfx950 only.
#pragma once
#include "common.h"

namespace sv {

enum { EPI_STORE = 0, EPI_MAXPOOL = 1, EPI_SUM4 = 2 };

// n / d for a divisor that is usually a power of two (image extents, channel counts).
__device__ __forceinline__ int dv(int n, int d) { return (d & (d - 1)) == 0 ? (n >> (31 - __clz(d))) : n / d; }
// Division of small non-negative ints by a loop-invariant divisor as one multiply: q = hi32(n * ceil(2^32 / d)), exact
// while n * d < 2^32 (tensor indices here stay far below that).  fd_make costs one real division; hoist it.
struct FDiv {
    uint32_t m;
    int d;
};
__host__ __device__ __forceinline__ FDiv fd_make(int d) { return FDiv{0xFFFFFFFFu / (uint32_t)d + 1u, d}; }
__device__ __forceinline__ int fd(int n, const FDiv& f) { return f.d == 1 ? n : (int)__umulhi((uint32_t)n, f.m); }
inline int ilog2_exact(int v) { return (v > 0 && (v & (v - 1)) == 0) ? __builtin_ctz(v) : -1; }

__host__ __device__ constexpr int floordiv_c(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// stored-patch row used by logical patch row q (q = pixel offset + tap) for pad PADL and upsampling UP
template <int UP, int PADL>
__host__ __device__ constexpr int patch_row(int q) { return floordiv_c(q - PADL, UP) - floordiv_c(-PADL, UP); }

// ---- staging: NHWC tensor in global memory -> planar images in LDS ----------------------------------------------
// A global round trip costs far more than the few instructions that consume it, so staging is split into a LOAD
// phase that issues up to U independent 16-byte loads per thread into registers and a STORE phase that decodes
// positions and writes LDS; callers put other work (LDS zero fill, the previous pass's arithmetic) between the two.
struct PlanarSrc {
    const float* x;       // [n][Hs][Ws][C]
    const float* zy;      // mode 1/2: activation output the gradient is taken through
    const uint8_t* zidx;  // mode 2: argmax slot (0..3) of each 2x2 window
    // 0: x as is;  1: x * act'(zy);  2: x, zy, zidx live at half resolution and the value lands on the argmax slot of
    // its 2x2 window of the LDS image (zeros on the other three);  3: x replicated to a 2x2 block (nearest upsampling)
    int mode;
    ActGradCoef zk;
    int Hs, Ws, C;        // extents of the STORED tensor (mode 2: the pooled extents) and its channel count
};
struct PlanarDst {
    int cpl;    // channel planes per sample in LDS
    int plane;  // floats per plane
    int wp;     // floats per row
    int off;    // row/col offset of pixel (0,0) inside a plane (the zero halo)
};
struct StageJob {
    int kind;   // 0: float4 = 4 channels of one pixel; 1: float4 = 4 consecutive floats of the contiguous pass; 2: scalars
    int units;  // float4s (kind 0/1) or floats (kind 2) in this pass
    int c0, cc, C4;
    size_t base;  // element offset of sample n0
};
__device__ __forceinline__ StageJob make_stage_job(const PlanarSrc& s, int n0, int ns, int c0, int cc) {
    StageJob j;
    const int HW = s.Hs * s.Ws;
    j.c0 = c0; j.cc = cc; j.C4 = cc >> 2;
    j.base = (size_t)n0 * HW * s.C;
    if ((cc & 3) == 0 && (s.C & 3) == 0 && (c0 & 3) == 0) {
        j.kind = 0; j.units = ns * HW * j.C4;
    } else if (c0 == 0 && cc == s.C && ((HW * s.C) & 3) == 0) {
        j.kind = 1; j.units = (ns * HW * s.C) >> 2;
    } else {
        j.kind = 2; j.units = ns * HW * cc;
    }
    return j;
}

// one staged value -> its place(s) in the LDS image of channel plane q0 (already offset to the sample and channel)
template <int MODE>
__device__ __forceinline__ void stage_put(const PlanarSrc& s, const PlanarDst& d, float* q0, int row, int col, float x,
                                          float y, uint32_t k) {
    if (MODE == 0) {
        q0[(row + d.off) * d.wp + col + d.off] = x;
    } else if (MODE == 1) {
        q0[(row + d.off) * d.wp + col + d.off] = x * act_grad_q(y, s.zk);
    } else if (MODE == 3) {
        float* q = q0 + (2 * row + d.off) * d.wp + 2 * col + d.off;
        q[0] = x; q[1] = x; q[d.wp] = x; q[d.wp + 1] = x;
    } else {
        const float v = x * act_grad_q(y, s.zk);
        float* q = q0 + (2 * row + d.off) * d.wp + 2 * col + d.off;
        q[0] = k == 0 ? v : 0.f; q[1] = k == 1 ? v : 0.f; q[d.wp] = k == 2 ? v : 0.f; q[d.wp + 1] = k == 3 ? v : 0.f;
    }
}

// One pass: every thread walks its units with a stride of nt.  The loop bodies are branch-free and unrolled by U so
// that the global loads of U units are issued before the first LDS write (a global round trip costs far more
// than the instructions that consume it).
template <int MODE, int U = 4>
__device__ __forceinline__ void stage_pass_m(const PlanarSrc& s, const PlanarDst& d, const StageJob& j, float* lds, int nt) {
    const int HW = s.Hs * s.Ws;
    const FDiv fHW = fd_make(HW), fW = fd_make(s.Ws);
    constexpr bool need_y = MODE == 1 || MODE == 2;
    // U units per batch: all their loads are issued (index clamped, so unconditional) before any use
    if (j.kind == 0) {
        const FDiv fC4 = fd_make(j.C4);
        for (int u0 = threadIdx.x; u0 < j.units; u0 += U * nt) {
            float4 v[U], y[U];
            uint32_t k[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int u = min(u0 + i * nt, j.units - 1);
                const int p = fd(u, fC4), cl = u - p * j.C4;
                const size_t gi = j.base + (size_t)p * s.C + j.c0 + 4 * cl;
                v[i] = *reinterpret_cast<const float4*>(s.x + gi);
                y[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                k[i] = 0;
                if (need_y) y[i] = *reinterpret_cast<const float4*>(s.zy + gi);
                if (MODE == 2) k[i] = *reinterpret_cast<const uint32_t*>(s.zidx + gi);
            }
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int u = u0 + i * nt;
                if (u < j.units) {
                    const int p = fd(u, fC4), cl = u - p * j.C4;
                    const int ss = fd(p, fHW), pp = p - ss * HW;
                    const int row = fd(pp, fW), col = pp - row * s.Ws;
                    float* q = lds + (size_t)(ss * d.cpl + 4 * cl) * d.plane;
                    stage_put<MODE>(s, d, q, row, col, v[i].x, y[i].x, k[i] & 0xff);
                    stage_put<MODE>(s, d, q + d.plane, row, col, v[i].y, y[i].y, (k[i] >> 8) & 0xff);
                    stage_put<MODE>(s, d, q + 2 * d.plane, row, col, v[i].z, y[i].z, (k[i] >> 16) & 0xff);
                    stage_put<MODE>(s, d, q + 3 * d.plane, row, col, v[i].w, y[i].w, k[i] >> 24);
                }
            }
        }
    } else if (j.kind == 1) {
        const FDiv fC = fd_make(s.C);
        for (int u0 = threadIdx.x; u0 < j.units; u0 += U * nt) {
            float4 v[U], y[U];
            uint32_t k[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const size_t gi = j.base + 4 * (size_t)min(u0 + i * nt, j.units - 1);
                v[i] = *reinterpret_cast<const float4*>(s.x + gi);
                y[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                k[i] = 0;
                if (need_y) y[i] = *reinterpret_cast<const float4*>(s.zy + gi);
                if (MODE == 2) k[i] = *reinterpret_cast<const uint32_t*>(s.zidx + gi);
            }
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int u = u0 + i * nt;
                if (u < j.units) {
                    const float xv[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
                    const float yv[4] = {y[i].x, y[i].y, y[i].z, y[i].w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int e = 4 * u + t;
                        const int p = fd(e, fC), c = e - p * s.C;
                        const int ss = fd(p, fHW), pp = p - ss * HW;
                        const int row = fd(pp, fW), col = pp - row * s.Ws;
                        stage_put<MODE>(s, d, lds + (size_t)(ss * d.cpl + c) * d.plane, row, col, xv[t], yv[t],
                                        (k[i] >> (8 * t)) & 0xff);
                    }
                }
            }
        }
    } else {
        const FDiv fcc = fd_make(j.cc);
#pragma unroll 4
        for (int u = threadIdx.x; u < j.units; u += nt) {
            const int p = fd(u, fcc), c = u - p * j.cc;
            const size_t gi = j.base + (size_t)p * s.C + j.c0 + c;
            const float v = s.x[gi];
            const float y = need_y ? s.zy[gi] : 0.f;
            const uint32_t k = MODE == 2 ? s.zidx[gi] : 0u;
            const int ss = fd(p, fHW), pp = p - ss * HW;
            const int row = fd(pp, fW), col = pp - row * s.Ws;
            stage_put<MODE>(s, d, lds + (size_t)(ss * d.cpl + c) * d.plane, row, col, v, y, k);
        }
    }
}

__device__ __forceinline__ void stage_pass(const PlanarSrc& s, const PlanarDst& d, const StageJob& j, float* lds, int nt) {
    switch (s.mode) {
        case 0: stage_pass_m<0>(s, d, j, lds, nt); break;
        case 1: stage_pass_m<1>(s, d, j, lds, nt); break;
        case 2: stage_pass_m<2>(s, d, j, lds, nt); break;
        default: stage_pass_m<3>(s, d, j, lds, nt); break;
    }
}

// LDS filter layout: [channel tile][channel (wcs per tile)][tap][CO_T]
template <int K, int CO_T>
__device__ __forceinline__ int w_lds_index(int co, int c, int tap, int wcs) {
    return (((co / CO_T) * wcs + c) * K * K + tap) * CO_T + (co % CO_T);
}

// ---- the register-tile FMA loop ---------------------------------------------------------------------------------
// A thread owns 2x2 output pixels x CO_T output channels.  Per input channel it loads the (K+1)x(K+1) patch (UP == 1)
// or the smaller stored patch (UP == 2: nearest upsampling folded into the indexing) once and reuses it for all K*K
// taps and CO_T channels; filter taps come from LDS at a wave-uniform address (broadcast reads).
//   img    : planar image of this thread's sample, [channel][plane]
//   wtile  : filters of this thread's channel tile, laid out [channel][tap][CO_T] (compile-time strides, so every
//            filter read is one base register + an immediate offset)
//   r0, c0 : patch origin inside a plane
//   channels c_begin, c_begin + c_step, ... < c_end are summed
template <int K, int UP, int CO_T>
__device__ __forceinline__ void conv_tile_fma(float (&acc)[2][2][CO_T], const float* img, const float* wtile, int plane,
                                              int wp, int r0, int c0, int c_begin, int c_end, int c_step) {
    constexpr int PADL = (K - 1) / 2;
    constexpr int NPR = patch_row<UP, PADL>(K) + 1;  // patch rows == patch cols
    for (int c = c_begin; c < c_end; c += c_step) {
        float patch[NPR][NPR + 1];
        const float* pl = img + c * plane + r0 * wp + c0;
#pragma unroll
        for (int pr = 0; pr < NPR; ++pr) {
            if (UP == 1) {  // NPR = K + 1 is even and c0 is even: 8-byte aligned pairs
#pragma unroll
                for (int pc = 0; pc < NPR; pc += 2) {
                    const float2 v2 = *reinterpret_cast<const float2*>(pl + pr * wp + pc);
                    patch[pr][pc] = v2.x;
                    patch[pr][pc + 1] = v2.y;
                }
            } else {
#pragma unroll
                for (int pc = 0; pc < NPR; ++pc) patch[pr][pc] = pl[pr * wp + pc];
            }
        }
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                float wv[CO_T];
                const float* wq = wtile + (c * K * K + kh * K + kw) * CO_T;
#pragma unroll
                for (int q = 0; q < CO_T; q += 4) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wq + q);
                    wv[q] = w4.x; wv[q + 1] = w4.y; wv[q + 2] = w4.z; wv[q + 3] = w4.w;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float v = patch[patch_row<UP, PADL>(i + kh)][patch_row<UP, PADL>(j + kw)];
#pragma unroll
                        for (int q = 0; q < CO_T; ++q) acc[i][j][q] = fmaf(v, wv[q], acc[i][j][q]);
                    }
            }
    }
}

}  // namespace sv
