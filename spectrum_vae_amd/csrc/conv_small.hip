// Input gradient of a strided conv / forward of a Conv2DTranspose whose image side has at most 4 channels: the output
// layers of the context encoder (context_encoder.py:296-304, Conv2DTranspose(3, 4x4, stride 2, same, tanh)) and of the
// U-Net (unet.py:104-112, Conv2DTranspose(3, 2x2, stride 2)).
//
// As an implicit GEMM these layers have 3 columns: every output pixel gathers its taps x channels row again (1 KB per
// pixel, 1 GB of L2 reads per launch at batch 1024 for 67 MB of input).  Here a workgroup stages a 16 x 16 tile of the
// channel-rich (small-extent) tensor with its halo into LDS once, 16 channels at a time; a thread owns one of its pixels
// and computes the S x S image pixels above it for all (<= 4) image channels.  The filter index depends on loop counters
// only, so the filters arrive through scalar loads (uniform across the wave) and cost no LDS or vector-memory traffic.
//
// Geometry follows sv_conv_geom of the MIRRORED conv: (H, W, Cin) is the image side (Cin <= 4), (Ho, Wo, Cout) the
// channel-rich side, filters [KH][KW][Cin][Cout].  Covered: KH = KW = K, SH = SW = S, K a multiple of S, H = S Ho,
// W = S Wo, pad_t = pad_l = (K - S) / 2, Cout a multiple of 4.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace sv {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kSmallCib = 16;           // channels staged per pass (small: several workgroups per CU hide the scalar loads)
constexpr int kSmallPitch = kSmallCib + 4;  // floats per staged pixel: 16-byte reads of 16 neighbouring pixels hit 64 banks

// Four consecutive filter values at a wave-uniform address, as two pairs.  bf16: read as two dwords (sub-dword reads
// would leave the scalar path) and widened with shifts / masks, which stay on the scalar unit for uniform values.
__device__ __forceinline__ void load_w4(const float* __restrict__ p, v2f& lo, v2f& hi) {
    lo = v2f{p[0], p[1]};
    hi = v2f{p[2], p[3]};
}
__device__ __forceinline__ void load_w4(const __bf16* __restrict__ p, v2f& lo, v2f& hi) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
    const uint32_t a = q[0], b = q[1];
    lo = v2f{__builtin_bit_cast(float, a << 16), __builtin_bit_cast(float, a & 0xffff0000u)};
    hi = v2f{__builtin_bit_cast(float, b << 16), __builtin_bit_cast(float, b & 0xffff0000u)};
}

// T: storage type of x, w and y (float or bf16); the staged tile, the sums and the bias are f32.
template <int K, int S, int CB, typename T>
__global__ __launch_bounds__(256) void convt_small_k(const T* __restrict__ x, const T* __restrict__ w,
                                                     const float* __restrict__ bias, T* __restrict__ y, int Hs, int Ws,
                                                     int Cs, int act, float alpha, int tiles_w, int tiles_per_img) {
    constexpr int VN = 16 / sizeof(T);  // channels per 16-byte piece of x
    constexpr int Q = K / S, PT = (K - S) / 2, HB = Q - 1, HA = (S - 1 + PT) / S, WIN = HB + 1 + HA, TP = 16 + HB + HA;
    extern __shared__ float4 smem4[];
    float* sm = reinterpret_cast<float*>(smem4);
    const int n = blockIdx.x / tiles_per_img, t = blockIdx.x - n * tiles_per_img;
    const int th0 = (t / tiles_w) * 16, tw0 = (t % tiles_w) * 16;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    // running sums as (even channels, odd channels) pairs: one packed FMA takes two consecutive channels of the staged
    // pixel and two consecutive filter values (an SGPR pair straight from the scalar load)
    v2f acc[S][S][CB];
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
        for (int c = 0; c < S; ++c)
#pragma unroll
            for (int co = 0; co < CB; ++co) acc[r][c][co] = v2f{0.f, 0.f};
    for (int c0 = 0; c0 < Cs; c0 += kSmallCib) {
        const int cq = (Cs - c0 < kSmallCib ? Cs - c0 : kSmallCib) >> 2;  // channel quads in this pass
        __syncthreads();  // the previous pass has been consumed
        for (int u = threadIdx.x; u < TP * TP * (kSmallCib / VN); u += 256) {
            const int p = u / (kSmallCib / VN), v = u - p * (kSmallCib / VN);
            const int ph = p / TP, pw = p - ph * TP;
            const int ih = th0 + ph - HB, iw = tw0 + pw - HB;
            struct alignas(16) Piece { T v[VN]; } val;
#pragma unroll
            for (int e = 0; e < VN; ++e) val.v[e] = (T)0.f;
            if (v * VN < 4 * cq && ih >= 0 && ih < Hs && iw >= 0 && iw < Ws)
                val = *reinterpret_cast<const Piece*>(x + (((int64_t)n * Hs + ih) * Ws + iw) * Cs + c0 + VN * v);
#pragma unroll
            for (int e4 = 0; e4 < VN / 4; ++e4)
                *reinterpret_cast<float4*>(sm + p * kSmallPitch + VN * v + 4 * e4) =
                    make_float4((float)val.v[4 * e4], (float)val.v[4 * e4 + 1], (float)val.v[4 * e4 + 2],
                                (float)val.v[4 * e4 + 3]);
        }
        __syncthreads();
        for (int v = 0; v < cq; ++v) {
            float4 xq[WIN][WIN];
#pragma unroll
            for (int dh = 0; dh < WIN; ++dh)
#pragma unroll
                for (int dw = 0; dw < WIN; ++dw)
                    xq[dh][dw] = *reinterpret_cast<const float4*>(sm + ((ty + dh) * TP + tx + dw) * kSmallPitch + 4 * v);
            const T* wv = w + c0 + 4 * v;  // wave-uniform: the filter reads below are scalar loads
#pragma unroll
            for (int r = 0; r < S; ++r)
#pragma unroll
                for (int jh = 0; jh < Q; ++jh)
#pragma unroll
                    for (int c = 0; c < S; ++c)
#pragma unroll
                        for (int jw = 0; jw < Q; ++jw) {
                            // image row S m + r takes tap a = (r + PT) % S + S jh from channel-rich row m + (r + PT) / S - jh
                            const int a = (r + PT) % S + S * jh, b = (c + PT) % S + S * jw;
                            const float4 xv = xq[HB + (r + PT) / S - jh][HB + (c + PT) / S - jw];
                            const v2f xlo{xv.x, xv.y}, xhi{xv.z, xv.w};
#pragma unroll
                            for (int co = 0; co < CB; ++co) {
                                v2f wlo, whi;
                                load_w4(wv + (int64_t)((a * K + b) * CB + co) * Cs, wlo, whi);
                                acc[r][c][co] = __builtin_elementwise_fma(xlo, wlo, acc[r][c][co]);
                                acc[r][c][co] = __builtin_elementwise_fma(xhi, whi, acc[r][c][co]);
                            }
                        }
        }
    }
    const int m = th0 + ty, q = tw0 + tx;
    if (m >= Hs || q >= Ws) return;
    const int Hb = Hs * S, Wb = Ws * S;
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
        for (int c = 0; c < S; ++c) {
            T* o = y + (((int64_t)n * Hb + S * m + r) * Wb + S * q + c) * CB;
#pragma unroll
            for (int co = 0; co < CB; ++co) {
                const float z = (acc[r][c][co].x + acc[r][c][co].y) + (bias != nullptr ? bias[co] : 0.f);
                o[co] = (T)act_apply(z, act, alpha);
            }
        }
}


// Small launches (fewer 16 x 16 tiles than CUs: the GAN drivers' batch sizes): a thread's chain of scalar filter loads and
// dependent FMAs is then the whole duration, with nothing else on the CU to hide it.  This form cuts the chain in four:
// tiles of 8 x 8 pixels (4 x the workgroups), one wave per tile position set, wave v taking channel quad v of every staged
// pass; the four partial sums of an output meet in LDS.
template <int K, int S, int CB, typename T>
__global__ __launch_bounds__(256) void convt_small_split_k(const T* __restrict__ x, const T* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ y, int Hs, int Ws,
                                                           int Cs, int act, float alpha, int tiles_w, int tiles_per_img) {
    constexpr int VN = 16 / sizeof(T);
    constexpr int Q = K / S, PT = (K - S) / 2, HB = Q - 1, HA = (S - 1 + PT) / S, WIN = HB + 1 + HA, TP = 8 + HB + HA;
    constexpr int NOUT = S * S * CB;  // results per pixel of the tile
    extern __shared__ float4 smem4[];
    float* sm = reinterpret_cast<float*>(smem4);
    const int n = blockIdx.x / tiles_per_img, t = blockIdx.x - n * tiles_per_img;
    const int th0 = (t / tiles_w) * 8, tw0 = (t % tiles_w) * 8;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ty = lane >> 3, tx = lane & 7;
    v2f acc[S][S][CB];
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
        for (int c = 0; c < S; ++c)
#pragma unroll
            for (int co = 0; co < CB; ++co) acc[r][c][co] = v2f{0.f, 0.f};
    for (int c0 = 0; c0 < Cs; c0 += kSmallCib) {
        const int cq = (Cs - c0 < kSmallCib ? Cs - c0 : kSmallCib) >> 2;
        __syncthreads();
        for (int u = threadIdx.x; u < TP * TP * (kSmallCib / VN); u += 256) {
            const int p = u / (kSmallCib / VN), v = u - p * (kSmallCib / VN);
            const int ph = p / TP, pw = p - ph * TP;
            const int ih = th0 + ph - HB, iw = tw0 + pw - HB;
            struct alignas(16) Piece { T v[VN]; } val;
#pragma unroll
            for (int e = 0; e < VN; ++e) val.v[e] = (T)0.f;
            if (v * VN < 4 * cq && ih >= 0 && ih < Hs && iw >= 0 && iw < Ws)
                val = *reinterpret_cast<const Piece*>(x + (((int64_t)n * Hs + ih) * Ws + iw) * Cs + c0 + VN * v);
#pragma unroll
            for (int e4 = 0; e4 < VN / 4; ++e4)
                *reinterpret_cast<float4*>(sm + p * kSmallPitch + VN * v + 4 * e4) =
                    make_float4((float)val.v[4 * e4], (float)val.v[4 * e4 + 1], (float)val.v[4 * e4 + 2],
                                (float)val.v[4 * e4 + 3]);
        }
        __syncthreads();
        const int v = __builtin_amdgcn_readfirstlane(wv);  // wave-uniform by construction: keeps the filter loads scalar
        if (v < cq) {
            float4 xq[WIN][WIN];
#pragma unroll
            for (int dh = 0; dh < WIN; ++dh)
#pragma unroll
                for (int dw = 0; dw < WIN; ++dw)
                    xq[dh][dw] = *reinterpret_cast<const float4*>(sm + ((ty + dh) * TP + tx + dw) * kSmallPitch + 4 * v);
            const T* wvp = w + c0 + 4 * v;
#pragma unroll
            for (int r = 0; r < S; ++r)
#pragma unroll
                for (int jh = 0; jh < Q; ++jh)
#pragma unroll
                    for (int c = 0; c < S; ++c)
#pragma unroll
                        for (int jw = 0; jw < Q; ++jw) {
                            const int a = (r + PT) % S + S * jh, b = (c + PT) % S + S * jw;
                            const float4 xv = xq[HB + (r + PT) / S - jh][HB + (c + PT) / S - jw];
                            const v2f xlo{xv.x, xv.y}, xhi{xv.z, xv.w};
#pragma unroll
                            for (int co = 0; co < CB; ++co) {
                                v2f wlo, whi;
                                load_w4(wvp + (int64_t)((a * K + b) * CB + co) * Cs, wlo, whi);
                                acc[r][c][co] = __builtin_elementwise_fma(xlo, wlo, acc[r][c][co]);
                                acc[r][c][co] = __builtin_elementwise_fma(xhi, whi, acc[r][c][co]);
                            }
                        }
        }
    }
    __syncthreads();  // the last pass has been consumed: the staging area becomes red[wave][result][lane]
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
        for (int c = 0; c < S; ++c)
#pragma unroll
            for (int co = 0; co < CB; ++co)
                sm[(wv * NOUT + (r * S + c) * CB + co) * 64 + lane] = acc[r][c][co].x + acc[r][c][co].y;
    __syncthreads();
    const int Hb = Hs * S, Wb = Ws * S;
    for (int idx = threadIdx.x; idx < 64 * NOUT; idx += 256) {
        const int o = idx >> 6, l = idx & 63;
        const int m = th0 + (l >> 3), q = tw0 + (l & 7);
        if (m >= Hs || q >= Ws) continue;
        const int co = o % CB, rc = o / CB, r = rc / S, c = rc - r * S;
        const float z = ((sm[o * 64 + l] + sm[(NOUT + o) * 64 + l]) + (sm[(2 * NOUT + o) * 64 + l] + sm[(3 * NOUT + o) * 64 + l])) +
                        (bias != nullptr ? bias[co] : 0.f);
        y[(((int64_t)n * Hb + S * m + r) * Wb + S * q + c) * CB + co] = (T)act_apply(z, act, alpha);
    }
}

// ---- bf16 storage on the matrix cores.  Over the STORED pixels of the channel-rich tensor the layer is one GEMM:
//   D[pixel][(phase r, c; image channel co)] = sum over the WIN x WIN window around the pixel and ch of
//       x[pixel + (dh, dw) - HB][ch] * Wp[(dh, dw)][ch][(r, c, co)],
// where Wp holds the tap that connects window position (dh, dw) with output phase (r, c) (zero where there is none: a 4x4
// stride-2 layer reads a 3 x 3 window of which each phase uses 2 x 2).  16 pixels x 16 columns x 16 channels per
// v_mfma_f32_16x16x16_bf16; S * S * CB <= 16 columns.  A workgroup builds Wp in LDS once (16-channel pieces copied from
// the filter tensor) and walks 8 x 8 pixel tiles: the tile and its halo are staged as bf16 [pixel][channel] with a
// 16-byte pad per pixel (the 8-byte operand reads of 16 pixels then cover all banks), wave v owns rows 2v, 2v + 1.
// The layer is bound by reading x once instead of by the vector pipe's scalar filter loads.
typedef short sm_v4s __attribute__((ext_vector_type(4)));
typedef float sm_v4f __attribute__((ext_vector_type(4)));

template <int K, int S, typename T>
__global__ __launch_bounds__(256) void convt_small_mfma_k(const T* __restrict__ x, const T* __restrict__ w,
                                                          const float* __restrict__ bias, T* __restrict__ y, int Hs,
                                                          int Ws, int Cs, int CB, int act, float alpha, int tiles_w,
                                                          int tiles_per_img, int n_tiles) {
    constexpr int Q = K / S, PT = (K - S) / 2, HB = Q - 1, HA = (S - 1 + PT) / S, WIN = HB + 1 + HA, TP = 8 + HB + HA;
    constexpr bool LP = sizeof(T) == 2;
    constexpr int EB = sizeof(T);            // bytes per element
    constexpr int KS = LP ? 16 : 4;          // channels per matrix instruction (v_mfma_f32_16x16x16_bf16 / _16x16x4_f32)
    constexpr int BLK = 16 * KS * EB;        // bytes of the filter image per (window position, channel step)
    constexpr int LB = KS / 4 * EB;          // bytes a lane reads per operand: its KS / 4 consecutive channels
    extern __shared__ float4 smem4[];
    unsigned char* sm = reinterpret_cast<unsigned char*>(smem4);
    const int KK = Cs / KS;                  // channel steps
    const int pitch = Cs * EB + 16;          // bytes per staged pixel
    unsigned char* wp = sm;                  // [WIN * WIN][KK][16 columns][KS channels]
    unsigned char* xt = sm + WIN * WIN * KK * BLK;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int ncol = S * S * CB;
    for (int u = threadIdx.x; u < WIN * WIN * KK * 16; u += 256) {
        const int col = u & 15, kk = (u >> 4) % KK, pos = (u >> 4) / KK;
        const int dh = pos / WIN, dw = pos - dh * WIN;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (col < ncol) {
            const int phase = col / CB, co = col - phase * CB;
            const int r = phase / S, c = phase - r * S;
            const int jh = HB + (r + PT) / S - dh, jw = HB + (c + PT) / S - dw;
            if (jh >= 0 && jh < Q && jw >= 0 && jw < Q) {
                const int a = (r + PT) % S + S * jh, b = (c + PT) % S + S * jw;
                const float4* src = reinterpret_cast<const float4*>(w + (int64_t)((a * K + b) * CB + co) * Cs + KS * kk);
                v0 = src[0];
                if (LP) v1 = src[1];
            }
        }
        float4* dst = reinterpret_cast<float4*>(wp + (size_t)u * (KS * EB));
        dst[0] = v0;
        if (LP) dst[1] = v1;
    }
    const int prow = 2 * wv + (li >> 3), pcol = li & 7;
    const int abase = (prow * TP + pcol) * pitch + lk * LB;
    const int Hb = Hs * S, Wb = Ws * S;
    const int pieces = Cs * EB >> 4;  // 16-byte pieces per pixel
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int n = tile / tiles_per_img, t = tile - n * tiles_per_img;
        const int th0 = (t / tiles_w) * 8, tw0 = (t % tiles_w) * 8;
        __syncthreads();  // the previous tile has been consumed (first turn: nothing to wait for but Wp's writers)
        for (int u = threadIdx.x; u < TP * TP * pieces; u += 256) {
            const int p = u / pieces, v = u - p * pieces;
            const int ph = p / TP, pw = p - ph * TP;
            const int ih = th0 + ph - HB, iw = tw0 + pw - HB;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ih >= 0 && ih < Hs && iw >= 0 && iw < Ws)
                val = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(x) +
                                                       ((((int64_t)n * Hs + ih) * Ws + iw) * Cs * EB + 16 * v));
            *reinterpret_cast<float4*>(xt + p * pitch + v * 16) = val;
        }
        __syncthreads();
        sm_v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pos = 0; pos < WIN * WIN; ++pos) {
            const unsigned char* ap = xt + abase + ((pos / WIN) * TP + (pos % WIN)) * pitch;
            const unsigned char* bp = wp + ((size_t)pos * KK * 16 + li) * (KS * EB) + lk * LB;
            for (int kk = 0; kk < KK; ++kk) {
                if constexpr (LP) {
                    const sm_v4s av = *reinterpret_cast<const sm_v4s*>(ap + kk * (KS * EB));
                    const sm_v4s bv = *reinterpret_cast<const sm_v4s*>(bp + kk * BLK);
                    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, acc, 0, 0, 0);
                } else {
                    const float av = *reinterpret_cast<const float*>(ap + kk * (KS * EB));
                    const float bv = *reinterpret_cast<const float*>(bp + kk * BLK);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                }
            }
        }
        // the lane holds column li = (phase, co) of pixels 4 lk .. 4 lk + 3 of the wave's 16
        if (li < ncol) {
            const int phase = li / CB, co = li - phase * CB;
            const int r = phase / S, c = phase - r * S;
            const float bz = bias != nullptr ? bias[co] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = 4 * lk + j;
                const int m = th0 + 2 * wv + (p >> 3), q = tw0 + (p & 7);
                if (m < Hs && q < Ws)
                    y[(((int64_t)n * Hb + S * m + r) * Wb + S * q + c) * CB + co] = (T)act_apply(acc[j] + bz, act, alpha);
            }
        }
    }
}

// 1: this form does not take the layer (the filter image and a tile do not fit 64 KB of LDS)
template <int K, int S, typename T>
static int launch_small_mfma(const sv_conv_geom& g, const T* x, const T* w, const float* bias, T* y, int act, float alpha,
                             hipStream_t st) {
    constexpr int Q = K / S, PT = (K - S) / 2, WIN = Q + (S - 1 + PT) / S, TP = 8 + WIN - 1;
    const int th8 = (int)ceil_div(g.Ho, 8), tw8 = (int)ceil_div(g.Wo, 8);
    const int64_t tiles = (int64_t)g.N * th8 * tw8;
    if (tiles > (1ll << 30)) return 1;
    const size_t lds = (size_t)WIN * WIN * g.Cout * 16 * sizeof(T) + (size_t)TP * TP * (g.Cout * sizeof(T) + 16);
    if (lds > 64 * 1024) return 1;
    const int grid = tiles < 2048 ? (int)tiles : 2048;
    hipLaunchKernelGGL((convt_small_mfma_k<K, S, T>), dim3((unsigned)grid), dim3(256), lds, st, x, w, bias, y, g.Ho, g.Wo,
                       g.Cout, g.Cin, act, alpha, tw8, th8 * tw8, (int)tiles);
    return check_launch("convt_small_mfma");
}

template <int K, int S, typename T>
static int launch_small_ks(const sv_conv_geom& g, const T* x, const T* w, const float* bias, T* y, int act, float alpha,
                           hipStream_t st) {
    constexpr int Q = K / S, PT = (K - S) / 2, TP = 16 + (Q - 1) + (S - 1 + PT) / S;
    const int tiles_h = (int)ceil_div(g.Ho, 16), tiles_w = (int)ceil_div(g.Wo, 16);
    const int64_t blocks = (int64_t)g.N * tiles_h * tiles_w;
    if (blocks > (1ll << 30)) return 1;
    if (blocks < 256) {  // about one workgroup per CU or less: the split form
        constexpr int TPS = 8 + (Q - 1) + (S - 1 + PT) / S;
        const int th8 = (int)ceil_div(g.Ho, 8), tw8 = (int)ceil_div(g.Wo, 8);
        const size_t stage = (size_t)TPS * TPS * kSmallPitch * sizeof(float);
        const dim3 grid8((unsigned)((int64_t)g.N * th8 * tw8)), block8(256);
#define SV_SMALL_SPLIT(CBV)                                                                                             \
    do {                                                                                                                \
        const size_t red = (size_t)4 * S * S * CBV * 64 * sizeof(float);                                                \
        hipLaunchKernelGGL((convt_small_split_k<K, S, CBV, T>), grid8, block8, stage > red ? stage : red, st, x, w, bias, \
                           y, g.Ho, g.Wo, g.Cout, act, alpha, tw8, th8 * tw8);                                          \
    } while (0)
        switch (g.Cin) {
            case 1: SV_SMALL_SPLIT(1); break;
            case 2: SV_SMALL_SPLIT(2); break;
            case 3: SV_SMALL_SPLIT(3); break;
            default: SV_SMALL_SPLIT(4); break;
        }
#undef SV_SMALL_SPLIT
        return check_launch("convt_small_split");
    }
    const size_t lds = (size_t)TP * TP * kSmallPitch * sizeof(float);
    const dim3 grid((unsigned)blocks), block(256);
    switch (g.Cin) {
        case 1: hipLaunchKernelGGL((convt_small_k<K, S, 1, T>), grid, block, lds, st, x, w, bias, y, g.Ho, g.Wo, g.Cout, act, alpha, tiles_w, tiles_h * tiles_w); break;
        case 2: hipLaunchKernelGGL((convt_small_k<K, S, 2, T>), grid, block, lds, st, x, w, bias, y, g.Ho, g.Wo, g.Cout, act, alpha, tiles_w, tiles_h * tiles_w); break;
        case 3: hipLaunchKernelGGL((convt_small_k<K, S, 3, T>), grid, block, lds, st, x, w, bias, y, g.Ho, g.Wo, g.Cout, act, alpha, tiles_w, tiles_h * tiles_w); break;
        default: hipLaunchKernelGGL((convt_small_k<K, S, 4, T>), grid, block, lds, st, x, w, bias, y, g.Ho, g.Wo, g.Cout, act, alpha, tiles_w, tiles_h * tiles_w); break;
    }
    return check_launch("convt_small");
}

bool convt_small_ok(const sv_conv_geom& g, int vn) {
    if (g.Cin < 1 || g.Cin > 4 || g.Cout < 16 || (g.Cout % vn)) return false;
    if (g.KH != g.KW || g.SH != g.SW || g.up_h != 1 || g.up_w != 1) return false;
    if (!((g.KH == 4 && g.SH == 2) || (g.KH == 2 && g.SH == 2) || (g.KH == 4 && g.SH == 1))) return false;
    if (g.H != g.Ho * g.SH || g.W != g.Wo * g.SW) return false;
    const int pt = (g.KH - g.SH) / 2;
    return g.pad_t == pt && g.pad_l == pt;
}

template <typename T>
static int launch_small_any(const sv_conv_geom& g, const T* x, const T* w, const float* bias, T* y, int act, float alpha,
                            hipStream_t st) {
    if (!convt_small_ok(g, 16 / (int)sizeof(T))) return 1;
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return 1;
    if (opt(OPT_SMALL_CONVT) == 0) return 1;  // the general (implicit GEMM) path, for A/B measurements
    // The matrix-core form (option value 2 keeps the vector form, for A/B runs).  bf16 storage: whole 16-channel steps,
    // any batch.  f32 storage only on request (option value 3): the f32 matrix instruction has the vector pipe's peak and
    // this form multiplies the zero taps too - measured equal below 256 tiles (both at the launch floor), 1.3-1.5 x
    // faster around 1 000 tiles, 1.15-1.85 x slower at 4 000.
    if (opt(OPT_SMALL_CONVT) != 2 && g.SH == 2 && (g.Cout % (sizeof(T) == 2 ? 16 : 4)) == 0 &&
        (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        if (sizeof(T) == 2 || opt(OPT_SMALL_CONVT) == 3) {
            const int rc = g.KH == 4 ? launch_small_mfma<4, 2, T>(g, x, w, bias, y, act, alpha, st)
                                     : launch_small_mfma<2, 2, T>(g, x, w, bias, y, act, alpha, st);
            if (rc <= 0) return rc;
        }
    }
    if (g.KH == 4 && g.SH == 1) return launch_small_ks<4, 1, T>(g, x, w, bias, y, act, alpha, st);
    if (g.KH == 4) return launch_small_ks<4, 2, T>(g, x, w, bias, y, act, alpha, st);
    return launch_small_ks<2, 2, T>(g, x, w, bias, y, act, alpha, st);
}

// Return 1 when the geometry is not covered (the caller takes its general path), 0 on success, < 0 on error.
// x: [N][Ho][Wo][Cout] (the channel-rich tensor), y: [N][H][W][Cin].
int launch_convt_small(const sv_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act,
                       float alpha, hipStream_t st) {
    return launch_small_any<float>(g, x, w, bias, y, act, alpha, st);
}

int launch_convt_small_bf16(const sv_conv_geom& g, const void* x, const void* w, const float* bias, void* y, int act,
                            float alpha, hipStream_t st) {
    return launch_small_any<__bf16>(g, (const __bf16*)x, (const __bf16*)w, bias, (__bf16*)y, act, alpha, st);
}

}  // namespace sv
