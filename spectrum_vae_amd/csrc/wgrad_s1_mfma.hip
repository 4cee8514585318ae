// Kernel (+ bias) gradients of stride-1 SAME 3x3 / 5x5 convs with up to 128 channels on small images, on the f32 matrix
// cores (the gradient tape of run_model.py:281-289 for the layer shapes the hyper-parameter grid of run_model.py:164-219
// gives CAEPadded, autoencoder.py:437-526, outside its few-channel default stack):
//
//   dw[kh][kw][ci][co] = sum over samples and pixels p of x[p + (kh - pad, kw - pad)][ci] * dz[p][co],   db[co] = sum dz[p][co]
//
// The reduction runs over the pixels: one v_mfma_f32_16x16x4_f32 takes 16 input channels x 4 consecutive pixels of a row
// (A, read at the tap's shift from a channel-last LDS image of x with a zero halo) and 4 pixels x 16 output channels (B,
// from the image of dz).  A workgroup owns a SLICE of the filters - 16 or 32 input channels x 16 or 32 output channels,
// all taps - and a share of the samples: it stages only those channels of x and dz, its 4 waves split the pixel quads of
// a pass and keep all K x K x (slice tiles) accumulators in registers (25 tiles for 5x5 with 16 x 16 slices, 36 for 3x3
// with 32 x 32), the waves are summed as (w0 + w1) + (w2 + w3) through LDS and the slice lands in the workgroup's slab.
// The slabs of the sample shares are added in a fixed order by the library's slab-sum kernel (bit-reproducible, no float
// atomics).  An upsampled x (UpSampling2D folded into the conv) is read through (row >> 1, col >> 1) while it is staged.
// Operand reads are single 4-byte LDS reads at a per-lane base plus a per-tap offset; the pixel stride of a 32-channel
// slice is 48 floats, so that the two pixel lanes of a half-wave fall on different banks.
#include <type_traits>

#include "common.h"
#include "tile_common.h"

namespace sv {

typedef float ws_v4f __attribute__((ext_vector_type(4)));

struct WsArgs {
    const float* x;    // [N][H / up][W / up][Cin]
    const float* dz;   // [N][H][W][Cout]
    float* slab_w;     // [S][K * K][Cin][Cout]
    float* slab_b;     // [S][Cout] or null
    int N, H, W, Hs, Ws, Cin, Cout, up, pad;
    int CISp, COSp;    // LDS floats per pixel of the x / dz slice images
    int nci, nco, S;   // slices per side, sample shares
    int BR, nb_log, ns, units;
    int XR, XW;        // rows / columns of a unit's x window (halo included)
    int x_unit, z_unit;     // LDS floats per unit
    int xp_unit, zp_unit;   // 16-byte staging pieces per unit
    int qpu_log, wq_log;    // log2(quads per unit), log2(quads per image row)
    FDiv f_xp, f_zp, f_w;
};

constexpr int kWsChunk = 4;  // staging pieces per thread in flight

template <int K, int CT, int OT>
__global__ __launch_bounds__(256, 2) void wgrad_s1_mfma_k(WsArgs a) {
    constexpr int KK = K * K, NTILE = KK * CT * OT;
    constexpr int CIS = 16 * CT, COS = 16 * OT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4;
    int b = blockIdx.x;
    const int ic = b % a.nci; b /= a.nci;
    const int oc = b % a.nco;
    const int sp = b / a.nco;
    const int ci0 = ic * CIS, co0 = oc * COS;
    const int H = a.H, W = a.W, pad = a.pad;
    float* xs = smem;
    float* zs = smem + a.ns * a.x_unit;
    const int nb_mask = (1 << a.nb_log) - 1;

    // zero once: halo columns and padding channels are never written afterwards
    {
        const int total = (a.ns * (a.x_unit + a.z_unit)) >> 2;
        for (int i = threadIdx.x; i < total; i += 256) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    ws_v4f acc[KK][CT][OT];
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int o = 0; o < OT; ++o) acc[t][c][o] = ws_v4f{0.f, 0.f, 0.f, 0.f};
    float bsum[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) bsum[o] = 0.f;

    // ---- staging of one pass (units u0 .. u0 + nu): x window rows (zeros outside the image), dz rows
    auto stage = [&](int u0, int nu) {
        const int xtot = nu * a.xp_unit;
        for (int i0 = threadIdx.x; i0 < xtot; i0 += 256 * kWsChunk) {
            float4 v[kWsChunk];
            int at[kWsChunk];
#pragma unroll
            for (int c = 0; c < kWsChunk; ++c) {
                const int q = i0 + c * 256;
                v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                at[c] = -1;
                if (q >= xtot) continue;
                const int ul = fd(q, a.f_xp), r = q - ul * a.xp_unit;
                const int c4 = r & (CIS / 4 - 1), pix = r / (CIS / 4);
                const int row = fd(pix, a.f_w), col = pix - row * W;
                const int u = u0 + ul, n = u >> a.nb_log, r0 = (u & nb_mask) * a.BR;
                const int yy = r0 - pad + row, ch = ci0 + 4 * c4;
                at[c] = ul * a.x_unit + (row * a.XW + col + pad) * a.CISp + 4 * c4;
                if ((unsigned)yy < (unsigned)H && ch < a.Cin) {
                    const float* src = a.x + (((size_t)n * a.Hs + yy / a.up) * a.Ws + col / a.up) * a.Cin + ch;
                    if ((a.Cin & 3) == 0) {
                        v[c] = *reinterpret_cast<const float4*>(src);
                    } else {
                        v[c].x = src[0];
                        if (ch + 1 < a.Cin) v[c].y = src[1];
                        if (ch + 2 < a.Cin) v[c].z = src[2];
                        if (ch + 3 < a.Cin) v[c].w = src[3];
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < kWsChunk; ++c)
                if (at[c] >= 0) *reinterpret_cast<float4*>(xs + at[c]) = v[c];
        }
        const int ztot = nu * a.zp_unit;
        for (int i0 = threadIdx.x; i0 < ztot; i0 += 256 * kWsChunk) {
            float4 v[kWsChunk];
            int at[kWsChunk];
#pragma unroll
            for (int c = 0; c < kWsChunk; ++c) {
                const int q = i0 + c * 256;
                v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                at[c] = -1;
                if (q >= ztot) continue;
                const int ul = fd(q, a.f_zp), r = q - ul * a.zp_unit;
                const int c4 = r & (COS / 4 - 1), pix = r / (COS / 4);
                const int u = u0 + ul, n = u >> a.nb_log, r0 = (u & nb_mask) * a.BR;
                const int ch = co0 + 4 * c4;
                at[c] = ul * a.z_unit + pix * a.COSp + 4 * c4;
                if (ch < a.Cout) {
                    const float* src = a.dz + (((size_t)n * H + r0) * W + pix) * a.Cout + ch;
                    if ((a.Cout & 3) == 0) {
                        v[c] = *reinterpret_cast<const float4*>(src);
                    } else {
                        v[c].x = src[0];
                        if (ch + 1 < a.Cout) v[c].y = src[1];
                        if (ch + 2 < a.Cout) v[c].z = src[2];
                        if (ch + 3 < a.Cout) v[c].w = src[3];
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < kWsChunk; ++c)
                if (at[c] >= 0) *reinterpret_cast<float4*>(zs + at[c]) = v[c];
        }
    };

    // per-tap offsets into the x image (wave-uniform)
    int tapoff[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) tapoff[t] = ((t / K) * a.XW + (t % K)) * a.CISp;
    const int qpu_mask = (1 << a.qpu_log) - 1, wq_mask = (1 << a.wq_log) - 1;

    for (int u0 = sp * a.ns; u0 < a.units; u0 += a.S * a.ns) {
        const int nu = min(a.ns, a.units - u0);
        __syncthreads();  // zero fill landed / the previous pass's operand reads are done
        stage(u0, nu);
        __syncthreads();
        const int nq = (nu << a.qpu_log) >> 2;  // quads of this wave
        float av0[KK][CT], bv0[OT], av1[KK][CT], bv1[OT];
        auto read_quad = [&](int i, float (&av)[KK][CT], float (&bv)[OT]) {
            const int q = wave + 4 * i;
            const int ul = q >> a.qpu_log, rem = q & qpu_mask;
            const int y = rem >> a.wq_log, x0 = (rem & wq_mask) << 2;
            const float* ap = xs + ul * a.x_unit + (y * a.XW + x0 + lk) * a.CISp + li;
            const float* bp = zs + ul * a.z_unit + ((y * W) + x0 + lk) * a.COSp + li;
#pragma unroll
            for (int o = 0; o < OT; ++o) bv[o] = bp[16 * o];
#pragma unroll
            for (int t = 0; t < KK; ++t)
#pragma unroll
                for (int c = 0; c < CT; ++c) av[t][c] = ap[tapoff[t] + 16 * c];
        };
        auto mfma_quad = [&](const float (&av)[KK][CT], const float (&bv)[OT]) {
#pragma unroll
            for (int o = 0; o < OT; ++o) bsum[o] += bv[o];
#pragma unroll
            for (int t = 0; t < KK; ++t)
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int o = 0; o < OT; ++o)
                        acc[t][c][o] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][c], bv[o], acc[t][c][o], 0, 0, 0);
        };
        if (nq > 0) read_quad(0, av0, bv0);
        for (int i = 0; i < nq; i += 2) {
            if (i + 1 < nq) read_quad(i + 1, av1, bv1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_quad(av0, bv0);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 2 < nq) read_quad(i + 2, av0, bv0);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < nq) mfma_quad(av1, bv1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- sum the 4 waves as (w0 + w1) + (w2 + w3) through LDS (the images are dead), then wave 0 writes the slice
    float4* xch = reinterpret_cast<float4*>(smem);  // [2][NTILE + 1][64]
#pragma unroll
    for (int o = 0; o < OT; ++o) {  // the 4 pixel lanes of an output channel
        bsum[o] += __shfl_xor(bsum[o], 16, 64);
        bsum[o] += __shfl_xor(bsum[o], 32, 64);
    }
    auto put = [&](int slot) {
        float4* d = xch + (size_t)slot * (NTILE + 1) * 64 + lane;
#pragma unroll
        for (int t = 0; t < KK; ++t)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int o = 0; o < OT; ++o)
                    d[((t * CT + c) * OT + o) * 64] = make_float4(acc[t][c][o][0], acc[t][c][o][1], acc[t][c][o][2], acc[t][c][o][3]);
        d[NTILE * 64] = make_float4(bsum[0], OT > 1 ? bsum[OT - 1] : 0.f, 0.f, 0.f);
    };
    auto add = [&](int slot) {
        const float4* d = xch + (size_t)slot * (NTILE + 1) * 64 + lane;
#pragma unroll
        for (int t = 0; t < KK; ++t)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int o = 0; o < OT; ++o) {
                    const float4 v = d[((t * CT + c) * OT + o) * 64];
                    acc[t][c][o][0] += v.x; acc[t][c][o][1] += v.y; acc[t][c][o][2] += v.z; acc[t][c][o][3] += v.w;
                }
        const float4 bb = d[NTILE * 64];
        bsum[0] += bb.x;
        if (OT > 1) bsum[OT - 1] += bb.y;
    };
    __syncthreads();
    if (wave == 1) put(0);
    if (wave == 3) put(1);
    __syncthreads();
    if (wave == 0) add(0);
    if (wave == 2) add(1);
    __syncthreads();
    if (wave == 2) put(0);
    __syncthreads();
    if (wave != 0) return;
    add(0);
    // lane (li, lk), register r of tile (tap, c, o): dw[tap][ci0 + 16 c + 4 lk + r][co0 + 16 o + li]
    float* out = a.slab_w + (size_t)sp * KK * a.Cin * a.Cout;
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int o = 0; o < OT; ++o) {
                const int co = co0 + 16 * o + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = ci0 + 16 * c + 4 * lk + r;
                    if (ci < a.Cin && co < a.Cout) out[((size_t)t * a.Cin + ci) * a.Cout + co] = acc[t][c][o][r];
                }
            }
    if (a.slab_b != nullptr && ic == 0 && lk == 0) {
#pragma unroll
        for (int o = 0; o < OT; ++o) {
            const int co = co0 + 16 * o + li;
            if (co < a.Cout) a.slab_b[(size_t)sp * a.Cout + co] = bsum[o];
        }
    }
}

// dz[N][H][W][C] from the gradient dy with respect to the layer output y (both at the pooled extent when idx != null):
// dy * act'(y), placed at each pooling window's argmax and zero elsewhere.
__global__ __launch_bounds__(256) void form_dz_k(const float* __restrict__ dy, const float* __restrict__ y,
                                                  const uint8_t* __restrict__ idx, float* __restrict__ dz, int64_t n_el,
                                                  int C, int Wp, int pooled, ActGradCoef zk) {
    // one thread per stored element group: 4 consecutive channels when C % 4 == 0, else one element
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if ((C & 3) == 0) {
        const int64_t n4 = n_el >> 2;
        const int c4n = C >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 v = reinterpret_cast<const float4*>(dy)[i];
            const float4 yy = reinterpret_cast<const float4*>(y)[i];
            v.x *= act_grad_q(yy.x, zk); v.y *= act_grad_q(yy.y, zk); v.z *= act_grad_q(yy.z, zk); v.w *= act_grad_q(yy.w, zk);
            if (!pooled) {
                reinterpret_cast<float4*>(dz)[i] = v;
                continue;
            }
            const uint32_t k = reinterpret_cast<const uint32_t*>(idx)[i];
            const uint32_t k0 = k & 0xff, k1 = (k >> 8) & 0xff, k2 = (k >> 16) & 0xff, k3 = k >> 24;
            const int64_t pix = i / c4n;           // pooled pixel (n, py, px) flattened
            const int c4 = (int)(i - pix * c4n);
            const int64_t prow = pix / Wp;          // n * Hp + py
            const int px = (int)(pix - prow * Wp);
            float* q = dz + ((prow * 2) * (2 * Wp) + 2 * px) * (int64_t)C + 4 * c4;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 o = make_float4(k0 == (uint32_t)s ? v.x : 0.f, k1 == (uint32_t)s ? v.y : 0.f,
                                             k2 == (uint32_t)s ? v.z : 0.f, k3 == (uint32_t)s ? v.w : 0.f);
                *reinterpret_cast<float4*>(q + ((int64_t)(s >> 1) * (2 * Wp) + (s & 1)) * C) = o;
            }
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += stride) {
            const float v = dy[i] * act_grad_q(y[i], zk);
            if (!pooled) {
                dz[i] = v;
                continue;
            }
            const int64_t pix = i / C;
            const int c = (int)(i - pix * C);
            const int64_t prow = pix / Wp;
            const int px = (int)(pix - prow * Wp);
            float* q = dz + ((prow * 2) * (2 * Wp) + 2 * px) * (int64_t)C + c;
            const int k = idx[i];
#pragma unroll
            for (int s = 0; s < 4; ++s) q[((int64_t)(s >> 1) * (2 * Wp) + (s & 1)) * C] = k == s ? v : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- host side
struct WsPlan {
    bool ok;
    int K, ct, ot;
    WsArgs a;
    size_t lds;
    int grid;
};

constexpr size_t kWsLdsBudget = 80 * 1024;

static WsPlan plan_ws(const sv_conv_geom& g) {
    WsPlan p{};
    p.ok = false;
    if (!(g.KH == g.KW && (g.KH == 3 || g.KH == 5) && g.SH == 1 && g.SW == 1 && g.pad_t == (g.KH - 1) / 2 &&
          g.pad_l == (g.KW - 1) / 2 && g.Ho == g.H && g.Wo == g.W && g.up_h == g.up_w && (g.up_h == 1 || g.up_h == 2)))
        return p;
    const int K = g.KH, pad = (K - 1) / 2, H = g.H, W = g.W, up = g.up_h;
    if (!(W == 4 || W == 8 || W == 16 || W == 32) || H < 4 || (H & (H - 1)) || H > 64) return p;
    if (up == 2 && ((H | W) & 1)) return p;
    if (g.Cin > 128 || g.Cout > 128) return p;
    if ((int64_t)g.N * H * W * (g.Cin > g.Cout ? g.Cin : g.Cout) >= (1ll << 31)) return p;
    const int ct = (K == 3 && g.Cin >= 32 && g.Cout >= 32) ? 2 : 1, ot = ct;
    const int cis = 16 * ct, cos = 16 * ot;
    WsArgs& a = p.a;
    a.N = g.N; a.H = H; a.W = W; a.Hs = H / up; a.Ws = W / up; a.Cin = g.Cin; a.Cout = g.Cout; a.up = up; a.pad = pad;
    a.CISp = cis == 32 ? 48 : 16; a.COSp = cos == 32 ? 48 : 16;
    a.nci = (g.Cin + cis - 1) / cis; a.nco = (g.Cout + cos - 1) / cos;
    // a unit = the largest band of rows (whole images where they fit) whose x window and dz rows take at most 48 KB
    a.XW = W + 2 * pad;
    for (a.BR = H; a.BR >= 2; a.BR >>= 1) {
        a.XR = a.BR + 2 * pad;
        a.x_unit = (a.XR * a.XW * a.CISp + 3) & ~3;
        a.z_unit = a.BR * W * a.COSp;
        if ((size_t)(a.x_unit + a.z_unit) * sizeof(float) <= 48 * 1024) break;
    }
    if (a.BR < 2 || a.BR * W < 16) return p;
    const int nb = H / a.BR;
    a.nb_log = __builtin_ctz((unsigned)nb);
    a.units = g.N * nb;
    a.xp_unit = a.XR * W * (cis / 4);
    a.zp_unit = a.BR * W * (cos / 4);
    a.qpu_log = __builtin_ctz((unsigned)(a.BR * W / 4));
    a.wq_log = __builtin_ctz((unsigned)(W / 4));
    a.f_xp = fd_make(a.xp_unit); a.f_zp = fd_make(a.zp_unit); a.f_w = fd_make(W);
    // units per pass: at least 16 quads (4 per wave), as many as 48 KB of LDS hold, at most 16
    const size_t unit_bytes = (size_t)(a.x_unit + a.z_unit) * sizeof(float);
    int ns = (int)(48 * 1024 / unit_bytes);
    if (ns < 1) ns = 1;
    if (ns > 16) ns = 16;
    if (ns > a.units) ns = a.units;
    if (((ns * a.BR * W) >> 2) < 4) return p;
    a.ns = ns;
    const int ntile = K * K * ct * ot;
    size_t lds = (size_t)ns * unit_bytes;
    const size_t xch = (size_t)2 * (ntile + 1) * 64 * sizeof(float4);
    if (xch > lds) lds = xch;
    if (lds > kWsLdsBudget) return p;
    // sample shares: about 512 workgroups in all, every workgroup at least two passes where the batch allows
    const int pairs = a.nci * a.nco;
    int S = 512 / pairs;
    const int max_s = (a.units + 2 * ns - 1) / (2 * ns);
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    a.S = S;
    p.K = K; p.ct = ct; p.ot = ot; p.lds = (lds + 15) & ~(size_t)15;
    p.grid = pairs * S;
    p.ok = true;
    return p;
}

template <int K, int CT, int OT>
static int ws_launch(const WsPlan& p, hipStream_t st) {
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_s1_mfma_k<K, CT, OT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWsLdsBudget);
        if (e != hipSuccess) return fail(SV_E_LAUNCH, "wgrad_s1_mfma: raising the LDS limit failed: %s", hipGetErrorString(e));
        raised = true;
    }
    hipLaunchKernelGGL((wgrad_s1_mfma_k<K, CT, OT>), dim3((unsigned)p.grid), dim3(256), p.lds, st, p.a);
    return check_launch("wgrad_s1_mfma");
}

bool wgrad_s1_mfma_supported(const sv_conv_geom& g) { return opt(OPT_S1_MFMA) != 0 && plan_ws(g).ok; }

// bytes of partial sums launch_wgrad_s1_mfma needs (0: geometry outside the family)
size_t wgrad_s1_mfma_workspace(const sv_conv_geom& g) {
    const WsPlan p = plan_ws(g);
    if (!p.ok) return 0;
    return (size_t)p.a.S * ((size_t)g.KH * g.KW * g.Cin * g.Cout + g.Cout) * sizeof(float);
}

int launch_slab_sum(const float* part, float* dst, int64_t n, int slabs, int accumulate, hipStream_t st);  // conv_direct.hip

// dw (+)= kernel gradient, db (+)= bias gradient (db may be null) from x and an explicit dz.  1: geometry outside the
// family; 0: launched; < 0: error.
int launch_wgrad_s1_mfma(const sv_conv_geom& g, const float* x, const float* dz, float* dw, float* db, int accumulate,
                         void* ws, size_t ws_bytes, hipStream_t st) {
    if (opt(OPT_S1_MFMA) == 0) return 1;
    WsPlan p = plan_ws(g);
    if (!p.ok) return 1;
    const size_t need = wgrad_s1_mfma_workspace(g);
    if (ws == nullptr || ws_bytes < need)
        return fail(SV_E_WORKSPACE, "wgrad_s1_mfma: workspace %zu bytes < required %zu", ws_bytes, need);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(ws)) & 15) return 1;
    const int64_t kc = (int64_t)g.KH * g.KW * g.Cin * g.Cout;
    p.a.x = x; p.a.dz = dz;
    p.a.slab_w = (float*)ws;
    p.a.slab_b = db != nullptr ? (float*)ws + (size_t)p.a.S * kc : nullptr;
    int rc;
    if (p.K == 5) rc = ws_launch<5, 1, 1>(p, st);
    else if (p.ct == 2) rc = ws_launch<3, 2, 2>(p, st);
    else rc = ws_launch<3, 1, 1>(p, st);
    if (rc) return rc;
    rc = launch_slab_sum(p.a.slab_w, dw, kc, p.a.S, accumulate, st);
    if (rc || db == nullptr) return rc;
    return launch_slab_sum(p.a.slab_b, db, g.Cout, p.a.S, accumulate, st);
}

int launch_form_dz(const float* dy, const float* y, const uint8_t* idx, float* dz, int N, int H, int W, int C, int act,
                   float alpha, hipStream_t st) {
    const bool pooled = idx != nullptr;
    const int64_t n_el = pooled ? (int64_t)N * (H / 2) * (W / 2) * C : (int64_t)N * H * W * C;
    const int64_t items = (C & 3) == 0 ? n_el / 4 : n_el;
    hipLaunchKernelGGL(form_dz_k, dim3(ew_grid(items, 256, 4096)), dim3(256), 0, st, dy, y, idx, dz, n_el, C, W / 2,
                       pooled ? 1 : 0, act_grad_coef(act, alpha));
    return check_launch("form_dz");
}

}  // namespace sv
