// Whole-network kernels for chains of stride-1 SAME convs (optionally with a folded 2x nearest upsampling on the
// input and a folded 2x2 max-pool on the output): the CAEPadded family without BatchNorm / Dense.
//
// Samples are independent through every layer, so one workgroup carries S samples through ALL layers without
// leaving the CU: activations live in two LDS arenas as planar images with a zero halo (the layout the register-tile
// FMA loop reads), layer l reads arena l % 2 and writes arena (l + 1) % 2.  Per layer the workgroup
//   [A] zeroes the output arena and drops the (prefetched) filters + bias of the layer into the filter arena,
//   [B] puts the next layer's filter loads in flight (registers),
//   [C] computes: a thread owns 2x2 output pixels x CO_T channels, bias + activation (+ 2x2 max-pool with argmax)
//       in registers, results written to the output arena in the next layer's input layout,
//   [E] copies the output arena to global memory as NHWC with full-width stores (the backward pass reads it).
// One launch replaces one launch per layer, and no layer waits for a global-memory round trip of its input.
#include <cstdlib>
#include <type_traits>

#include <utility>

#include "tile_common.h"
#include "dgrad_nhwc_body.h"

namespace sv {

#ifdef SV_TRACE
// diagnostic builds: thread 0 of every workgroup stores the 100 MHz wall clock at phase marks, 32 slots per workgroup
__device__ unsigned long long* g_net_trace = nullptr;
__device__ __forceinline__ void net_mark(int slot) {
    if (threadIdx.x == 0 && g_net_trace != nullptr && slot < 32) g_net_trace[(size_t)blockIdx.x * 32 + slot] = wall_clock64();
}
#define NET_MARK(slot) net_mark(slot)
#else
#define NET_MARK(slot)
#endif

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kMaxNetLayers = 12;
constexpr int kNetWPre = 6;  // float4 registers per thread holding the next layer's filters

struct NetLayer {
    int Cin, Cout;
    int Hs, Ws;      // stored input extent
    int up, pool;    // input read through 2x nearest upsampling / output 2x2 max-pooled
    int act;
    float alpha;
    int w_off, b_off;                  // float offsets into the flat parameter buffer (b_off < 0: no bias)
    int in_off, out_off;               // LDS float offsets of the input / output arena
    int idx_off;                       // LDS float offset of this layer's argmax bytes (two arenas, by layer parity: the
                                       // copy-out of layer l runs beside the arithmetic of layer l + 1)
    int in_plane, in_wp, in_halo;      // planar layout of the input
    int out_plane, out_wp, out_halo;   // planar layout of the output (= the next layer's input layout)
    int co_t, n_cot, ks;
    int pair;        // vector path, 1: a thread carries its pixel tile through two samples at once (a.S is even);
                     // the variants compiled for it are listed in net_layer_fwd_any
    int mfma;        // 0: vector path; 1..3: the matrix-core variant compiled for this layer shape (net_layer_fwd_mfma)
    float* y;        // global output [N][Hout][Wout][Cout] (pooled extent when pool)
    uint8_t* idx;    // global argmax slots (pool only, may be null)
};

struct NetFwdArgs {
    const float* x;
    const float* params;
    int N, S, n_layers;
    int w_off;    // LDS float offset of the filter arena ([tap][Cin][cop] + cop bias values)
    // squared-error loss evaluated while the network output leaves (all null / zero when the forward runs alone):
    // loss_t = targets shaped like the output, loss_dpred receives loss_gscale * (y - t), loss_partials[blockIdx.x] the
    // workgroup's sum of (y - t)^2
    const float* loss_t;
    float* loss_dpred;
    float* loss_partials;
    float loss_gscale;
    // the batch gathered while it is staged (rows.row_index == null: x is the batch): sample n is row
    // row_index[(counter[0] - counter_base[0]) * batch_stride + n] of x_rows and is also written to x_batch[n]
    sv_batch_rows rows;
    NetLayer L[kMaxNetLayers];
};

// Staging of the network input through a row index, in two halves so that the LDS zero fill runs while the loads are
// in flight: each sample's row is looked up once per thread (S <= 4 samples a pass, at most U * nt 16-byte pieces), its
// pieces go to the planar LDS image and, as they are, to the gathered batch in global memory.
template <int U>
__device__ __forceinline__ void net_rows_load(const NetFwdArgs& a, const PlanarSrc& s, int n0, int ns, int nt, float4 (&v)[U]) {
    const sv_batch_rows& r = a.rows;
    const int HW = s.Hs * s.Ws, per = (HW * s.C) >> 2;  // 16-byte pieces per sample
    const int units = ns * per;
    const long long first = r.counter != nullptr ? (r.counter[0] - r.counter_base[0]) * r.batch_stride : 0;
    long long row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long long pos = first + n0 + min(i, ns - 1);
        pos = pos < 0 ? 0 : (pos >= r.n_index ? r.n_index - 1 : pos);
        row[i] = r.row_index[pos];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) row[i] = row[i] < 0 ? 0 : (row[i] >= r.n_rows ? r.n_rows - 1 : row[i]);
    const FDiv fper = fd_make(per);
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int u = min((int)threadIdx.x + i * nt, units - 1);
        const int ss = fd(u, fper), off = u - ss * per;
        const long long rw = ss == 0 ? row[0] : (ss == 1 ? row[1] : (ss == 2 ? row[2] : row[3]));
        v[i] = reinterpret_cast<const float4*>(r.x_rows + (size_t)rw * (size_t)(HW * s.C))[off];
    }
}
template <int U>
__device__ __forceinline__ void net_rows_store(const NetFwdArgs& a, const PlanarSrc& s, const PlanarDst& d, int n0, int ns,
                                               float* lds, int nt, const float4 (&v)[U]) {
    const int HW = s.Hs * s.Ws, per = (HW * s.C) >> 2;
    const int units = ns * per;
    const FDiv fC = fd_make(s.C), fHW = fd_make(HW), fW = fd_make(s.Ws);
    float4* out = reinterpret_cast<float4*>(a.rows.x_batch) + (size_t)n0 * per;
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int u = threadIdx.x + i * nt;
        if (u < units) {
            out[u] = v[i];
            const float xv[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int e = 4 * u + t;
                const int p = fd(e, fC), c = e - p * s.C;
                const int ss = fd(p, fHW), pp = p - ss * HW;
                const int rr = fd(pp, fW), col = pp - rr * s.Ws;
                stage_put<0>(s, d, lds + (size_t)(ss * d.cpl + c) * d.plane, rr, col, xv[t], 0.f, 0u);
            }
        }
    }
}

// ---- filters: global -> registers (prefetch) -> LDS [tap][ci][cop] followed by cop bias values ----------------------
template <int K>
__device__ __forceinline__ void net_w_load(const NetFwdArgs& a, const NetLayer& L, float4 (&wpre)[kNetWPre], float& bpre,
                                           int nt) {
    const int cop = L.n_cot * L.co_t;
    const float* w = a.params + L.w_off;
    const int rows = K * K * L.Cin;
    if ((L.Cout & 3) == 0 && (L.w_off & 3) == 0) {
        const int c4n = cop >> 2, units = rows * c4n;
        const FDiv f4 = fd_make(c4n);
#pragma unroll
        for (int i = 0; i < kNetWPre; ++i) {
            const int u = threadIdx.x + i * nt;
            wpre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < units) {
                const int r = fd(u, f4), c4 = u - r * c4n;
                if (4 * c4 < L.Cout) wpre[i] = *reinterpret_cast<const float4*>(w + (size_t)r * L.Cout + 4 * c4);
            }
        }
    } else {
        const int units = rows * cop;
        const FDiv fcop = fd_make(cop);
#pragma unroll
        for (int i = 0; i < kNetWPre; ++i) {
            const int u = threadIdx.x + i * nt;
            wpre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < units) {
                const int r = fd(u, fcop), co = u - r * cop;
                if (co < L.Cout) wpre[i].x = w[(size_t)r * L.Cout + co];
            }
        }
    }
    bpre = (L.b_off >= 0 && (int)threadIdx.x < L.Cout) ? a.params[L.b_off + threadIdx.x] : 0.f;
}

// LDS filter layout [channel tile][ci][tap][co_t] for a runtime co_t
template <int K>
__device__ __forceinline__ int net_w_index(const NetLayer& L, int co, int c, int tap) {
    return L.co_t == 8 ? w_lds_index<K, 8>(co, c, tap, L.Cin) : w_lds_index<K, 4>(co, c, tap, L.Cin);
}

template <int K>
__device__ __forceinline__ void net_w_store(const NetFwdArgs& a, const NetLayer& L, const float4 (&wpre)[kNetWPre],
                                            float bpre, float* w_lds, int nt) {
    const int cop = L.n_cot * L.co_t;
    const FDiv fCin = fd_make(L.Cin);
    const float* w = a.params + L.w_off;
    const int rows = K * K * L.Cin;
    if ((L.Cout & 3) == 0 && (L.w_off & 3) == 0) {
        const int c4n = cop >> 2, units = rows * c4n;
        const FDiv f4 = fd_make(c4n);
#pragma unroll
        for (int i = 0; i < kNetWPre; ++i) {
            const int u = threadIdx.x + i * nt;
            if (u < units) {
                const int r = fd(u, f4), c4 = u - r * c4n;
                const int tap = fd(r, fCin), c = r - tap * L.Cin;
                *reinterpret_cast<float4*>(w_lds + net_w_index<K>(L, 4 * c4, c, tap)) = wpre[i];
            }
        }
        for (int u = threadIdx.x + kNetWPre * nt; u < units; u += nt) {  // filters larger than the prefetch registers
            const int r = fd(u, f4), c4 = u - r * c4n;
            const int tap = fd(r, fCin), c = r - tap * L.Cin;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * c4 < L.Cout) v = *reinterpret_cast<const float4*>(w + (size_t)r * L.Cout + 4 * c4);
            *reinterpret_cast<float4*>(w_lds + net_w_index<K>(L, 4 * c4, c, tap)) = v;
        }
    } else {
        const int units = rows * cop;
        const FDiv fcop = fd_make(cop);
#pragma unroll
        for (int i = 0; i < kNetWPre; ++i) {
            const int u = threadIdx.x + i * nt;
            if (u < units) {
                const int r = fd(u, fcop), co = u - r * cop;
                const int tap = fd(r, fCin), c = r - tap * L.Cin;
                w_lds[net_w_index<K>(L, co, c, tap)] = wpre[i].x;
            }
        }
        for (int u = threadIdx.x + kNetWPre * nt; u < units; u += nt) {
            const int r = fd(u, fcop), co = u - r * cop;
            const int tap = fd(r, fCin), c = r - tap * L.Cin;
            w_lds[net_w_index<K>(L, co, c, tap)] = co < L.Cout ? w[(size_t)r * L.Cout + co] : 0.f;
        }
    }
    if ((int)threadIdx.x < cop) w_lds[rows * cop + threadIdx.x] = bpre;
}

// ---- one layer, LDS -> LDS ---------------------------------------------------------------------------------------------
// The register-tile FMA loop of tile_common.h for NS samples at once: a thread owns the same 2x2 output pixels x CO_T
// channels in NS samples, so one filter read (a broadcast LDS read, the larger part of this loop's LDS traffic) feeds
// NS times the arithmetic.
template <int K, int UP, int CO_T, int NS>
__device__ __forceinline__ void net_tile_fma(float (&acc)[NS][2][2][CO_T], const float* img, int sample_stride,
                                             const float* wtile, int plane, int wp, int r0, int c0, int c_begin, int c_end,
                                             int c_step) {
    constexpr int PADL = (K - 1) / 2;
    constexpr int NPR = patch_row<UP, PADL>(K) + 1;  // patch rows == patch cols
    for (int c = c_begin; c < c_end; c += c_step) {
        float patch[NS][NPR][NPR + 1];
#pragma unroll
        for (int e = 0; e < NS; ++e) {
            const float* pl = img + e * sample_stride + c * plane + r0 * wp + c0;
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {
                if (UP == 1) {  // NPR = K + 1 is even and c0 is even: 8-byte aligned pairs
#pragma unroll
                    for (int pc = 0; pc < NPR; pc += 2) {
                        const float2 v2 = *reinterpret_cast<const float2*>(pl + pr * wp + pc);
                        patch[e][pr][pc] = v2.x;
                        patch[e][pr][pc + 1] = v2.y;
                    }
                } else {
#pragma unroll
                    for (int pc = 0; pc < NPR; ++pc) patch[e][pr][pc] = pl[pr * wp + pc];
                }
            }
        }
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
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
                for (int e = 0; e < NS; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float v = patch[e][patch_row<UP, PADL>(i + kh)][patch_row<UP, PADL>(j + kw)];
#pragma unroll
                            for (int q = 0; q < CO_T; ++q) acc[e][i][j][q] = fmaf(v, wv[q], acc[e][i][j][q]);
                        }
            }
            // with two samples' sums and patches in registers the filter reads of one kernel row at a time may be in
            // flight, not those of the whole channel
            if (NS > 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// NS = 2 needs an even a.S: item = (channel tile, sample pair, pixel tile, channel share).
template <int K, int UP, int CO_T, int POOL, int NS>
__device__ __forceinline__ void net_layer_fwd(const NetFwdArgs& a, const NetLayer& L, float* smem, int ns, int nt) {
    constexpr int PADL = (K - 1) / 2;
    const int Ho = L.Hs * UP, Wo = L.Ws * UP;
    const int tiles_x = Wo >> 1, tiles_y = Ho >> 1;
    const int T = tiles_x * tiles_y;
    const int cop = L.n_cot * CO_T;
    const int ks = L.ks;
    const int per_cot = (a.S / NS) * T * ks;
    const int total = L.n_cot * per_cot;
    const float* in = smem + L.in_off;
    float* out = smem + L.out_off;
    const float* w_lds = smem + a.w_off;
    const float* b_lds = w_lds + K * K * L.Cin * cop;
    uint8_t* idx_lds = reinterpret_cast<uint8_t*>(smem + L.idx_off);
    const int oh = L.out_halo;
    const FDiv f_percot = fd_make(per_cot), f_ks = fd_make(ks), f_T = fd_make(T), f_tx = fd_make(tiles_x);

    for (int item0 = 0; item0 < total; item0 += nt) {
        const int item = item0 + threadIdx.x;
        const bool live = item < total;
        int cot = 0, sp = 0, ty = 0, tx = 0, kl = 0;
        if (live) {
            cot = fd(item, f_percot);
            const int r = item - cot * per_cot;
            const int r1 = fd(r, f_ks);
            kl = r - r1 * ks;
            sp = fd(r1, f_T);
            const int t = r1 - sp * T;
            ty = fd(t, f_tx);
            tx = t - ty * tiles_x;
        }
        const int s0 = sp * NS;
        const bool work = live && s0 < ns;
        float acc[NS][2][2][CO_T];
#pragma unroll
        for (int e = 0; e < NS; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int c = 0; c < CO_T; ++c) acc[e][i][j][c] = 0.f;
        if (work) {
            const int r0 = (UP == 1 ? 2 * ty : ty) + L.in_halo + floordiv_c(-PADL, UP);
            const int c0 = (UP == 1 ? 2 * tx : tx) + L.in_halo + floordiv_c(-PADL, UP);
            // a sample past ns (the last, short pass) reads its zeroed image; its results are dropped below
            net_tile_fma<K, UP, CO_T, NS>(acc, in + (size_t)s0 * L.Cin * L.in_plane, L.Cin * L.in_plane,
                                          w_lds + cot * L.Cin * K * K * CO_T, L.in_plane, L.in_wp, r0, c0, kl, L.Cin, ks);
        }
        for (int o = 1; o < ks; o <<= 1) {
#pragma unroll
            for (int e = 0; e < NS; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < CO_T; ++q) acc[e][i][j][q] += __shfl_xor(acc[e][i][j][q], o, 64);
        }
        if (work && kl == 0) {
            const int cg = cot * CO_T;
#pragma unroll
            for (int e = 0; e < NS; ++e) {
                const int s = s0 + e;
                if (s >= ns) continue;
                if (POOL == 0) {
#pragma unroll
                    for (int q = 0; q < CO_T; ++q)
                        if (cg + q < L.Cout) {
                            const float b = b_lds[cg + q];
                            float* o = out + (size_t)(s * L.Cout + cg + q) * L.out_plane + (2 * ty + oh) * L.out_wp + 2 * tx + oh;
                            o[0] = act_apply(acc[e][0][0][q] + b, L.act, L.alpha);
                            o[1] = act_apply(acc[e][0][1][q] + b, L.act, L.alpha);
                            o[L.out_wp] = act_apply(acc[e][1][0][q] + b, L.act, L.alpha);
                            o[L.out_wp + 1] = act_apply(acc[e][1][1][q] + b, L.act, L.alpha);
                        }
                } else {
                    // max of the pre-activations, then one activation per window (all activations here are
                    // non-decreasing); relu's flattened ties record slot 0, the slot the unfused order would pick
#pragma unroll
                    for (int q = 0; q < CO_T; ++q)
                        if (cg + q < L.Cout) {
                            float best = acc[e][0][0][q];
                            uint32_t bi = 0;
                            if (acc[e][0][1][q] > best) { best = acc[e][0][1][q]; bi = 1; }
                            if (acc[e][1][0][q] > best) { best = acc[e][1][0][q]; bi = 2; }
                            if (acc[e][1][1][q] > best) { best = acc[e][1][1][q]; bi = 3; }
                            const float v = act_apply(best + b_lds[cg + q], L.act, L.alpha);
                            if (L.act == SV_ACT_RELU && v <= 0.f) bi = 0;
                            out[(size_t)(s * L.Cout + cg + q) * L.out_plane + (ty + oh) * L.out_wp + tx + oh] = v;
                            idx_lds[((size_t)s * T + ty * tiles_x + tx) * L.Cout + cg + q] = (uint8_t)bi;
                        }
                }
            }
        }
    }
}

template <int K>
__device__ __forceinline__ void net_layer_fwd_any(const NetFwdArgs& a, const NetLayer& L, float* smem, int ns, int nt) {
    if (L.pair == 1) net_layer_fwd<K, 1, 8, 1, 2>(a, L, smem, ns, nt);
    else if (L.up == 1) {
        if (L.co_t == 8) {
            if (L.pool) net_layer_fwd<K, 1, 8, 1, 1>(a, L, smem, ns, nt);
            else net_layer_fwd<K, 1, 8, 0, 1>(a, L, smem, ns, nt);
        } else {
            if (L.pool) net_layer_fwd<K, 1, 4, 1, 1>(a, L, smem, ns, nt);
            else net_layer_fwd<K, 1, 4, 0, 1>(a, L, smem, ns, nt);
        }
    } else {
        if (L.co_t == 8) {
            if (L.pool) net_layer_fwd<K, 2, 8, 1, 1>(a, L, smem, ns, nt);
            else net_layer_fwd<K, 2, 8, 0, 1>(a, L, smem, ns, nt);
        } else {
            if (L.pool) net_layer_fwd<K, 2, 4, 1, 1>(a, L, smem, ns, nt);
            else net_layer_fwd<K, 2, 4, 0, 1>(a, L, smem, ns, nt);
        }
    }
}

// ---- one layer on the matrix cores, LDS -> LDS (3x3; CIN in {8, 16, 32} input channels, 16 * NT output channels; conv
// extent 8 or 16 pixels wide).  y[px][co] = sum_{tap, ci} x[px + tap][ci] * w[tap][ci][co] as v_mfma_f32_16x16x4_f32
// tiles: 16 pixels x 16 output channels, 4 input channels of one tap per instruction.  Lane l supplies
// A[pixel l % 16][channel l / 16] from the planar input arena and B[channel l / 16][co l % 16] from the filter arena
// (the vector path's [co / 8][ci][tap][8] layout), so both are 4-byte LDS reads at a per-lane base plus constants.
// A wave owns units of 32 pixels (two tiles: 4 rows x 8 or 2 rows x 16), which contain whole 2 x 2 pooling windows.
typedef float nf_v4f __attribute__((ext_vector_type(4)));

// Order of one turn of a matrix-core loop: NM matrix instructions with the NR LDS reads of the NEXT turn's operands
// placed two behind each of the first ones (the reads issue in the shadow of the matrix pipe).
template <int NM, int NR, int... I>
__device__ __forceinline__ void nf_interleave_seq(std::integer_sequence<int, I...>) {
    ((__builtin_amdgcn_sched_group_barrier(0x008, 1, 0),
      __builtin_amdgcn_sched_group_barrier(0x100, (NR - 2 * I) >= 2 ? 2 : ((NR - 2 * I) > 0 ? (NR - 2 * I) : 0), 0)), ...);
}
template <int NM, int NR>
__device__ __forceinline__ void nf_interleave() {
    nf_interleave_seq<NM, NR>(std::make_integer_sequence<int, NM>{});
}

template <int CIN, int NT, int UP, int POOL>
__device__ __forceinline__ void net_layer_fwd_mfma(const NetFwdArgs& a, const NetLayer& L, float* smem, int ns, int nt) {
    constexpr int K = 3, KK = 9, CO8 = 8;
    const int Ho = L.Hs * UP, Wo = L.Ws * UP;
    const bool wide = Wo == 16;  // else Wo == 8
    const int units_ps = (Ho * Wo) >> 5;
    const int n_units = ns * units_ps;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nt >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const float* in = smem + L.in_off;
    float* out = smem + L.out_off;
    const float* w_lds = smem + a.w_off;
    const int cop = L.n_cot * CO8;
    const float* b_lds = w_lds + KK * CIN * cop;
    uint8_t* idx_lds = reinterpret_cast<uint8_t*>(smem + L.idx_off);
    const int oh = L.out_halo, ih = L.in_halo;
    int boff[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 16 + li;
        boff[n] = ((co >> 3) * CIN + lk) * (KK * CO8) + (co & 7);
    }
    const FDiv f_ups = fd_make(units_ps);
    const int T = (Ho >> 1) * (Wo >> 1), tiles_x = Wo >> 1;
    for (int u = wave; u < n_units; u += nw) {
        const int s = fd(u, f_ups), uu = u - s * units_ps;
        // pixel (row, col) of A row li in the two tiles of the unit
        int pr[2], pc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            pr[t] = wide ? 2 * uu + t : 4 * uu + 2 * t + (li >> 3);
            pc[t] = wide ? li : (li & 7);
        }
        nf_v4f acc[2][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = nf_v4f{0.f, 0.f, 0.f, 0.f};
        const float* ins = in + ((size_t)s * CIN + lk) * L.in_plane;
        // operands of a whole tap (all channel groups) are read into registers one tap ahead of the MFMAs that use them,
        // so the matrix pipe never waits on an LDS round trip issued just before it
        constexpr int CG = CIN / 4;
        float av[2][CG][2], bv[2][CG][NT];
        auto load_tap = [&](int tap, float (&a_)[CG][2], float (&b_)[CG][NT]) {
            const int dr = tap / K - 1, dc = tap % K - 1;
            int ao[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int sr = UP == 1 ? pr[t] + dr : ((pr[t] + dr) >> 1);  // arithmetic shift: -1 stays -1 (the halo)
                const int sc = UP == 1 ? pc[t] + dc : ((pc[t] + dc) >> 1);
                ao[t] = (sr + ih) * L.in_wp + sc + ih;
            }
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
#pragma unroll
                for (int t = 0; t < 2; ++t) a_[cg][t] = ins[cg * 4 * L.in_plane + ao[t]];
#pragma unroll
                for (int n = 0; n < NT; ++n) b_[cg][n] = w_lds[boff[n] + cg * 4 * (KK * CO8) + tap * CO8];
            }
        };
        constexpr bool MIX = CG * (2 + NT) <= 2 * CG * 2 * NT;  // at most two reads per matrix instruction
        load_tap(0, av[0], bv[0]);
        if (MIX) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            if (tap + 1 < KK) load_tap(tap + 1, av[(tap + 1) & 1], bv[(tap + 1) & 1]);
            if (!MIX) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap & 1][cg][t], bv[tap & 1][cg][n], acc[t][n], 0, 0, 0);
            if (MIX && tap + 1 < KK) nf_interleave<CG * 2 * NT, CG * (2 + NT)>();
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: lane holds, per tile and channel co = 16 n + li, the pixels 4 lk .. 4 lk + 3 of the tile
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n * 16 + li;
            const float b = b_lds[co];
            float* oc = out + (size_t)(s * L.Cout + co) * L.out_plane;
            if (POOL == 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int p = 4 * lk + j;
                        const int row = wide ? 2 * uu + t : 4 * uu + 2 * t + (p >> 3);
                        const int col = wide ? p : (p & 7);
                        oc[(row + oh) * L.out_wp + col + oh] = act_apply(acc[t][n][j] + b, L.act, L.alpha);
                    }
            } else {
                // windows: slot 0 (r, c), 1 (r, c + 1), 2 (r + 1, c), 3 (r + 1, c + 1); first maximum wins
                auto emit = [&](float v0, float v1, float v2, float v3, int ty, int tx) {
                    float best = v0;
                    uint32_t bi = 0;
                    if (v1 > best) { best = v1; bi = 1; }
                    if (v2 > best) { best = v2; bi = 2; }
                    if (v3 > best) { best = v3; bi = 3; }
                    const float v = act_apply(best + b, L.act, L.alpha);
                    if (L.act == SV_ACT_RELU && v <= 0.f) bi = 0;
                    oc[(ty + oh) * L.out_wp + tx + oh] = v;
                    idx_lds[((size_t)s * T + ty * tiles_x + tx) * L.Cout + co] = (uint8_t)bi;
                };
                if (wide) {  // tile 0 = row 2 uu, tile 1 = row 2 uu + 1; this lane's columns 4 lk .. 4 lk + 3
                    emit(acc[0][n][0], acc[0][n][1], acc[1][n][0], acc[1][n][1], uu, 2 * lk);
                    emit(acc[0][n][2], acc[0][n][3], acc[1][n][2], acc[1][n][3], uu, 2 * lk + 1);
                } else {     // a tile = rows (r, r + 1) x 8 columns: lanes lk < 2 hold row r, lanes lk >= 2 row r + 1
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        float o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = __shfl_xor(acc[t][n][j], 32, 64);
                        if (lk < 2) {
                            emit(acc[t][n][0], acc[t][n][1], o[0], o[1], 2 * uu + t, 2 * lk);
                            emit(acc[t][n][2], acc[t][n][3], o[2], o[3], 2 * uu + t, 2 * lk + 1);
                        }
                    }
                }
            }
        }
    }
}

// ---- a 3x3 conv behind a 2x nearest upsampling on the matrix cores, LDS -> LDS.  The four output pixels (2r + a, 2c + b)
// above a stored pixel (r, c) read the SAME 3 x 3 stored neighbourhood, each through its own 2 x 2 subset of it with
// sums of the original taps as weights (rows: a = 0 reads stored rows r - 1 with tap 0 and r with taps 1 + 2; a = 1 reads
// r with taps 0 + 1 and r + 1 with tap 2; columns alike).  So the layer is one GEMM over STORED pixels:
//   D[stored pixel][(phase, co)] = sum_{stored tap, ci} x[pixel + tap][ci] * Wp[(tap, ci)][(phase, co)],
// a quarter of the A-operand reads and, for 8 output channels, full 16-wide tiles (4 phases x 8 channels = 32 columns)
// where the direct form leaves half of each tile empty.  Wp is built once per layer and workgroup from the nine taps
// of each (ci, co) filter pair; its LDS layout [tap][ci / 4][column / 16][ci % 4][column % 16] makes a wave's B read
// 64 consecutive floats at lane + a compile-time offset.
template <int CIN, int COUT>
struct NetPhase {
    static constexpr int NT = (4 * COUT + 15) / 16, CG = CIN / 4, W_FLOATS = 9 * CG * NT * 64;
};
__host__ __device__ constexpr int net_phase_tap(int a, int k) { return floordiv_c(a + k - 1, 2) + 1; }  // stored tap 0..2

// thread t < Cin * Cout keeps the nine taps of filter pair (ci, co) = (t / Cout, t % Cout): w[tap][ci][co]
__device__ __forceinline__ void net_wp_load(const NetFwdArgs& a, const NetLayer& L, float4 (&wpre)[kNetWPre], float& bpre) {
    const int pairs = L.Cin * L.Cout;
    const float* w = a.params + L.w_off + min((int)threadIdx.x, pairs - 1);
    wpre[0] = make_float4(w[0], w[pairs], w[2 * pairs], w[3 * pairs]);
    wpre[1] = make_float4(w[4 * pairs], w[5 * pairs], w[6 * pairs], w[7 * pairs]);
    wpre[2] = make_float4(w[8 * pairs], 0.f, 0.f, 0.f);
    bpre = (L.b_off >= 0 && (int)threadIdx.x < L.Cout) ? a.params[L.b_off + threadIdx.x] : 0.f;
}

template <int CIN, int COUT>
__device__ __forceinline__ void net_wp_store(const float4 (&wpre)[kNetWPre], float bpre, float* w_lds) {
    typedef NetPhase<CIN, COUT> G;
    if ((int)threadIdx.x < CIN * COUT) {
        const int ci = threadIdx.x / COUT, co = threadIdx.x % COUT;
        const float r[9] = {wpre[0].x, wpre[0].y, wpre[0].z, wpre[0].w, wpre[1].x, wpre[1].y, wpre[1].z, wpre[1].w, wpre[2].x};
        float wp[4][9];
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int t = 0; t < 9; ++t) wp[ph][t] = 0.f;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) wp[ph][net_phase_tap(ph >> 1, kh) * 3 + net_phase_tap(ph & 1, kw)] += r[kh * 3 + kw];
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int col = ph * COUT + co;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                w_lds[(((t * G::CG + (ci >> 2)) * G::NT + (col >> 4)) * 4 + (ci & 3)) * 16 + (col & 15)] = wp[ph][t];
        }
    }
    if (4 * COUT < 16 * G::NT) {  // columns past the last phase: zero, so their (dropped) sums stay finite
        constexpr int spare = 16 * G::NT - 4 * COUT;
        for (int u = threadIdx.x; u < 9 * G::CG * 4 * spare; u += blockDim.x) {
            const int row = u / spare, col = 4 * COUT + u % spare;  // row = (tap * CG + cg) * 4 + ci % 4
            w_lds[(((row >> 2) * G::NT + (col >> 4)) * 4 + (row & 3)) * 16 + (col & 15)] = 0.f;
        }
    }
    if ((int)threadIdx.x < COUT) w_lds[G::W_FLOATS + threadIdx.x] = bpre;
}

template <int CIN, int COUT>
__device__ __forceinline__ void net_layer_fwd_mfma_up2(const NetFwdArgs& a, const NetLayer& L, float* smem, int ns, int nt) {
    typedef NetPhase<CIN, COUT> G;
    constexpr int KK = 9, NT = G::NT, CG = G::CG;
    const bool wide = L.Ws == 16;  // else 8 stored columns
    const int units_ps = (L.Hs * L.Ws) >> 5;
    const int n_units = ns * units_ps;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nt >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const float* in = smem + L.in_off;
    float* out = smem + L.out_off;
    const float* w_lane = smem + a.w_off + lane;
    const float* b_lds = smem + a.w_off + G::W_FLOATS;
    const int oh = L.out_halo, ih = L.in_halo;
    const FDiv f_ups = fd_make(units_ps);
    for (int u = wave; u < n_units; u += nw) {
        const int s = fd(u, f_ups), uu = u - s * units_ps;
        int abase[2];  // LDS offset of stored pixel (row li of tile t) inside a plane, tap (0, 0)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int pr = wide ? 2 * uu + t : 4 * uu + 2 * t + (li >> 3);
            const int pc = wide ? li : (li & 7);
            abase[t] = (pr + ih) * L.in_wp + pc + ih;
        }
        nf_v4f acc[2][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = nf_v4f{0.f, 0.f, 0.f, 0.f};
        const float* ins = in + ((size_t)s * CIN + lk) * L.in_plane;
        float av[2][CG][2], bv[2][CG][NT];
        auto load_tap = [&](int tap, float (&a_)[CG][2], float (&b_)[CG][NT]) {
            const int d = (tap / 3 - 1) * L.in_wp + (tap % 3 - 1);
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
#pragma unroll
                for (int t = 0; t < 2; ++t) a_[cg][t] = ins[cg * 4 * L.in_plane + abase[t] + d];
#pragma unroll
                for (int n = 0; n < NT; ++n) b_[cg][n] = w_lane[((tap * CG + cg) * NT + n) * 64];
            }
        };
        constexpr bool MIX = CG * (2 + NT) <= 2 * CG * 2 * NT;  // at most two reads per matrix instruction
        load_tap(0, av[0], bv[0]);
        if (MIX) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            if (tap + 1 < KK) load_tap(tap + 1, av[(tap + 1) & 1], bv[(tap + 1) & 1]);
            if (!MIX) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap & 1][cg][t], bv[tap & 1][cg][n], acc[t][n], 0, 0, 0);
            if (MIX && tap + 1 < KK) nf_interleave<CG * 2 * NT, CG * (2 + NT)>();
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: the lane holds column 16 n + li = (phase, co) of stored pixels 4 lk .. 4 lk + 3 of each tile
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int col = 16 * n + li;
            if (col < 4 * COUT) {
                const int ph = col / COUT, co = col - ph * COUT;
                const float b = b_lds[co];
                float* oc = out + (size_t)(s * L.Cout + co) * L.out_plane + ((ph >> 1) + oh) * L.out_wp + (ph & 1) + oh;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int p = 4 * lk + j;
                        const int row = wide ? 2 * uu + t : 4 * uu + 2 * t + (p >> 3);
                        const int c = wide ? p : (p & 7);
                        oc[2 * row * L.out_wp + 2 * c] = act_apply(acc[t][n][j] + b, L.act, L.alpha);
                    }
            }
        }
    }
}

// output arena (planar) -> global NHWC, argmax bytes -> global.  LOSS: the values are also compared with the targets
// (squared error summed into lacc, its gradient written) on their way out.
template <bool LOSS>
__device__ __forceinline__ void net_copy_out(const NetFwdArgs& a, const NetLayer& L, const float* smem, int n0, int ns,
                                             int nt, float& lacc) {
    const int Ho = L.Hs * L.up, Wo = L.Ws * L.up;
    const int Hout = L.pool ? Ho >> 1 : Ho, Wout = L.pool ? Wo >> 1 : Wo;
    const int P = Hout * Wout;
    const float* out = smem + L.out_off;
    const int oh = L.out_halo;
    const FDiv fP = fd_make(P), fW = fd_make(Wout);
    constexpr int U = 2;  // units per batch: their LDS reads (and target loads) are all issued before the first use
    auto emit = [&](size_t gi, const float4& v, const float4& t) {
        *reinterpret_cast<float4*>(L.y + gi) = v;
        if (LOSS) {
            const float4 d = make_float4(v.x - t.x, v.y - t.y, v.z - t.z, v.w - t.w);
            lacc += d.x * d.x;
            lacc += d.y * d.y;
            lacc += d.z * d.z;
            lacc += d.w * d.w;
            *reinterpret_cast<float4*>(a.loss_dpred + gi) =
                make_float4(a.loss_gscale * d.x, a.loss_gscale * d.y, a.loss_gscale * d.z, a.loss_gscale * d.w);
        }
    };
    if ((L.Cout & 3) == 0) {
        const int c4n = L.Cout >> 2;
        const int units = ns * P * c4n;
        const FDiv f4 = fd_make(c4n);
        for (int u0 = threadIdx.x; u0 < units; u0 += U * nt) {
            float4 v[U], t[U];
            size_t gi[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int u = min(u0 + i * nt, units - 1);
                const int r = fd(u, f4), c4 = u - r * c4n;
                const int s = fd(r, fP), p = r - s * P;
                const int row = fd(p, fW), col = p - row * Wout;
                const float* q = out + (size_t)(s * L.Cout + 4 * c4) * L.out_plane + (row + oh) * L.out_wp + col + oh;
                gi[i] = ((size_t)n0 * P + r) * L.Cout + 4 * c4;
                if (LOSS) t[i] = *reinterpret_cast<const float4*>(a.loss_t + gi[i]);
                v[i] = make_float4(q[0], q[L.out_plane], q[2 * L.out_plane], q[3 * L.out_plane]);
            }
#pragma unroll
            for (int i = 0; i < U; ++i)
                if (u0 + i * nt < units) emit(gi[i], v[i], t[i]);
        }
        if (L.pool && L.idx != nullptr) {
            const uint32_t* ib = reinterpret_cast<const uint32_t*>(smem + L.idx_off);
#pragma unroll 4
            for (int u = threadIdx.x; u < units; u += nt)
                *reinterpret_cast<uint32_t*>(L.idx + ((size_t)n0 * P) * L.Cout + 4 * (size_t)u) = ib[u];
        }
    } else if (((P * L.Cout) & 3) == 0) {
        // 16-byte pieces of the contiguous NHWC range of these samples; a piece's four floats may belong to two pixels
        const int units = (ns * P * L.Cout) >> 2;
        const FDiv fC = fd_make(L.Cout);
        for (int u0 = threadIdx.x; u0 < units; u0 += U * nt) {
            float4 v[U], t[U];
            size_t gi[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int u = min(u0 + i * nt, units - 1);
                gi[i] = (size_t)n0 * P * L.Cout + 4 * (size_t)u;
                if (LOSS) t[i] = *reinterpret_cast<const float4*>(a.loss_t + gi[i]);
                float e4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = 4 * u + k;
                    const int r = fd(e, fC), c = e - r * L.Cout;
                    const int s = fd(r, fP), p = r - s * P;
                    const int row = fd(p, fW), col = p - row * Wout;
                    e4[k] = out[(size_t)(s * L.Cout + c) * L.out_plane + (row + oh) * L.out_wp + col + oh];
                }
                v[i] = make_float4(e4[0], e4[1], e4[2], e4[3]);
            }
#pragma unroll
            for (int i = 0; i < U; ++i)
                if (u0 + i * nt < units) emit(gi[i], v[i], t[i]);
        }
        if (L.pool && L.idx != nullptr) {
            const uint32_t* ib = reinterpret_cast<const uint32_t*>(smem + L.idx_off);
#pragma unroll 4
            for (int u = threadIdx.x; u < units; u += nt)
                *reinterpret_cast<uint32_t*>(L.idx + (size_t)n0 * P * L.Cout + 4 * (size_t)u) = ib[u];
        }
    } else {
        const int units = ns * P * L.Cout;
        const FDiv fC = fd_make(L.Cout);
        for (int u = threadIdx.x; u < units; u += nt) {
            const int r = fd(u, fC), c = u - r * L.Cout;
            const int s = fd(r, fP), p = r - s * P;
            const int row = fd(p, fW), col = p - row * Wout;
            const float v = out[(size_t)(s * L.Cout + c) * L.out_plane + (row + oh) * L.out_wp + col + oh];
            const size_t gi = (size_t)n0 * P * L.Cout + u;
            L.y[gi] = v;
            if (LOSS) {
                const float d = v - a.loss_t[gi];
                lacc += d * d;
                a.loss_dpred[gi] = a.loss_gscale * d;
            }
        }
        if (L.pool && L.idx != nullptr) {
            const uint8_t* ib = reinterpret_cast<const uint8_t*>(smem + L.idx_off);
            for (int u = threadIdx.x; u < units; u += nt) L.idx[(size_t)n0 * P * L.Cout + u] = ib[u];
        }
    }
}

__device__ __forceinline__ void net_zero(float* arena, int floats, int nt) {
    float4* z = reinterpret_cast<float4*>(arena);
    for (int i = threadIdx.x; i < (floats >> 2); i += nt) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int I, int N, typename F>
__device__ __forceinline__ void net_static_for(F& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        net_static_for<I + 1, N>(f);
    }
}

// The pass loop of net_fwd_s1 (below) for a layer table known at compile time: the layer loop is unrolled, layer(l)
// returns a table row whose extents and LDS layout are constants, and every index division, layer dispatch and filter
// layout folds to the one case the row names.  256 threads, exactly ceil(N / S) workgroups.
template <int K, int NL, typename LayerOf>
__device__ __forceinline__ void net_fwd_run(const NetFwdArgs& a, LayerOf layer, float* smem) {
    static_assert(NL > 0, "compile-time layer tables only");
    constexpr int nt = 256;
    float* w_lds = smem + a.w_off;
    float4 wpre[kNetWPre];
    float bpre;
    float lacc = 0.f;
    NET_MARK(0);
    // one pass per workgroup (the launch has ceil(N / S) of them): with a pass LOOP around constant layer extents the
    // compiler hoists every layer's per-lane index arithmetic in front of it and spills what it hoisted
    {
        const int n0 = blockIdx.x * a.S;
        const int ns = min(a.S, a.N - n0);
        {
            const NetLayer& L = layer(0);
            net_w_load<K>(a, L, wpre, bpre, nt);
            NET_MARK(23);
            PlanarSrc src;
            src.x = a.x; src.zy = nullptr; src.zidx = nullptr; src.mode = 0; src.zk = ActGradCoef{};
            src.Hs = L.Hs; src.Ws = L.Ws; src.C = L.Cin;
            const PlanarDst dst{L.Cin, L.in_plane, L.in_wp, L.in_halo};
            const bool rows = a.rows.row_index != nullptr;
            float4 xin[8];  // the gathered input on its way in while the arenas are cleared
            if (rows) net_rows_load<8>(a, src, n0, ns, nt, xin);
            __syncthreads();  // the previous pass is done with every arena
            net_zero(smem + L.in_off, a.S * L.Cin * L.in_plane, nt);
            net_zero(smem + L.out_off, a.S * L.Cout * L.out_plane, nt);
            __syncthreads();
            NET_MARK(24);
            // 8 pieces per thread in flight: the 32 x 32 x 3 input of two samples arrives in one global round trip
            if (rows) net_rows_store<8>(a, src, dst, n0, ns, smem + L.in_off, nt, xin);
            else stage_pass_m<0, 8>(src, dst, make_stage_job(src, n0, ns, 0, L.Cin), smem + L.in_off, nt);
            NET_MARK(25);
            net_w_store<K>(a, L, wpre, bpre, w_lds, nt);
            NET_MARK(26);
            __syncthreads();
        }
        NET_MARK(1);
        auto step = [&](auto lc) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
            const NetLayer L = layer(l);
            const bool more = l + 1 < a.n_layers;
            // the layer's filters are in the filter arena and its output arena is zero (set up by the previous step)
            NET_MARK(2 + 3 * l);
            // [B]
            if (more) {
                const NetLayer Ln = layer(l + 1);
                if (Ln.mfma >= 4) net_wp_load(a, Ln, wpre, bpre);
                else net_w_load<K>(a, Ln, wpre, bpre, nt);
            }
            // [C]
            if (L.mfma == 1) net_layer_fwd_mfma<8, 1, 1, 1>(a, L, smem, ns, nt);
            else if (L.mfma == 2) net_layer_fwd_mfma<16, 2, 1, 1>(a, L, smem, ns, nt);
            else if (L.mfma == 3) net_layer_fwd_mfma<32, 1, 2, 0>(a, L, smem, ns, nt);
            else if (L.mfma == 4) net_layer_fwd_mfma_up2<16, 8>(a, L, smem, ns, nt);
            else if (L.mfma == 5) net_layer_fwd_mfma_up2<8, 3>(a, L, smem, ns, nt);
            else net_layer_fwd_any<K>(a, L, smem, ns, nt);
            __syncthreads();
            NET_MARK(3 + 3 * l);
            // [A] of the next layer: its filters (loaded a whole layer ago) into the filter arena, its output arena (this
            // layer's input arena) zeroed; both are free now
            if (more) {
                const NetLayer Ln = layer(l + 1);
                if (Ln.mfma == 4) net_wp_store<16, 8>(wpre, bpre, w_lds);
                else if (Ln.mfma == 5) net_wp_store<8, 3>(wpre, bpre, w_lds);
                else net_w_store<K>(a, Ln, wpre, bpre, w_lds, nt);
                net_zero(smem + Ln.out_off, a.S * Ln.Cout * Ln.out_plane, nt);
                __syncthreads();
            }
            if (l == 1) NET_MARK(20);
            // [E] no barrier after it: the copy only reads what the next layer's arithmetic also only reads (the argmax
            // bytes of consecutive layers live in different arenas), so a wave goes from its last store straight on
            if (a.loss_t != nullptr && !more) net_copy_out<true>(a, L, smem, n0, ns, nt, lacc);
            else net_copy_out<false>(a, L, smem, n0, ns, nt, lacc);
            if (l == 1) NET_MARK(21);
            NET_MARK(4 + 3 * l);
        };
        net_static_for<0, NL>(step);
    }
    if (a.loss_t != nullptr) {  // fixed-order sum over the workgroup: lanes of a wave, then the waves in index order
        __shared__ float lred[16];
        for (int o = 32; o > 0; o >>= 1) lacc += __shfl_down(lacc, o);
        if ((threadIdx.x & 63) == 0) lred[threadIdx.x >> 6] = lacc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int w = 0; w < (nt >> 6); ++w) t += lred[w];
            a.loss_partials[blockIdx.x] = t;
        }
    }
}

template <int K>
__global__ __launch_bounds__(512) void net_fwd_s1(NetFwdArgs args_by_value) {
    // the layer table is indexed at run time: read it where the launch put it (the kernel-argument segment, through the
    // scalar cache) instead of letting the compiler copy the by-value struct to per-lane scratch memory
    (void)args_by_value;
    const NetFwdArgs& a = *(const NetFwdArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nt = blockDim.x;
    float* w_lds = smem + a.w_off;
    float4 wpre[kNetWPre];
    float bpre;
    float lacc = 0.f;
    NET_MARK(0);
    for (int n0 = blockIdx.x * a.S; n0 < a.N; n0 += gridDim.x * a.S) {
        const int ns = min(a.S, a.N - n0);
        {
            const NetLayer& L = a.L[0];
            net_w_load<K>(a, L, wpre, bpre, nt);
            NET_MARK(23);
            __syncthreads();  // the previous pass is done with every arena
            net_zero(smem + L.in_off, a.S * L.Cin * L.in_plane, nt);
            net_zero(smem + L.out_off, a.S * L.Cout * L.out_plane, nt);
            __syncthreads();
            NET_MARK(24);
            PlanarSrc src;
            src.x = a.x; src.zy = nullptr; src.zidx = nullptr; src.mode = 0; src.zk = ActGradCoef{};
            src.Hs = L.Hs; src.Ws = L.Ws; src.C = L.Cin;
            const PlanarDst dst{L.Cin, L.in_plane, L.in_wp, L.in_halo};
            // 8 pieces per thread in flight: the 32 x 32 x 3 input of two samples arrives in one global round trip
            stage_pass_m<0, 8>(src, dst, make_stage_job(src, n0, ns, 0, L.Cin), smem + L.in_off, nt);
            NET_MARK(25);
            net_w_store<K>(a, L, wpre, bpre, w_lds, nt);
            NET_MARK(26);
            __syncthreads();
        }
        NET_MARK(1);
        for (int l = 0; l < a.n_layers; ++l) {
            const NetLayer& L = a.L[l];
            // the layer's filters are in the filter arena and its output arena is zero (set up by the previous step)
            NET_MARK(2 + 3 * l);
            // [B]
            if (l + 1 < a.n_layers) {
                if (a.L[l + 1].mfma >= 4) net_wp_load(a, a.L[l + 1], wpre, bpre);
                else net_w_load<K>(a, a.L[l + 1], wpre, bpre, nt);
            }
            // [C]
            if (L.mfma == 1) net_layer_fwd_mfma<8, 1, 1, 1>(a, L, smem, ns, nt);
            else if (L.mfma == 2) net_layer_fwd_mfma<16, 2, 1, 1>(a, L, smem, ns, nt);
            else if (L.mfma == 3) net_layer_fwd_mfma<32, 1, 2, 0>(a, L, smem, ns, nt);
            else if (L.mfma == 4) net_layer_fwd_mfma_up2<16, 8>(a, L, smem, ns, nt);
            else if (L.mfma == 5) net_layer_fwd_mfma_up2<8, 3>(a, L, smem, ns, nt);
            else net_layer_fwd_any<K>(a, L, smem, ns, nt);
            __syncthreads();
            NET_MARK(3 + 3 * l);
            // [A] of the next layer: its filters (loaded a whole layer ago) into the filter arena, its output arena (this
            // layer's input arena) zeroed; both are free now
            if (l + 1 < a.n_layers) {
                if (a.L[l + 1].mfma == 4) net_wp_store<16, 8>(wpre, bpre, w_lds);
                else if (a.L[l + 1].mfma == 5) net_wp_store<8, 3>(wpre, bpre, w_lds);
                else net_w_store<K>(a, a.L[l + 1], wpre, bpre, w_lds, nt);
                net_zero(smem + a.L[l + 1].out_off, a.S * a.L[l + 1].Cout * a.L[l + 1].out_plane, nt);
                __syncthreads();
            }
            if (l == 1) NET_MARK(20);
            // [E] no barrier after it: the copy only reads what the next layer's arithmetic also only reads (the argmax
            // bytes of consecutive layers live in different arenas), so a wave goes from its last store straight on
            if (a.loss_t != nullptr && l + 1 == a.n_layers) net_copy_out<true>(a, L, smem, n0, ns, nt, lacc);
            else net_copy_out<false>(a, L, smem, n0, ns, nt, lacc);
            if (l == 1) NET_MARK(21);
            NET_MARK(4 + 3 * l);
        }
    }
    if (a.loss_t != nullptr) {  // fixed-order sum over the workgroup: lanes of a wave, then the waves in index order
        __shared__ float lred[16];
        for (int o = 32; o > 0; o >>= 1) lacc += __shfl_down(lacc, o);
        if ((threadIdx.x & 63) == 0) lred[threadIdx.x >> 6] = lacc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int w = 0; w < (nt >> 6); ++w) t += lred[w];
            a.loss_partials[blockIdx.x] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------------ host side
struct NetPlan {
    bool ok;
    int S, nt, grid;
    size_t lds_bytes;
    NetFwdArgs args;
};

constexpr size_t kNetLdsPerBlock = 78 * 1024;  // two workgroups share a CU's 160 KiB

// Extents, LDS layout and kernel variants of every layer for S samples per workgroup of nt threads: everything in the
// table that does not depend on pointers or parameter offsets.  constexpr: the same function lays out the table the
// compile-time kernel (net_fwd_cae) is built for.
struct NetLayoutFlags {
    bool no_pair, no_phase, valu;
};
constexpr size_t net_layout(NetFwdArgs& a, const sv_net_layer* layers, int n_layers, int H, int W, int S, int nt,
                            NetLayoutFlags f) {
    // activation i is the input of layer i (i = n_layers: the network output); it lives in arena i % 2
    size_t arena[2] = {0, 0};
    size_t w_floats = 0, idx_bytes[2] = {0, 0};
    int h = H, w = W;
    for (int l = 0; l <= n_layers; ++l) {
        int C = 0, plane = 0, wp = 0, halo = 0;
        if (l < n_layers) {
            C = layers[l].Cin; halo = 1;
            wp = (w + 2 * halo + 3) & ~3;
            plane = (h + 2 * halo) * wp + 4;
        } else {
            C = layers[n_layers - 1].Cout; halo = 0;
            wp = w;
            plane = (h * w + 3) & ~3;
        }
        const size_t need = (size_t)S * C * plane;
        if (need > arena[l & 1]) arena[l & 1] = need;
        if (l > 0) {
            NetLayer& P = a.L[l - 1];
            P.out_plane = plane; P.out_wp = wp; P.out_halo = halo;
        }
        if (l == n_layers) break;
        const sv_net_layer& s = layers[l];
        NetLayer& L = a.L[l];
        L.Cin = s.Cin; L.Cout = s.Cout; L.Hs = h; L.Ws = w; L.up = s.up; L.pool = s.pool; L.act = s.act; L.alpha = s.alpha;
        L.in_plane = plane; L.in_wp = wp; L.in_halo = halo;
        L.co_t = s.Cout > 4 ? 8 : 4;
        L.n_cot = (s.Cout + L.co_t - 1) / L.co_t;
        const int Ho = h * s.up, Wo = w * s.up;
        const int T = (Ho / 2) * (Wo / 2);
        L.pair = 0;
        if (S % 2 == 0 && !f.no_pair) {
            if (s.up == 1 && L.co_t == 8 && s.pool) L.pair = 1;
        }
        int ks = 1;
        while (ks < 8 && s.Cin / (2 * ks) >= 2 && (int64_t)L.n_cot * (L.pair ? S / 2 : S) * T * ks < nt) ks <<= 1;
        L.ks = ks;
        // matrix-core variants exist for the shapes whose 16-wide tiles are full: (Cin, Cout, up, pool)
        L.mfma = 0;
        if ((Wo == 8 || Wo == 16) && (Ho * Wo) % 32 == 0 && (nt & 63) == 0 && l + 1 < n_layers && !f.valu) {
            if (s.Cin == 8 && s.Cout == 16 && s.up == 1 && s.pool) L.mfma = 1;
            else if (s.Cin == 16 && s.Cout == 32 && s.up == 1 && s.pool) L.mfma = 2;
            else if (s.Cin == 32 && s.Cout == 16 && s.up == 2 && !s.pool) L.mfma = 3;
        }
        // an upsampled input as one GEMM over the stored pixels (net_layer_fwd_mfma_up2); never the first layer, whose
        // filters take the generic path into LDS
        if (L.mfma == 0 && l > 0 && s.up == 2 && !s.pool && (w == 8 || w == 16) && (h * w) % 32 == 0 && (nt & 63) == 0 &&
            s.Cin * s.Cout <= nt && !f.no_phase && !f.valu) {
            if (s.Cin == 16 && s.Cout == 8) L.mfma = 4;
            else if (s.Cin == 8 && s.Cout == 3) L.mfma = 5;
        }
        size_t wf = (size_t)9 * s.Cin * L.n_cot * L.co_t + L.n_cot * L.co_t;
        if (L.mfma == 4) wf = NetPhase<16, 8>::W_FLOATS + 8;
        else if (L.mfma == 5) wf = NetPhase<8, 3>::W_FLOATS + 4;
        if (wf > w_floats) w_floats = wf;
        if (s.pool) {
            const size_t ib = (size_t)S * T * s.Cout;
            if (ib > idx_bytes[l & 1]) idx_bytes[l & 1] = ib;
        }
        h = s.pool ? Ho / 2 : Ho;
        w = s.pool ? Wo / 2 : Wo;
    }
    const size_t o0 = 0, o1 = (arena[0] + 3) & ~(size_t)3, ow = o1 + ((arena[1] + 3) & ~(size_t)3);
    const size_t oi = ow + ((w_floats + 3) & ~(size_t)3);
    const size_t oi1 = oi + (((idx_bytes[0] + 3) / 4 + 3) & ~(size_t)3);
    for (int l = 0; l < n_layers; ++l) {
        a.L[l].in_off = (int)((l & 1) ? o1 : o0);
        a.L[l].out_off = (int)((l & 1) ? o0 : o1);
        a.L[l].idx_off = (int)((l & 1) ? oi1 : oi);
    }
    a.w_off = (int)ow;
    a.S = S;
    a.n_layers = n_layers;
    return oi1 + (((idx_bytes[1] + 3) / 4 + 3) & ~(size_t)3);
}

static int net_fill(NetFwdArgs& a, const sv_net_layer* layers, int n_layers, int N, int H, int W, int S, int nt,
                    size_t* lds_floats) {
    // the options are read at every plan, so a process can compare the variants
    const NetLayoutFlags flags{opt(OPT_NETFWD_PAIR) == 0, opt(OPT_NETFWD_PHASE) == 0, opt(OPT_NETFWD_VALU) == 1};
    *lds_floats = net_layout(a, layers, n_layers, H, W, S, nt, flags);
    for (int l = 0; l < n_layers; ++l) {
        const sv_net_layer& s = layers[l];
        NetLayer& L = a.L[l];
        L.w_off = (int)s.w_off; L.b_off = (int)s.b_off;
        L.y = (float*)s.y; L.idx = s.idx;
    }
    return 0;
}

// ---- the forward launch built for ONE layer table: the conv stack of CAEPadded at the run_model.py defaults
// (autoencoder.py:437-526 with global_vars / run_model.py:80-104: 3 -> 8 -> 16 -> 32 pooled, 32 -> 16 -> 8 -> 3
// upsampled, 32 x 32 inputs, ELU / tanh), two samples per workgroup of 256 threads.  Only pointers, parameter offsets
// and the batch size are launch arguments; a launch takes this kernel when the table net_fill computes for it equals
// the compile-time one field by field, and the run-time-table kernel otherwise.
constexpr int kCaeLayers = 6;
constexpr sv_net_layer kCaeChain[kCaeLayers] = {  // ELU (alpha 1) inside, tanh on the output: the activations are constants too
    {3, 8, 3, 1, 1, SV_ACT_ELU, 1.f, 0, 0, 0, nullptr, nullptr},   {8, 16, 3, 1, 1, SV_ACT_ELU, 1.f, 0, 0, 0, nullptr, nullptr},
    {16, 32, 3, 1, 1, SV_ACT_ELU, 1.f, 0, 0, 0, nullptr, nullptr}, {32, 16, 3, 2, 0, SV_ACT_ELU, 1.f, 0, 0, 0, nullptr, nullptr},
    {16, 8, 3, 2, 0, SV_ACT_ELU, 1.f, 0, 0, 0, nullptr, nullptr},  {8, 3, 3, 2, 0, SV_ACT_TANH, 1.f, 0, 0, 0, nullptr, nullptr}};
struct NetCaeTable {
    NetFwdArgs a;
    size_t lds_floats;
};
constexpr NetCaeTable net_cae_table() {
    NetCaeTable t{};
    t.lds_floats = net_layout(t.a, kCaeChain, kCaeLayers, 32, 32, 2, 256, NetLayoutFlags{false, false, false});
    return t;
}
constexpr NetCaeTable kCae = net_cae_table();
static_assert(kCae.lds_floats * sizeof(float) <= 78 * 1024, "two workgroups per CU");

struct NetCaeLayerArgs {
    int w_off, b_off;
    float* y;
    uint8_t* idx;
};
struct NetCaeArgs {
    const float* x;
    const float* params;
    int N;
    const float* loss_t;
    float* loss_dpred;
    float* loss_partials;
    float loss_gscale;
    sv_batch_rows rows;
    NetCaeLayerArgs rt[kCaeLayers];
};

__global__ __launch_bounds__(256, 2) void net_fwd_cae(NetCaeArgs r) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    NetFwdArgs a{};
    a.x = r.x; a.params = r.params; a.N = r.N; a.S = kCae.a.S; a.n_layers = kCaeLayers; a.w_off = kCae.a.w_off;
    a.loss_t = r.loss_t; a.loss_dpred = r.loss_dpred; a.loss_partials = r.loss_partials; a.loss_gscale = r.loss_gscale;
    a.rows = r.rows;
    net_fwd_run<3, kCaeLayers>(a, [&](int l) __attribute__((always_inline)) {
        NetLayer L = kCae.a.L[l];
        L.w_off = r.rt[l].w_off; L.b_off = r.rt[l].b_off;
        L.y = r.rt[l].y; L.idx = r.rt[l].idx;
        return L;
    }, smem);
}

// The same forward pass followed, in the same launch, by the input-gradient chain of the backward pass: dL/dy of the
// squared error leaves with the last layer's output, and every tensor a sample's gradient chain reads (that gradient,
// the layers' outputs and argmax bytes) was written by the workgroup that carries the sample, so it goes straight on
// instead of waiting for the slowest workgroup of the forward launch and for the next launch to ramp up.
struct NetCaeTrainArgs {
    NetCaeArgs r;
    DnChainArgs c;  // S == the forward's samples per workgroup
};

__global__ __launch_bounds__(256, 2) void net_fwd_dgrad_cae(NetCaeTrainArgs t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const NetCaeArgs& r = t.r;
    NetFwdArgs a{};
    a.x = r.x; a.params = r.params; a.N = r.N; a.S = kCae.a.S; a.n_layers = kCaeLayers; a.w_off = kCae.a.w_off;
    a.loss_t = r.loss_t; a.loss_dpred = r.loss_dpred; a.loss_partials = r.loss_partials; a.loss_gscale = r.loss_gscale;
    a.rows = r.rows;
    net_fwd_run<3, kCaeLayers>(a, [&](int l) __attribute__((always_inline)) {
        NetLayer L = kCae.a.L[l];
        L.w_off = r.rt[l].w_off; L.b_off = r.rt[l].b_off;
        L.y = r.rt[l].y; L.idx = r.rt[l].idx;
        return L;
    }, smem);
#if defined(__HIP_DEVICE_COMPILE__)
    // the table is indexed by a loop counter: read it where the launch put it (scalar loads), not through the parameter
    const DnChainArgs& c = *(const DnChainArgs*)((const char*)(const void*)__builtin_amdgcn_kernarg_segment_ptr() +
                                                 offsetof(NetCaeTrainArgs, c));
    dn_chain_run<false>(c, blockIdx.x * c.S, true);
#endif
}

// true when the plan's table is the compile-time one (pointers and parameter offsets aside)
static bool net_plan_is_cae(const NetFwdArgs& p, int nt) {
    if (opt(OPT_NETFWD_GENERIC) == 1 || nt != 256 || p.n_layers != kCaeLayers || p.S != kCae.a.S || p.w_off != kCae.a.w_off) return false;
    for (int l = 0; l < kCaeLayers; ++l) {
        const NetLayer &x = p.L[l], &y = kCae.a.L[l];
        if (x.Cin != y.Cin || x.Cout != y.Cout || x.Hs != y.Hs || x.Ws != y.Ws || x.up != y.up || x.pool != y.pool ||
            x.in_off != y.in_off || x.out_off != y.out_off || x.idx_off != y.idx_off || x.in_plane != y.in_plane ||
            x.in_wp != y.in_wp || x.in_halo != y.in_halo || x.out_plane != y.out_plane || x.out_wp != y.out_wp ||
            x.out_halo != y.out_halo || x.co_t != y.co_t || x.n_cot != y.n_cot || x.ks != y.ks || x.pair != y.pair ||
            x.mfma != y.mfma || x.act != y.act)  // ELU and tanh take no parameter
            return false;
    }
    return true;
}

int net_validate(const sv_net_layer* layers, int n_layers, int N, int H, int W, const char* who) {
    if (layers == nullptr || n_layers < 1 || n_layers > kMaxNetLayers)
        return fail(SV_E_INVALID, "%s: 1..%d layers expected", who, kMaxNetLayers);
    if (N < 1 || H < 1 || W < 1) return fail(SV_E_INVALID, "%s: empty input", who);
    int h = H, w = W, c = layers[0].Cin;
    for (int l = 0; l < n_layers; ++l) {
        const sv_net_layer& s = layers[l];
        if (s.K != 3) return fail(SV_E_UNSUPPORTED, "%s: layer %d: only 3x3 kernels", who, l);
        if (s.Cin != c || s.Cin < 1 || s.Cout < 1) return fail(SV_E_INVALID, "%s: layer %d: channel mismatch", who, l);
        if (!(s.up == 1 || s.up == 2) || !(s.pool == 0 || s.pool == 1))
            return fail(SV_E_INVALID, "%s: layer %d: up must be 1|2, pool 0|1", who, l);
        const int Ho = h * s.up, Wo = w * s.up;
        if ((Ho & 1) || (Wo & 1)) return fail(SV_E_UNSUPPORTED, "%s: layer %d: odd conv extent %dx%d", who, l, Ho, Wo);
        if (s.w_off < 0 || s.w_off > INT32_MAX || s.b_off > INT32_MAX)
            return fail(SV_E_INVALID, "%s: layer %d: parameter offsets out of range", who, l);
        h = s.pool ? Ho / 2 : Ho;
        w = s.pool ? Wo / 2 : Wo;
        c = s.Cout;
    }
    return SV_OK;
}

NetPlan plan_net_fwd(const sv_net_layer* layers, int n_layers, int N, int H, int W) {
    NetPlan p{};
    p.ok = false;
    p.nt = 256;
    const int s_cap = opt(OPT_NETFWD_S) > 0 ? opt(OPT_NETFWD_S) : 4;  // cap on the samples a workgroup carries
    for (int S = 4; S >= 1; --S) {
        if (S == 3 || S > s_cap) continue;
        size_t fl = 0;
        NetFwdArgs a{};
        net_fill(a, layers, n_layers, N, H, W, S, p.nt, &fl);
        if (fl * sizeof(float) <= kNetLdsPerBlock) {
            p.S = S > N ? N : S;
            if (p.S != S) net_fill(a, layers, n_layers, N, H, W, p.S, p.nt, &fl);
            p.args = a;
            p.lds_bytes = (fl * sizeof(float) + 15) & ~(size_t)15;
            p.ok = true;
            break;
        }
    }
    if (!p.ok) return p;
    const int passes = (N + p.S - 1) / p.S;
    const int g_cap = opt(OPT_NETFWD_GRID) > 0 ? opt(OPT_NETFWD_GRID) : 512;  // cap on the workgroups of the launch
    p.grid = passes < g_cap ? passes : g_cap;
    p.args.N = N; p.args.S = p.S; p.args.n_layers = n_layers;
    return p;
}

static int net_enable_lds(size_t bytes) {
    static size_t enabled = 0;  // benign race: the attribute is idempotent
    if (bytes <= enabled) return SV_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&net_fwd_s1<3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNetLdsPerBlock);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&net_fwd_cae), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kNetLdsPerBlock);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&net_fwd_dgrad_cae),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNetLdsPerBlock);
    if (e != hipSuccess) return fail(SV_E_LAUNCH, "net_fwd: cannot raise the LDS limit: %s", hipGetErrorString(e));
    enabled = kNetLdsPerBlock;
    return SV_OK;
}

}  // namespace sv

using namespace sv;

extern "C" {

#ifdef SV_TRACE
int sv_debug_set_net_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(sv::g_net_trace), &buf, sizeof(buf)); }
#endif

int sv_convnet_supported(const sv_net_layer* layers, int n_layers, int N, int H, int W) {
    if (net_validate(layers, n_layers, N, H, W, "sv_convnet_supported")) return 0;
    return plan_net_fwd(layers, n_layers, N, H, W).ok ? 1 : 0;
}

// whether the launch can gather its batch through a row index: the kernel built for the default stack, whole 16-byte
// pieces per sample, at most four samples a pass
static bool net_rows_ok(const NetPlan& p, const sv_net_layer* layers, int H, int W) {
    return net_plan_is_cae(p.args, p.nt) && p.S <= 4 && ((H * W * layers[0].Cin) & 3) == 0 &&
           p.S * ((H * W * layers[0].Cin) >> 2) <= 8 * 256;  // the pieces of a pass fit the registers of one load batch
}

static int convnet_fwd_any(const char* who, const sv_net_layer* layers, int n_layers, int N, int H, int W, const void* x,
                           const void* params, const float* y_true, float* dpred, float loss_scale, float* partials,
                           int* n_partials, sv_stream_t stream, const sv_batch_rows* rows = nullptr,
                           const sv_dgrad_job* jobs = nullptr, int n_jobs = 0) {
    int rc = net_validate(layers, n_layers, N, H, W, who);
    if (rc) return rc;
    if (rows != nullptr) x = rows->x_batch;
    SV_REQUIRE(x && params, "%s: null tensor", who);
    for (int l = 0; l < n_layers; ++l) SV_REQUIRE(layers[l].y != nullptr, "%s: layer %d has no output buffer", who, l);
    NetPlan p = plan_net_fwd(layers, n_layers, N, H, W);
    if (!p.ok) return fail(SV_E_UNSUPPORTED, "%s: the activations of one sample do not fit the LDS arenas", who);
    rc = net_enable_lds(p.lds_bytes);
    if (rc) return rc;
    p.args.x = (const float*)x;
    p.args.params = (const float*)params;
    if (y_true != nullptr) {
        const sv_net_layer& last = layers[n_layers - 1];
        SV_REQUIRE(!last.pool, "%s: the last layer's output must leave unpooled", who);
        int h = H, w = W;
        for (int l = 0; l < n_layers; ++l) {
            h = h * layers[l].up / (layers[l].pool ? 2 : 1);
            w = w * layers[l].up / (layers[l].pool ? 2 : 1);
        }
        const double count = (double)N * h * w * last.Cout;
        p.args.loss_t = y_true;
        p.args.loss_dpred = dpred;
        p.args.loss_partials = partials;
        p.args.loss_gscale = (float)(2.0 * (double)loss_scale / count);
        *n_partials = p.grid;
    }
    // one workgroup per pass there: with the loss, the per-workgroup sums must fit the caller's partials buffer
    // (sv_loss_workspace_bytes: 1024 floats)
    if (rows != nullptr && !(net_rows_ok(p, layers, H, W) && (N + p.S - 1) / p.S <= 1024))
        return fail(SV_E_UNSUPPORTED, "%s: this layer table does not gather its batch in the launch", who);
    if (net_plan_is_cae(p.args, p.nt) && (y_true == nullptr || (N + p.S - 1) / p.S <= 1024)) {
        NetCaeArgs r{};
        if (rows != nullptr) r.rows = *rows;
        r.x = p.args.x; r.params = p.args.params; r.N = N;
        r.loss_t = p.args.loss_t; r.loss_dpred = p.args.loss_dpred; r.loss_partials = p.args.loss_partials;
        r.loss_gscale = p.args.loss_gscale;
        for (int l = 0; l < kCaeLayers; ++l) {
            const NetLayer& L = p.args.L[l];
            r.rt[l] = NetCaeLayerArgs{L.w_off, L.b_off, L.y, L.idx};
        }
        const int passes = (N + p.S - 1) / p.S;
        if (y_true != nullptr) *n_partials = passes;
        if (jobs != nullptr) {
            NetCaeTrainArgs t{};
            t.r = r;
            int grid = 0;
            size_t chain_lds = 0;
            bool carried = plan_dgrad_chain(jobs, n_jobs, p.S, &t.c, &grid, &chain_lds) == 0;
            for (int j = 0; carried && j < n_jobs; ++j) carried = dn_code_in_default_stack(t.c.code[j]);
            if (!carried || grid != passes || jobs[0].g.N != N || chain_lds > kNetLdsPerBlock)
                return fail(SV_E_UNSUPPORTED, "%s: the input-gradient jobs do not run behind this forward launch", who);
            const size_t lds = chain_lds > p.lds_bytes ? chain_lds : p.lds_bytes;
            hipLaunchKernelGGL(net_fwd_dgrad_cae, dim3((unsigned)passes), dim3(256), lds, to_stream(stream), t);
            return check_launch("net_fwd_dgrad_cae");
        }
        hipLaunchKernelGGL(net_fwd_cae, dim3((unsigned)passes), dim3(256), p.lds_bytes, to_stream(stream), r);
        return check_launch("net_fwd_cae");
    }
    if (jobs != nullptr)
        return fail(SV_E_UNSUPPORTED, "%s: this layer table has no launch that also runs input-gradient jobs", who);
    hipLaunchKernelGGL((net_fwd_s1<3>), dim3((unsigned)p.grid), dim3((unsigned)p.nt), p.lds_bytes, to_stream(stream), p.args);
    return check_launch("net_fwd_s1");
}

int sv_convnet_fwd(const sv_net_layer* layers, int n_layers, int N, int H, int W, const void* x, const void* params,
                   int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_convnet_fwd");
    return convnet_fwd_any("sv_convnet_fwd", layers, n_layers, N, H, W, x, params, nullptr, nullptr, 0.f, nullptr, nullptr,
                           stream);
}

int sv_convnet_fwd_mse(const sv_net_layer* layers, int n_layers, int N, int H, int W, const void* x, const void* params,
                       const float* y_true, float* dpred, float loss_scale, float* partials, int* n_partials, int dtype,
                       sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_convnet_fwd_mse");
    SV_REQUIRE(y_true && dpred && partials && n_partials, "sv_convnet_fwd_mse: null loss argument");
    return convnet_fwd_any("sv_convnet_fwd_mse", layers, n_layers, N, H, W, x, params, y_true, dpred, loss_scale, partials,
                           n_partials, stream);
}

int sv_convnet_rows_supported(const sv_net_layer* layers, int n_layers, int N, int H, int W) {
    if (net_validate(layers, n_layers, N, H, W, "sv_convnet_rows_supported")) return 0;
    const NetPlan p = plan_net_fwd(layers, n_layers, N, H, W);
    return p.ok && net_rows_ok(p, layers, H, W) && (N + p.S - 1) / p.S <= 1024 ? 1 : 0;
}

int sv_convnet_fwd_mse_rows(const sv_net_layer* layers, int n_layers, int N, int H, int W, const sv_batch_rows* rows,
                            const void* params, const float* y_true, float* dpred, float loss_scale, float* partials,
                            int* n_partials, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_convnet_fwd_mse_rows");
    SV_REQUIRE(y_true && dpred && partials && n_partials, "sv_convnet_fwd_mse_rows: null loss argument");
    SV_REQUIRE(rows && rows->x_rows && rows->row_index && rows->x_batch, "sv_convnet_fwd_mse_rows: null row table");
    SV_REQUIRE(rows->n_rows > 0 && rows->n_index >= N && rows->batch_stride >= 0,
               "sv_convnet_fwd_mse_rows: the index list is shorter than the batch");
    SV_REQUIRE((rows->counter == nullptr) == (rows->counter_base == nullptr),
               "sv_convnet_fwd_mse_rows: counter and counter_base go together");
    SV_REQUIRE(((uintptr_t)rows->x_rows % 16 == 0) && ((uintptr_t)rows->x_batch % 16 == 0),
               "sv_convnet_fwd_mse_rows: tensors must be 16-byte aligned");
    return convnet_fwd_any("sv_convnet_fwd_mse_rows", layers, n_layers, N, H, W, nullptr, params, y_true, dpred, loss_scale,
                           partials, n_partials, stream, rows);
}

// ---- forward + squared error + the input-gradient chain of the backward pass in one launch
static bool dgrad_jobs_set(const sv_dgrad_job* jobs, int n_jobs) {
    if (jobs == nullptr || n_jobs < 1 || n_jobs > 8) return false;
    for (int j = 0; j < n_jobs; ++j)
        if (!jobs[j].w || !jobs[j].dy || !jobs[j].y || !jobs[j].dx) return false;
    return true;
}

int sv_convnet_fwd_mse_dgrad_supported(const sv_net_layer* layers, int n_layers, int N, int H, int W,
                                       const sv_dgrad_job* jobs, int n_jobs) {
    if (net_validate(layers, n_layers, N, H, W, "sv_convnet_fwd_mse_dgrad_supported") || !dgrad_jobs_set(jobs, n_jobs)) return 0;
    const NetPlan p = plan_net_fwd(layers, n_layers, N, H, W);
    if (!p.ok || !net_plan_is_cae(p.args, p.nt) || (N + p.S - 1) / p.S > 1024) return 0;
    DnChainArgs c;
    int grid = 0;
    size_t lds = 0;
    if (plan_dgrad_chain(jobs, n_jobs, p.S, &c, &grid, &lds)) return 0;
    for (int j = 0; j < n_jobs; ++j)
        if (!dn_code_in_default_stack(c.code[j])) return 0;
    return grid == (N + p.S - 1) / p.S && jobs[0].g.N == N && lds <= kNetLdsPerBlock ? 1 : 0;
}

int sv_convnet_fwd_mse_dgrad(const sv_net_layer* layers, int n_layers, int N, int H, int W, const void* x,
                             const sv_batch_rows* rows, const void* params, const float* y_true, float* dpred,
                             float loss_scale, float* partials, int* n_partials, const sv_dgrad_job* jobs, int n_jobs,
                             int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_convnet_fwd_mse_dgrad");
    SV_REQUIRE(y_true && dpred && partials && n_partials, "sv_convnet_fwd_mse_dgrad: null loss argument");
    SV_REQUIRE(dgrad_jobs_set(jobs, n_jobs), "sv_convnet_fwd_mse_dgrad: bad job list (1..8 jobs, tensors set)");
    if (rows != nullptr) {
        SV_REQUIRE(rows->x_rows && rows->row_index && rows->x_batch, "sv_convnet_fwd_mse_dgrad: null row table");
        SV_REQUIRE(rows->n_rows > 0 && rows->n_index >= N && rows->batch_stride >= 0,
                   "sv_convnet_fwd_mse_dgrad: the index list is shorter than the batch");
        SV_REQUIRE((rows->counter == nullptr) == (rows->counter_base == nullptr),
                   "sv_convnet_fwd_mse_dgrad: counter and counter_base go together");
        SV_REQUIRE(((uintptr_t)rows->x_rows % 16 == 0) && ((uintptr_t)rows->x_batch % 16 == 0),
                   "sv_convnet_fwd_mse_dgrad: tensors must be 16-byte aligned");
    }
    return convnet_fwd_any("sv_convnet_fwd_mse_dgrad", layers, n_layers, N, H, W, x, params, y_true, dpred, loss_scale,
                           partials, n_partials, stream, rows, jobs, n_jobs);
}

}  // extern "C"
