// LDS-tiled direct convolution for stride-1 SAME convs with odd kernels (the CAEPadded family), f32.
//
// One workgroup keeps S whole samples of the input in LDS as PLANAR images with a zero halo
// ([sample][channel][row][col], rows padded to a multiple of 4 floats), plus the filter taps it needs.
// A thread owns a 2x2 block of output pixels x CO_T output channels in registers: per input channel it
// loads the (K+1)x(K+1) input patch once and reuses it for all K*K taps and all CO_T channels, so the
// inner loop is ~95 % v_fma_f32 (f32 vector rate == f32 MFMA rate on gfx950, and channel counts of 3/8/16
// would leave most of an MFMA tile empty).  Filter taps are read from LDS at a wave-uniform address
// (broadcast).  Nearest upsampling of the input (UP = 2) is folded into the patch indexing: the LDS image
// stays at the stored resolution.
//
// The same kernel computes
//   - the forward conv (+bias, +activation),                                   EPI_STORE
//   - forward conv + activation + 2x2 max-pool with argmax,                     EPI_MAXPOOL
//   - the input gradient (a conv of dz with the taps flipped and channels swapped: WFLIP),
//   - the input gradient summed over each 2x2 block (backward of a folded upsampling).  EPI_SUM4
#include <cstdlib>

#include "tile_common.h"

namespace sv {

// Phase timestamps for kernel tuning (diagnostic builds only: make TRACE=1).  Thread 0 of every workgroup stores
// the 100 MHz wall clock at up to 8 phase marks into g_sv_trace[block][8] (slot 7: hardware id).
#ifdef SV_TRACE
__device__ unsigned long long* g_sv_trace = nullptr;
__device__ __forceinline__ void trace_mark(int slot) {
    if (threadIdx.x == 0 && g_sv_trace != nullptr) {
        const size_t b = (size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
        g_sv_trace[b * 8 + slot] = wall_clock64();
        if (slot == 0) g_sv_trace[b * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
                                                 ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32);
    }
}
#define SV_MARK(slot) trace_mark(slot)
#else
#define SV_MARK(slot)
#endif

struct TiledArgs {
    const float* x;     // [N][Hs][Ws][Cin]   stored input (for WFLIP: dz)
    const float* w;     // HWIO [K][K][Cw_in][Cw_out] as stored by the layer
    const float* bias;  // [Cout] or null
    float* y;           // EPI_STORE: [N][Ho][Wo][Cout]; MAXPOOL/SUM4: [N][Ho/2][Wo/2][Cout]
    uint8_t* idx;       // MAXPOOL: argmax [N][Ho/2][Wo/2][Cout] (may be null)
    int N, Hs, Ws, Cin, Cout;  // conv input channels / output channels of THIS launch (already swapped for WFLIP)
    int Ho, Wo;                // logical output extent = UP*Hs x UP*Ws
    int wflip;                 // 1: tap (kh,kw,ci,co) is read from w[K-1-kh][K-1-kw][co][ci]  (w stored [K][K][Cout][Cin] in launch terms)
    int act;
    float alpha;
    int S;        // samples per pass
    int cic;      // input channels per LDS chunk
    int coc;      // output channels per block (grid.y chunks), multiple of CO_T
    int plane;    // floats per LDS channel plane
    int wp;       // floats per LDS row
    int halo;     // stored halo (rows/cols of zeros around the image)
    // The staged input may be a gradient formed on the fly (input-gradient launches):
    //   zmode 0: x as is;  1: x * act'(zy) elementwise;  2: x, zy, zidx live at the POOLED resolution
    //   (Hs/2 x Ws/2) and the staged value is x*act'(zy) at the argmax position of each 2x2 window, 0 elsewhere.
    const float* zy;
    const uint8_t* zidx;
    int zmode;
    ActGradCoef zk;
    int ks;        // input channels of one output tile are split over ks adjacent lanes (1, 2 or 4) and summed by shuffles
    int epi_lds;   // 1: outputs leave through an LDS tile [sample][pixel][out_cs] as coalesced stores
    int out_cs;    // floats per pixel of that tile
    int mfma;      // 0: vector path; 1..3: the matrix-core variant compiled for this shape (conv_s1_mfma_pass)
};

// ---- filter chunk -> w_lds[tap][c][cop] (cop = channel tiles * CO_T; padded channels stay zero from the LDS fill)
struct WJob {
    int kind;  // 0: float4 over 4 output channels (plain layout); 1: float4 over 4 input channels (flipped); 2: scalars
    int units;
    int ci0, cic, co_base, coc, cop;
};
__device__ __forceinline__ WJob make_wjob(const TiledArgs& a, int ci0, int cic, int co_base, int coc, int cop, int K2) {
    WJob j;
    j.ci0 = ci0; j.cic = cic; j.co_base = co_base; j.coc = coc; j.cop = cop;
    if (!a.wflip && (a.Cout & 3) == 0 && (coc & 3) == 0 && (co_base & 3) == 0) {
        j.kind = 0; j.units = K2 * cic * (coc >> 2);
    } else if (a.wflip && (a.Cin & 3) == 0 && (cic & 3) == 0 && (ci0 & 3) == 0) {
        j.kind = 1; j.units = K2 * coc * (cic >> 2);
    } else {
        j.kind = 2; j.units = K2 * cic * coc;
    }
    return j;
}
template <int K, int CO_T>
__device__ __forceinline__ void w_pass(const TiledArgs& a, const WJob& j, float* w_lds, int nt) {
    const FDiv fcic = fd_make(j.cic), fcoc = fd_make(j.coc);
    constexpr int U = 4;  // float4 loads issued per thread before the first LDS write (indices clamped: unconditional)
    if (j.kind == 0) {
        const int c4n = j.coc >> 2;
        const FDiv f4 = fd_make(c4n);
        for (int u0 = threadIdx.x; u0 < j.units; u0 += U * nt) {
            float4 v[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int u = min(u0 + i * nt, j.units - 1);
                const int r = fd(u, f4), co4 = u - r * c4n;
                const int tap = fd(r, fcic), c = r - tap * j.cic;
                v[i] = *reinterpret_cast<const float4*>(a.w + ((size_t)tap * a.Cin + j.ci0 + c) * a.Cout + j.co_base + 4 * co4);
            }
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int u = u0 + i * nt;
                if (u < j.units) {
                    const int r = fd(u, f4), co4 = u - r * c4n;
                    const int tap = fd(r, fcic), c = r - tap * j.cic;
                    *reinterpret_cast<float4*>(w_lds + w_lds_index<K, CO_T>(4 * co4, c, tap, a.cic)) = v[i];
                }
            }
        }
    } else if (j.kind == 1) {
        const int c4n = j.cic >> 2;
        const FDiv f4 = fd_make(c4n);
        for (int u0 = threadIdx.x; u0 < j.units; u0 += U * nt) {
            float4 v[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int u = min(u0 + i * nt, j.units - 1);
                const int r = fd(u, f4), c4 = u - r * c4n;
                const int tap = fd(r, fcoc), co = r - tap * j.coc;
                v[i] = *reinterpret_cast<const float4*>(a.w + ((size_t)(K * K - 1 - tap) * a.Cout + j.co_base + co) * a.Cin +
                                                        j.ci0 + 4 * c4);
            }
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int u = u0 + i * nt;
                if (u < j.units) {
                    const int r = fd(u, f4), c4 = u - r * c4n;
                    const int tap = fd(r, fcoc), co = r - tap * j.coc;
                    float* q = w_lds + w_lds_index<K, CO_T>(co, 4 * c4, tap, a.cic);
                    q[0] = v[i].x; q[K * K * CO_T] = v[i].y; q[2 * K * K * CO_T] = v[i].z; q[3 * K * K * CO_T] = v[i].w;
                }
            }
        }
    } else if (a.wflip) {
#pragma unroll 4
        for (int u = threadIdx.x; u < j.units; u += nt) {
            const int r = fd(u, fcoc), co = u - r * j.coc;
            const int tap = fd(r, fcic), c = r - tap * j.cic;
            w_lds[w_lds_index<K, CO_T>(co, c, tap, a.cic)] =
                a.w[((size_t)(K * K - 1 - tap) * a.Cout + j.co_base + co) * a.Cin + j.ci0 + c];
        }
    } else {
#pragma unroll 4
        for (int u = threadIdx.x; u < j.units; u += nt) {
            const int r = fd(u, fcoc), co = u - r * j.coc;
            const int tap = fd(r, fcic), c = r - tap * j.cic;
            w_lds[w_lds_index<K, CO_T>(co, c, tap, a.cic)] = a.w[((size_t)tap * a.Cin + j.ci0 + c) * a.Cout + j.co_base + co];
        }
    }
}

// ---- one pass on the matrix cores (3x3, no upsampling, CIN in {8, 16, 32} staged channels, 16 * NT output channels,
// images 8 or 16 pixels wide, the block owns all channels and makes a single pass).  y[px][co] = sum_{tap, ci}
// x[px + tap][ci] * w[tap][ci][co] as v_mfma_f32_16x16x4_f32 tiles: lane l supplies A[pixel l % 16][channel l / 16]
// from the planar staged image and B[channel l / 16][co l % 16] from the filter chunk ([co / 8][ci][tap][8]).  A wave
// owns units of 32 pixels (4 rows x 8 or 2 rows x 16: whole 2 x 2 windows); results wait in registers until every wave
// is done with the staging buffers, then go to the LDS output tile [sample][pixel][out_cs] the stores read.
typedef float ct_v4f __attribute__((ext_vector_type(4)));

template <int CIN, int NT, int EPI>
__device__ __forceinline__ void conv_s1_mfma_pass(const TiledArgs& a, float* smem, int n0, int ns, int nt) {
    constexpr int K = 3, KK = 9, CO8 = 8;
    const float* in_lds = smem;
    const float* w_lds = smem + (size_t)a.S * a.cic * a.plane;
    const bool wide = a.Wo == 16;
    const int units_ps = (a.Ho * a.Wo) >> 5;
    const int nw = nt >> 6;
    const int upw = (a.S * units_ps + nw - 1) / nw;  // units per wave (compile-time bound below)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int T = (a.Ho >> 1) * (a.Wo >> 1), tiles_x = a.Wo >> 1;
    const int P = EPI == EPI_STORE ? a.Ho * a.Wo : T;
    const int cs = a.out_cs;
    int boff[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 16 + li;
        boff[n] = ((co >> 3) * a.cic + lk) * (KK * CO8) + (co & 7);
    }
    constexpr int MAXU = 4;  // the planner admits at most this many units per wave
    ct_v4f acc[MAXU][2][NT];
    const FDiv f_ups = fd_make(units_ps);
#pragma unroll
    for (int k = 0; k < MAXU; ++k) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[k][t][n] = ct_v4f{0.f, 0.f, 0.f, 0.f};
        const int u = wave + k * nw;
        if (k >= upw || u >= ns * units_ps) continue;  // uniform over the wave
        const int s = fd(u, f_ups), uu = u - s * units_ps;
        int pr[2], pc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            pr[t] = wide ? 2 * uu + t : 4 * uu + 2 * t + (li >> 3);
            pc[t] = wide ? li : (li & 7);
        }
        const float* ins = in_lds + ((size_t)s * a.cic + lk) * a.plane;
        // operands of a whole tap are read into registers one tap ahead of the MFMAs that use them
        constexpr int CG = CIN / 4;
        float av[2][CG][2], bv[2][CG][NT];
        auto load_tap = [&](int tap, float (&a_)[CG][2], float (&b_)[CG][NT]) {
            const int dr = tap / K - 1, dc = tap % K - 1;
            int ao[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) ao[t] = (pr[t] + dr + a.halo) * a.wp + pc[t] + dc + a.halo;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
#pragma unroll
                for (int t = 0; t < 2; ++t) a_[cg][t] = ins[cg * 4 * a.plane + ao[t]];
#pragma unroll
                for (int n = 0; n < NT; ++n) b_[cg][n] = w_lds[boff[n] + cg * 4 * (KK * CO8) + tap * CO8];
            }
        };
        load_tap(0, av[0], bv[0]);
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            if (tap + 1 < KK) load_tap(tap + 1, av[(tap + 1) & 1], bv[(tap + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[k][t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tap & 1][cg][t], bv[tap & 1][cg][n], acc[k][t][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();  // every wave is done reading the staged image and filters: the output tile may overwrite them
#pragma unroll
    for (int k = 0; k < MAXU; ++k) {
        const int u = wave + k * nw;
        if (k >= upw || u >= ns * units_ps) continue;
        const int s = fd(u, f_ups), uu = u - s * units_ps;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n * 16 + li;
            // lane holds, per tile, the pixels 4 lk .. 4 lk + 3 of channel co
            if (EPI == EPI_STORE) {
                const float b = a.bias != nullptr ? a.bias[co] : 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int p = 4 * lk + j;
                        const int row = wide ? 2 * uu + t : 4 * uu + 2 * t + (p >> 3);
                        const int col = wide ? p : (p & 7);
                        smem[((size_t)s * P + row * a.Wo + col) * cs + co] = act_apply(acc[k][t][n][j] + b, a.act, a.alpha);
                    }
            } else {  // EPI_SUM4: (v00 + v01) + (v10 + v11) per 2 x 2 window
                if (wide) {
                    smem[((size_t)s * P + uu * tiles_x + 2 * lk) * cs + co] =
                        (acc[k][0][n][0] + acc[k][0][n][1]) + (acc[k][1][n][0] + acc[k][1][n][1]);
                    smem[((size_t)s * P + uu * tiles_x + 2 * lk + 1) * cs + co] =
                        (acc[k][0][n][2] + acc[k][0][n][3]) + (acc[k][1][n][2] + acc[k][1][n][3]);
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        float o[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = __shfl_xor(acc[k][t][n][j], 32, 64);
                        if (lk < 2) {
                            float* q = smem + ((size_t)s * P + (2 * uu + t) * tiles_x + 2 * lk) * cs + co;
                            q[0] = (acc[k][t][n][0] + acc[k][t][n][1]) + (o[0] + o[1]);
                            q[cs] = (acc[k][t][n][2] + acc[k][t][n][3]) + (o[2] + o[3]);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    const int c4n = a.Cout >> 2;
    const FDiv f4 = fd_make(c4n);
    for (int u = threadIdx.x; u < ns * P * c4n; u += nt) {
        const int r = fd(u, f4), c4 = u - r * c4n;
        *reinterpret_cast<float4*>(a.y + ((size_t)n0 * P + r) * a.Cout + 4 * c4) =
            *reinterpret_cast<const float4*>(smem + (size_t)r * cs + 4 * c4);
    }
}

template <int K, int UP, int CO_T, int EPI>
__global__ __launch_bounds__(256, 2) void conv_s1_lds(TiledArgs a) {
    constexpr int PADL = (K - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* in_lds = smem;                                   // [S][cic][plane]
    float* w_lds = smem + (size_t)a.S * a.cic * a.plane;    // [K*K][cic][cop]
    const int nt = blockDim.x;

    const int tiles_x = a.Wo >> 1, tiles_y = a.Ho >> 1;
    const int T = tiles_x * tiles_y;
    const int co_base = blockIdx.y * a.coc;
    const int coc = min(a.coc, a.Cout - co_base);  // channels this block really owns
    const int n_cot = (coc + CO_T - 1) / CO_T;
    const int cop = n_cot * CO_T;
    const int ks = a.ks;
    const int per_cot = a.S * T * ks;              // work items per channel tile in one pass
    const int total = n_cot * per_cot;
    const bool one_chunk = a.cic >= a.Cin;
    const FDiv f_percot = fd_make(per_cot), f_ks = fd_make(ks), f_T = fd_make(T), f_tx = fd_make(tiles_x);

    PlanarSrc src;
    src.x = a.x; src.zy = a.zy; src.zidx = a.zidx; src.mode = a.zmode; src.zk = a.zk; src.C = a.Cin;
    src.Hs = a.zmode == 2 ? a.Hs >> 1 : a.Hs;
    src.Ws = a.zmode == 2 ? a.Ws >> 1 : a.Ws;
    const PlanarDst dst{a.cic, a.plane, a.wp, a.halo};

    SV_MARK(0);
    // ---- prologue: zero the LDS (halo, row padding and padded channels are never overwritten afterwards), then
    // stage the filter chunk and the first pass's input
    int n0 = blockIdx.x * a.S;
    {
        const int lds_f4 = (a.S * a.cic * a.plane + K * K * a.cic * cop + 3) >> 2;
        for (int i = threadIdx.x; i < lds_f4; i += nt) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        SV_MARK(1);
        const int ns = min(a.S, a.N - n0);
        const int cic0 = min(a.cic, a.Cin);
        w_pass<K, CO_T>(a, make_wjob(a, 0, cic0, co_base, coc, cop, K * K), w_lds, nt);
        stage_pass(src, dst, make_stage_job(src, n0, ns, 0, cic0), in_lds, nt);
    }
    bool prestaged = true;

    if constexpr (K == 3 && UP == 1 && CO_T == 8 && (EPI == EPI_STORE || EPI == EPI_SUM4)) {
        if (a.mfma) {  // single pass, all channels (the planner's conditions)
            __syncthreads();
            SV_MARK(2);
            const int ns = min(a.S, a.N - n0);
            if (EPI == EPI_SUM4) {
                if (a.mfma == 1) conv_s1_mfma_pass<8, 1, EPI_SUM4>(a, smem, n0, ns, nt);
                else conv_s1_mfma_pass<16, 2, EPI_SUM4>(a, smem, n0, ns, nt);
            } else {
                conv_s1_mfma_pass<32, 1, EPI_STORE>(a, smem, n0, ns, nt);
            }
            SV_MARK(5);
            return;
        }
    }
    for (; n0 < a.N; n0 += gridDim.x * a.S) {
        const int ns = min(a.S, a.N - n0);
        // A thread owns one work item per pass of nt; its accumulators live across the ci-chunk loop.  When a
        // sample pass has more than nt items the passes are looped and each re-walks (re-stages) the chunks.
        for (int item0 = 0; item0 < total; item0 += nt) {
            const int item = item0 + threadIdx.x;
            const bool live = item < total;
            int cot = 0, s = 0, ty = 0, tx = 0, kl = 0;
            if (live) {
                cot = fd(item, f_percot);
                int r = item - cot * per_cot;
                const int r1 = fd(r, f_ks);
                kl = r - r1 * ks;  // which slice of the input channels this lane sums
                s = fd(r1, f_T);
                const int t = r1 - s * T;
                ty = fd(t, f_tx);
                tx = t - ty * tiles_x;
            }
            const bool work = live && s < ns;
            float acc[2][2][CO_T];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int c = 0; c < CO_T; ++c) acc[i][j][c] = 0.f;

            for (int ci0 = 0; ci0 < a.Cin; ci0 += a.cic) {
                const int cic = min(a.cic, a.Cin - ci0);
                if (!prestaged) {
                    __syncthreads();  // previous chunk / item pass is done with LDS
                    if (item0 == 0 || !one_chunk) stage_pass(src, dst, make_stage_job(src, n0, ns, ci0, cic), in_lds, nt);
                    if (!one_chunk) w_pass<K, CO_T>(a, make_wjob(a, ci0, cic, co_base, coc, cop, K * K), w_lds, nt);
                }
                prestaged = false;
                __syncthreads();
                SV_MARK(2);
                if (work) {
                    const float* img = in_lds + (size_t)s * a.cic * a.plane;
                    // patch origin in the padded stored image (even column when UP == 1: 8-byte aligned rows)
                    const int r0 = (UP == 1 ? 2 * ty : ty) + a.halo + floordiv_c(-PADL, UP);
                    const int c0 = (UP == 1 ? 2 * tx : tx) + a.halo + floordiv_c(-PADL, UP);
                    conv_tile_fma<K, UP, CO_T>(acc, img, w_lds + cot * a.cic * K * K * CO_T, a.plane, a.wp, r0, c0, kl, cic, ks);
                }
            }
            SV_MARK(3);
            // ---- fold the ks channel slices (adjacent lanes of one wave; dead lanes hold zeros)
            for (int o = 1; o < ks; o <<= 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < CO_T; ++q) acc[i][j][q] += __shfl_xor(acc[i][j][q], o, 64);
            }
            // ---- epilogue values: bias + activation (+ 2x2 max / 2x2 sum)
            const int cg = co_base + cot * CO_T;  // first global output channel of this thread
            const int n = n0 + s;
            if (EPI == EPI_STORE) {
#pragma unroll
                for (int q = 0; q < CO_T; ++q) {
                    const float b = (a.bias != nullptr && cg + q < a.Cout) ? a.bias[cg + q] : 0.f;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j][q] = act_apply(acc[i][j][q] + b, a.act, a.alpha);
                }
            }
            float red[CO_T];      // MAXPOOL / SUM4: one value per channel
            uint32_t slot[CO_T];  // MAXPOOL: argmax slot
            if (EPI == EPI_MAXPOOL) {
                // max of the pre-activations first (every activation here is non-decreasing, so the maximum and -
                // wherever the activation is strictly increasing - its position are those of the activated
                // values), then ONE activation per window.  relu flattens negatives into ties at 0 whose "first
                // maximum" is slot 0 in the unfused order; no gradient flows there (relu'(0) = 0), so only the
                // recorded slot could differ, and it is made to agree below.
#pragma unroll
                for (int q = 0; q < CO_T; ++q) {
                    const float b = (a.bias != nullptr && cg + q < a.Cout) ? a.bias[cg + q] : 0.f;
                    float best = acc[0][0][q];
                    uint32_t bi = 0;
                    if (acc[0][1][q] > best) { best = acc[0][1][q]; bi = 1; }
                    if (acc[1][0][q] > best) { best = acc[1][0][q]; bi = 2; }
                    if (acc[1][1][q] > best) { best = acc[1][1][q]; bi = 3; }
                    const float v = act_apply(best + b, a.act, a.alpha);
                    if (a.act == SV_ACT_RELU && v <= 0.f) bi = 0;
                    red[q] = v;
                    slot[q] = bi;
                }
            } else if (EPI == EPI_SUM4) {
#pragma unroll
                for (int q = 0; q < CO_T; ++q) red[q] = (acc[0][0][q] + acc[0][1][q]) + (acc[1][0][q] + acc[1][1][q]);
            }

            if (a.epi_lds) {
                // ---- transpose through LDS: every thread drops its values at [sample][pixel][channel] of the block's
                // output tile (the staging buffers are dead: one item pass, one chunk), then the whole tile - contiguous
                // in global memory when the block owns all channels - leaves as full-width coalesced stores.
                const int cs = a.out_cs;                       // floats per pixel in the LDS tile (padded)
                const int P = EPI == EPI_STORE ? a.Ho * a.Wo : T;  // output pixels per sample
                __syncthreads();
                if (work && kl == 0) {
                    if (EPI == EPI_STORE) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                float* o = smem + ((size_t)s * P + (2 * ty + i) * a.Wo + 2 * tx + j) * cs + cot * CO_T;
#pragma unroll
                                for (int q = 0; q < CO_T; q += 4)
                                    *reinterpret_cast<float4*>(o + q) =
                                        make_float4(acc[i][j][q], acc[i][j][q + 1], acc[i][j][q + 2], acc[i][j][q + 3]);
                            }
                    } else {
                        float* o = smem + ((size_t)s * P + ty * tiles_x + tx) * cs + cot * CO_T;
#pragma unroll
                        for (int q = 0; q < CO_T; q += 4)
                            *reinterpret_cast<float4*>(o + q) = make_float4(red[q], red[q + 1], red[q + 2], red[q + 3]);
                        if (EPI == EPI_MAXPOOL && a.idx != nullptr) {
                            uint32_t* ob = reinterpret_cast<uint32_t*>(smem + (size_t)a.S * P * cs) +
                                           (((size_t)s * P + ty * tiles_x + tx) * cop + cot * CO_T) / 4;
#pragma unroll
                            for (int q = 0; q < CO_T; q += 4)
                                ob[q >> 2] = slot[q] | (slot[q + 1] << 8) | (slot[q + 2] << 16) | (slot[q + 3] << 24);
                        }
                    }
                }
                __syncthreads();
                const int rows = ns * P;  // one row = the coc channels of one pixel
                if ((coc & 3) == 0 && (a.Cout & 3) == 0) {
                    const int c4n = coc >> 2;
                    const FDiv f4 = fd_make(c4n);
                    for (int u = threadIdx.x; u < rows * c4n; u += nt) {
                        const int r = fd(u, f4), c4 = u - r * c4n;
                        const float4 v = *reinterpret_cast<const float4*>(smem + (size_t)r * cs + 4 * c4);
                        *reinterpret_cast<float4*>(a.y + ((size_t)n0 * P + r) * a.Cout + co_base + 4 * c4) = v;
                    }
                    if (EPI == EPI_MAXPOOL && a.idx != nullptr) {
                        const uint32_t* ob = reinterpret_cast<const uint32_t*>(smem + (size_t)a.S * P * cs);
                        for (int u = threadIdx.x; u < rows * c4n; u += nt) {
                            const int r = fd(u, f4), c4 = u - r * c4n;
                            *reinterpret_cast<uint32_t*>(a.idx + ((size_t)n0 * P + r) * a.Cout + co_base + 4 * c4) =
                                ob[((size_t)r * cop) / 4 + c4];
                        }
                    }
                } else {
                    const FDiv fc = fd_make(coc);
                    for (int u = threadIdx.x; u < rows * coc; u += nt) {
                        const int r = fd(u, fc), c = u - r * coc;
                        a.y[((size_t)n0 * P + r) * a.Cout + co_base + c] = smem[(size_t)r * cs + c];
                    }
                    if (EPI == EPI_MAXPOOL && a.idx != nullptr) {
                        const uint8_t* ob = reinterpret_cast<const uint8_t*>(smem + (size_t)a.S * P * cs);
                        for (int u = threadIdx.x; u < rows * coc; u += nt) {
                            const int r = fd(u, fc), c = u - r * coc;
                            a.idx[((size_t)n0 * P + r) * a.Cout + co_base + c] = ob[(size_t)r * cop + c];
                        }
                    }
                }
                prestaged = false;
            } else if (work && kl == 0) {
                // ---- direct stores (multi-pass blocks, or an output tile larger than the staging footprint)
                if (EPI == EPI_STORE) {
                    const bool vec = (a.Cout & 3) == 0;  // cg is a multiple of 4: 16-byte aligned rows of channels
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            float* yp = a.y + (((size_t)n * a.Ho + 2 * ty + i) * a.Wo + 2 * tx + j) * a.Cout + cg;
                            if (vec) {
#pragma unroll
                                for (int q = 0; q < CO_T; q += 4)
                                    if (cg + q < a.Cout)
                                        *reinterpret_cast<float4*>(yp + q) =
                                            make_float4(acc[i][j][q], acc[i][j][q + 1], acc[i][j][q + 2], acc[i][j][q + 3]);
                            } else {
#pragma unroll
                                for (int q = 0; q < CO_T; ++q)
                                    if (cg + q < a.Cout) yp[q] = acc[i][j][q];
                            }
                        }
                } else {
                    const size_t o = (((size_t)n * tiles_y + ty) * tiles_x + tx) * a.Cout + cg;
#pragma unroll
                    for (int q = 0; q < CO_T; ++q)
                        if (cg + q < a.Cout) {
                            a.y[o + q] = red[q];
                            if (EPI == EPI_MAXPOOL && a.idx != nullptr) a.idx[o + q] = (uint8_t)slot[q];
                        }
                }
            }
            SV_MARK(4);
        }
    }
    __syncthreads();
    SV_MARK(5);
}

// ------------------------------------------------------------------------------------------------ host side
struct TiledPlan {
    bool ok;
    int S, cic, coc, plane, wp, halo, co_t, nt, ks;
    size_t lds_bytes;
    dim3 grid;
};

// LDS budget per workgroup: 64 KiB (the dynamic-LDS limit without an opt-in attribute) lets two workgroups
// share a CU's 160 KiB, i.e. at least two waves per SIMD in flight.
constexpr size_t kLdsBudget = 64 * 1024;

TiledPlan plan_tiled(int N, int Hs, int Ws, int Cin, int Cout, int K, int UP) {
    TiledPlan p{};
    p.ok = false;
    const int Ho = Hs * UP, Wo = Ws * UP;
    if ((Ho & 1) || (Wo & 1)) return p;
    const int padl = (K - 1) / 2;
    p.halo = (padl + UP - 1) / UP;
    p.wp = (Ws + 2 * p.halo + 3) & ~3;
    p.plane = (Hs + 2 * p.halo) * p.wp + 4;
    p.co_t = Cout > 4 ? 8 : 4;
    const int T = (Ho / 2) * (Wo / 2);
    // channels per block: as many as keep the filter chunk small; whole Cout when it is small
    int coc = ((Cout + p.co_t - 1) / p.co_t) * p.co_t;
    if (coc > 32) coc = 32;
    int cic = Cin;
    auto bytes = [&](int S, int cic_, int coc_) {
        return ((size_t)S * cic_ * p.plane + (size_t)K * K * cic_ * coc_) * sizeof(float);
    };
    while (cic > 1 && bytes(1, cic, coc) > kLdsBudget) cic = (cic + 1) / 2;
    if (bytes(1, cic, coc) > kLdsBudget) return p;
    const int n_cot = coc / p.co_t;
    const int ny = (Cout + coc - 1) / coc;
    // Parallelism first: the launch should put >= 2 waves on every SIMD (2048 waves on the chip).  When the
    // output tiles alone are too few (small images), each tile's input channels are split over 2 or 4 lanes.
    int ks = 1;
    while (ks < 4 && cic / (2 * ks) >= 4 && (int64_t)N * T * n_cot * ny * ks < 2048 * 64) ks <<= 1;
    // A pass holds ~256 work items; small layers take several samples per pass.
    const int items1 = T * n_cot * ks;  // items of one sample
    int S = items1 >= 256 ? 1 : 256 / items1;
    while (S > 1 && bytes(S, cic, coc) > kLdsBudget) --S;
    if (S > N) S = N;
    int nt = 256;
    while (nt > 64 && S * items1 <= nt / 2) nt >>= 1;
    p.nt = nt;
    p.ks = ks;
    p.S = S;
    p.cic = cic;
    p.coc = coc;
    p.lds_bytes = bytes(S, cic, coc);
    const int passes = (N + S - 1) / S;
    int gx = passes < 4096 ? passes : 4096;
    p.grid = dim3((unsigned)gx, (unsigned)ny);
    p.ok = true;
    return p;
}

template <int K, int UP, int CO_T>
int launch_epi(const TiledArgs& a, int epi, const TiledPlan& p, hipStream_t st) {
    switch (epi) {
        case EPI_STORE:
            hipLaunchKernelGGL((conv_s1_lds<K, UP, CO_T, EPI_STORE>), p.grid, dim3(p.nt), p.lds_bytes, st, a);
            break;
        case EPI_MAXPOOL:
            hipLaunchKernelGGL((conv_s1_lds<K, UP, CO_T, EPI_MAXPOOL>), p.grid, dim3(p.nt), p.lds_bytes, st, a);
            break;
        default:
            hipLaunchKernelGGL((conv_s1_lds<K, UP, CO_T, EPI_SUM4>), p.grid, dim3(p.nt), p.lds_bytes, st, a);
            break;
    }
    return check_launch("conv_s1_lds");
}

template <int K, int UP>
int launch_cot(const TiledArgs& a, int epi, const TiledPlan& p, hipStream_t st) {
    return p.co_t == 8 ? launch_epi<K, UP, 8>(a, epi, p, st) : launch_epi<K, UP, 4>(a, epi, p, st);
}

// Returns 1 when the shape is not covered (caller falls back to the direct kernels), 0 on success, <0 on error.
// Logical conv: input (Hs*UP, Ws*UP, Cin) -> output (Hs*UP, Ws*UP, Cout), K x K, stride 1, SAME.
bool tiled_s1_supported(int N, int Hs, int Ws, int Cin, int Cout, int K, int UP) {
    if (!((K == 3 || K == 5) && (UP == 1 || UP == 2))) return false;
    return plan_tiled(N, Hs, Ws, Cin, Cout, K, UP).ok;
}

int launch_tiled_s1(const float* x, const float* w, const float* bias, float* y, uint8_t* idx, int N, int Hs, int Ws,
                    int Cin, int Cout, int K, int UP, int wflip, int epi, int act, float alpha, hipStream_t st,
                    int zmode, const float* zy, const uint8_t* zidx, int zact, float zalpha) {
    if (!((K == 3 || K == 5) && (UP == 1 || UP == 2))) return 1;
    TiledPlan p = plan_tiled(N, Hs, Ws, Cin, Cout, K, UP);
    if (!p.ok) return 1;
    if (zmode == 2 && ((Hs | Ws) & 1)) return 1;
    TiledArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.idx = idx;
    a.N = N; a.Hs = Hs; a.Ws = Ws; a.Cin = Cin; a.Cout = Cout;
    a.Ho = Hs * UP; a.Wo = Ws * UP;
    a.wflip = wflip; a.act = act; a.alpha = alpha;
    a.S = p.S; a.cic = p.cic; a.coc = p.coc; a.plane = p.plane; a.wp = p.wp; a.halo = p.halo;
    a.zmode = zmode; a.zy = zy; a.zidx = zidx; a.zk = act_grad_coef(zact, zalpha);
    a.ks = p.ks;
    // Outputs leave through an LDS tile when the block makes exactly one pass (the tile reuses the staging buffers)
    {
        const int T = (a.Ho / 2) * (a.Wo / 2);
        const int n_cot = p.coc / p.co_t;
        const size_t P = epi == EPI_STORE ? (size_t)a.Ho * a.Wo : (size_t)T;
        const int cs = p.coc + 4;
        const size_t out_bytes = (size_t)p.S * P * cs * sizeof(float) + (epi == EPI_MAXPOOL ? (size_t)p.S * P * p.coc : 0);
        const bool single = (int64_t)p.S * T * n_cot * p.ks <= p.nt && p.cic >= Cin &&
                            (int64_t)p.grid.x * p.S >= N;
        a.epi_lds = single && out_bytes <= kLdsBudget;
        a.out_cs = cs;
        // matrix-core variants: shapes whose 16-wide tiles are full, one pass, all channels in the block, at most 4
        // 32-pixel units per wave
        a.mfma = 0;
        if (K == 3 && UP == 1 && a.epi_lds && p.grid.y == 1 && p.coc == Cout && (a.Wo == 8 || a.Wo == 16) &&
            (a.Ho * a.Wo) % 32 == 0 && (p.nt & 63) == 0 && (a.bias == nullptr || epi == EPI_STORE) &&
            (int64_t)p.S * ((a.Ho * a.Wo) >> 5) <= 4 * (p.nt >> 6) && opt(OPT_DGRAD_VALU) != 1) {
            if (epi == EPI_SUM4 && Cin == 8 && Cout == 16) a.mfma = 1;
            else if (epi == EPI_SUM4 && Cin == 16 && Cout == 32) a.mfma = 2;
            else if (epi == EPI_STORE && Cin == 32 && Cout == 16) a.mfma = 3;
        }
        if (a.epi_lds && out_bytes > p.lds_bytes) p.lds_bytes = out_bytes;
        p.lds_bytes = (p.lds_bytes + 15) & ~(size_t)15;
    }
    if (K == 3) return UP == 1 ? launch_cot<3, 1>(a, epi, p, st) : launch_cot<3, 2>(a, epi, p, st);
    return UP == 1 ? launch_cot<5, 1>(a, epi, p, st) : launch_cot<5, 2>(a, epi, p, st);
}

}  // namespace sv

#ifdef SV_TRACE
extern "C" int sv_debug_set_trace(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(sv::g_sv_trace), &buf, sizeof(buf));
}
#endif

// =================================================================================================== kernel gradient
// dw[kh][kw][ci][co] = sum over samples and output pixels of x[n, oh+kh-PADL, ow+kw-PADL, ci] * dz[n, oh, ow, co]
// (x read through nearest upsampling when UP > 1: the LDS image is staged at the logical resolution).
//
// A thread owns one (input channel, CO_T output channels) pair and ALL K*K taps in registers, for one of PG
// pixel groups: per strip of 4 output pixels it loads K rows x (K+3) cols of x and 4 x CO_T values of dz with
// 16-byte LDS reads and issues 4*K*K*CO_T v_fma.  Workgroups are persistent over sample passes; partial sums
// are folded across the PG lanes of a wave by shuffles, written as one slab per workgroup and summed in a
// fixed order by wgrad_reduce (bitwise reproducible, no float atomics).
namespace sv {

struct WgradArgs {
    const float* x;   // [N][Hs][Ws][Cin]
    const float* dz;  // [N][Ho][Wo][Cout]
    float* slab;      // [gridDim.x][ny*nz][chunk][combo][RS]: partial sums in fold order
    int N, Hs, Ws, Cin, Cout, Ho, Wo, up;
    int S, cib, cob;  // samples per pass; input channels per block (grid.y); output channels per block (grid.z)
    int PG;           // pixel groups (power of two <= 64)
    int xplane, xwp, zplane;
    // dz formed on the fly, as in TiledArgs: 0 = dz given; 1 = dz*act'(zy); 2 = pooled (dz, zy, zidx at Ho/2 x Wo/2)
    const float* zy;
    const uint8_t* zidx;
    int zmode;
    ActGradCoef zk;
    int with_bias;    // also emit sum(dz) per output channel
    int combos_max;   // cib * (cob / CO_T): combo stride of the slab layout
    int mfma;         // partial sums on the matrix cores (wgrad_body_mfma); the planner vouches for the shape
};

// Slab layout shared by wgrad_s1_lds (writer) and wgrad_reduce3 (reader):
//   part = blockIdx.z * gridDim.y + blockIdx.y;  slot = tap * CO_T + q  (kernel sums)  or  K*K*CO_T + q  (bias sums)
//   offset = ((part * NCHUNK + slot / RS) * combos_max + combo) * RS + slot % RS,   combo = ci_local * n_cot + co_tile
constexpr int kWgradRS = 20;

struct BlockPos {
    int bx, by, bz;  // workgroup coordinates
    int gx, gy, gz;  // grid extents
};

template <int CIN, int NT>
__device__ __forceinline__ void wgrad_body_mfma(const WgradArgs& a, const BlockPos bp);

template <int K, int CO_T>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const BlockPos bp) {
    constexpr int PADL = (K - 1) / 2;
    constexpr int XC = K + 3;  // x columns needed by a strip of 4 output pixels
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (K == 3 && CO_T == 8) {
        if (a.mfma) {  // 18 accumulator tiles per wave: (Cin, Cout) = (16, 32) or (32, 16)
            if (a.Cin == 16) wgrad_body_mfma<16, 2>(a, bp);
            else wgrad_body_mfma<32, 1>(a, bp);
            return;
        }
    }
    const int ci_base = bp.by * a.cib;
    const int cib = min(a.cib, a.Cin - ci_base);
    const int co_base = bp.bz * a.cob;
    const int cob = min(a.cob, a.Cout - co_base);
    const int n_cot = (cob + CO_T - 1) / CO_T;
    float* x_lds = smem;                                       // [S][cib][xplane]   logical resolution + halo
    float* z_lds = smem + (size_t)a.S * a.cib * a.xplane;      // [S][n_cot*CO_T][zplane]
    const int combos = cib * n_cot;
    const int PG = a.PG;
    const int combo = threadIdx.x / PG, pg = threadIdx.x - combo * PG;
    const bool live = combo < combos;
    const int cil = live ? combo / n_cot : 0;
    const int cot = live ? combo - cil * n_cot : 0;
    const int strips_x = a.Wo >> 2;
    const int strips = a.Ho * strips_x;
    const FDiv f_strips = fd_make(strips), f_sx = fd_make(strips_x);

    float acc[K * K][CO_T];
    float accb[CO_T];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int q = 0; q < CO_T; ++q) acc[t][q] = 0.f;
#pragma unroll
    for (int q = 0; q < CO_T; ++q) accb[q] = 0.f;
    const bool do_bias = a.with_bias && live && cil == 0 && bp.by == 0;

    PlanarSrc xs;
    xs.x = a.x; xs.zy = nullptr; xs.zidx = nullptr; xs.mode = a.up == 2 ? 3 : 0; xs.zk = a.zk;
    xs.Hs = a.Hs; xs.Ws = a.Ws; xs.C = a.Cin;
    const PlanarDst xd{a.cib, a.xplane, a.xwp, PADL};
    PlanarSrc zs;
    zs.x = a.dz; zs.zy = a.zy; zs.zidx = a.zidx; zs.mode = a.zmode; zs.zk = a.zk; zs.C = a.Cout;
    zs.Hs = a.zmode == 2 ? a.Ho >> 1 : a.Ho;
    zs.Ws = a.zmode == 2 ? a.Wo >> 1 : a.Wo;
    const PlanarDst zd{n_cot * CO_T, a.zplane, a.Wo, 0};

    SV_MARK(0);
    // zero once: halo, row padding and padded channels never get overwritten
    const int lds_f4 = (a.S * (a.cib * a.xplane + n_cot * CO_T * a.zplane) + 3) >> 2;
    for (int i = threadIdx.x; i < lds_f4; i += 256) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    SV_MARK(1);

    for (int n0 = bp.bx * a.S; n0 < a.N; n0 += bp.gx * a.S) {
        const int ns = min(a.S, a.N - n0);
        __syncthreads();  // zero fill landed / previous pass is done reading
        stage_pass(xs, xd, make_stage_job(xs, n0, ns, ci_base, cib), x_lds, 256);
        stage_pass(zs, zd, make_stage_job(zs, n0, ns, co_base, cob), z_lds, 256);
        __syncthreads();
        SV_MARK(2);
        if (live) {
            for (int w = pg; w < ns * strips; w += PG) {
                const int ss = fd(w, f_strips);
                const int st = w - ss * strips;
                const int oh = fd(st, f_sx);
                const int ow0 = (st - oh * strips_x) << 2;
                // x rows oh .. oh+K-1 and cols ow0 .. ow0+K+2 of the padded image
                const float* xp = x_lds + (ss * a.cib + cil) * a.xplane + oh * a.xwp + ow0;
                float xv[K][XC];
#pragma unroll
                for (int r = 0; r < K; ++r) {
                    const float4 v4 = *reinterpret_cast<const float4*>(xp + r * a.xwp);
                    xv[r][0] = v4.x; xv[r][1] = v4.y; xv[r][2] = v4.z; xv[r][3] = v4.w;
                    // the K - 1 columns past the strip as 8-byte pairs (XC - 4 is even): consecutive lanes sit 16 bytes
                    // apart, where 4-byte reads collide four ways on the LDS banks
#pragma unroll
                    for (int c = 4; c < XC; c += 2) {
                        const float2 v2 = *reinterpret_cast<const float2*>(xp + r * a.xwp + c);
                        xv[r][c] = v2.x; xv[r][c + 1] = v2.y;
                    }
                }
                const float* zp = z_lds + (ss * n_cot * CO_T + cot * CO_T) * a.zplane + oh * a.Wo + ow0;
                float zv[CO_T][4];
#pragma unroll
                for (int q = 0; q < CO_T; ++q) {
                    const float4 v4 = *reinterpret_cast<const float4*>(zp + q * a.zplane);
                    zv[q][0] = v4.x; zv[q][1] = v4.y; zv[q][2] = v4.z; zv[q][3] = v4.w;
                }
                if (do_bias) {
#pragma unroll
                    for (int q = 0; q < CO_T; ++q) accb[q] += (zv[q][0] + zv[q][1]) + (zv[q][2] + zv[q][3]);
                }
#pragma unroll
                for (int kh = 0; kh < K; ++kh)
#pragma unroll
                    for (int kw = 0; kw < K; ++kw)
#pragma unroll
                        for (int q = 0; q < CO_T; ++q) {
                            float s = acc[kh * K + kw][q];
#pragma unroll
                            for (int j = 0; j < 4; ++j) s = fmaf(xv[kh][kw + j], zv[q][j], s);
                            acc[kh * K + kw][q] = s;
                        }
            }
        }
    }
    SV_MARK(3);
    // ---- fold the PG partial sums of every (ci, co tile) through LDS (the staging buffers are dead now) and
    // write this workgroup's slab in fold order (contiguous, coalesced).  Slots = K*K*CO_T kernel-gradient sums
    // + CO_T bias sums, folded in chunks of RS slots so the scratch stays small: red[thread][RS + 1].
    constexpr int NSLOT = K * K * CO_T + CO_T;
    constexpr int RS = kWgradRS;
    constexpr int NCHUNK = (NSLOT + RS - 1) / RS;
    const int nparts = bp.gy * bp.gz;
    const int part = bp.bz * bp.gy + bp.by;
    float* out = a.slab + ((size_t)bp.bx * nparts + part) * NCHUNK * a.combos_max * RS;
    float* red = smem;  // 256 * (RS + 1) floats = 21 KiB <= the LDS size the planner requests
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ++ch) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS; ++j) {
            const int slot = ch * RS + j;
            if (slot < NSLOT) {
                const float v = slot < K * K * CO_T ? acc[slot / CO_T][slot % CO_T] : accb[(slot - K * K * CO_T) % CO_T];
                red[threadIdx.x * (RS + 1) + j] = v;
            }
        }
        __syncthreads();
        const int nout = combos * RS;
        for (int o = threadIdx.x; o < nout; o += 256) {
            const int cb = o / RS, j = o - cb * RS;
            float sum = 0.f;
            if (ch * RS + j < NSLOT)
                for (int g = 0; g < PG; ++g) sum += red[(cb * PG + g) * (RS + 1) + j];
            out[(size_t)ch * a.combos_max * RS + o] = sum;
        }
    }
    SV_MARK(5);
}

// ---- the same partial sums on the matrix cores (3x3 kernels; 8 / 16 / 32 input channels, 8 / 16 / 32 output channels,
// one workgroup covering all channels).  dw[(tap, ci)][co] = sum_px x[px + tap][ci] * dz[px][co] is a GEMM with
// M = 9 * Cin, N = Cout and the pixels as the reduction; one v_mfma_f32_16x16x4_f32 takes 16 rows (tap, ci), 16 output
// channels and 4 consecutive pixels of an image row.  Lane l supplies A[row l % 16][pixel l / 16] and
// B[pixel l / 16][channel l % 16]: both are single 4-byte reads of the planar LDS images at a per-lane base plus a
// per-tile constant, so the loop carries no per-element address arithmetic.  The 4 waves split the pixel quads; their
// accumulators (all M x N tiles per wave) are summed in wave order through the slab image in LDS (bit-reproducible).
typedef float wg_v4f __attribute__((ext_vector_type(4)));

template <int CIN, int NT>
__device__ __forceinline__ void wgrad_body_mfma(const WgradArgs& a, const BlockPos bp) {
    constexpr int K = 3, CO_T = 8;
    constexpr int M = K * K * CIN;
    constexpr int MT = (M + 15) / 16;
    constexpr int NSLOT = K * K * CO_T + CO_T, RS = kWgradRS, NCHUNK = (NSLOT + RS - 1) / RS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cob = a.Cout;
    const int n_cot = (cob + CO_T - 1) / CO_T;
    const int ncp = n_cot * CO_T;  // dz planes per sample
    float* x_lds = smem;
    float* z_lds = smem + (size_t)a.S * CIN * a.xplane;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;

    PlanarSrc xs;
    xs.x = a.x; xs.zy = nullptr; xs.zidx = nullptr; xs.mode = a.up == 2 ? 3 : 0; xs.zk = a.zk;
    xs.Hs = a.Hs; xs.Ws = a.Ws; xs.C = a.Cin;
    const PlanarDst xd{CIN, a.xplane, a.xwp, 1};
    PlanarSrc zs;
    zs.x = a.dz; zs.zy = a.zy; zs.zidx = a.zidx; zs.mode = a.zmode; zs.zk = a.zk; zs.C = a.Cout;
    zs.Hs = a.zmode == 2 ? a.Ho >> 1 : a.Ho;
    zs.Ws = a.zmode == 2 ? a.Wo >> 1 : a.Wo;
    const PlanarDst zd{ncp, a.zplane, a.Wo, 0};

    // per-lane operand bases.  A row m = mt * 16 + li = tap * CIN + ci reads x[ci][oh + kh][ow + kw] (halo 1 baked in).
    int aoff[MT];
    bool a_ok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + li;
        a_ok[mt] = m < M;
        const int mm = a_ok[mt] ? m : 0;
        const int tap = mm / CIN, ci = mm - tap * CIN;
        const int kh = tap / K, kw = tap - kh * K;
        aoff[mt] = ci * a.xplane + kh * a.xwp + kw + lk;
    }
    int boff[NT];
    bool b_ok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 16 + li;
        b_ok[nt] = co < cob;
        boff[nt] = (b_ok[nt] ? co : 0) * a.zplane + lk;
    }

    wg_v4f acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = wg_v4f{0.f, 0.f, 0.f, 0.f};
    float bsum[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

    const int lds_f4 = (a.S * (CIN * a.xplane + ncp * a.zplane) + 3) >> 2;
    for (int i = threadIdx.x; i < lds_f4; i += 256) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    const int quads_x = a.Wo >> 2;
    const int quads = a.Ho * quads_x;
    const FDiv f_quads = fd_make(quads), f_qx = fd_make(quads_x);
    for (int n0 = bp.bx * a.S; n0 < a.N; n0 += bp.gx * a.S) {
        const int ns = min(a.S, a.N - n0);
        __syncthreads();  // zero fill landed / previous pass is done reading
        stage_pass(xs, xd, make_stage_job(xs, n0, ns, 0, CIN), x_lds, 256);
        stage_pass(zs, zd, make_stage_job(zs, n0, ns, 0, cob), z_lds, 256);
        __syncthreads();
        // U pixel quads per iteration: all their operand reads are issued before the first MFMA needs one
        constexpr int U = MT * NT >= 16 ? 2 : 4;
        const int nq = ns * quads;
        for (int w0 = wave; w0 < nq; w0 += 4 * U) {
            float av[U][MT], bv[U][NT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int w = w0 + 4 * u;
                const bool live = w < nq;  // uniform over the wave; a dead quad multiplies zeros
                const int wc = live ? w : w0;
                const int ss = fd(wc, f_quads);
                const int q = wc - ss * quads;
                const int oh = fd(q, f_qx);
                const int ow0 = (q - oh * quads_x) << 2;
                const float* xq = x_lds + ss * CIN * a.xplane + oh * a.xwp + ow0;
                const float* zq = z_lds + ss * ncp * a.zplane + oh * a.Wo + ow0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float v = zq[boff[nt]];
                    bv[u][nt] = (live && b_ok[nt]) ? v : 0.f;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    av[u][mt] = xq[aoff[mt]];
                    if (M % 16 != 0 && mt == MT - 1) av[u][mt] = a_ok[mt] ? av[u][mt] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bsum[nt] += bv[u][nt];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mt], bv[u][nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }
    // ---- sum the 4 waves as (w0 + w1) + (w2 + w3) through LDS (the staging buffers are dead): accumulator tiles travel
    // in their register layout as 16-byte pieces (no read-modify-write, so the LDS traffic pipelines), wave 0 then
    // places the sums in the slab image, which all threads copy out.
    // slab position of (tap, ci, co): combo = ci * n_cot + co / 8, slot = tap * 8 + co % 8 (bias: slot 72 + co % 8, ci 0)
    constexpr int NTILE = MT * NT;
    float4* xch = reinterpret_cast<float4*>(smem);  // [2][NTILE + 1][64] float4: tiles of two waves (+ their bias sums)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {  // bias: fold the 4 pixel lanes of a channel (lanes li, li + 16, li + 32, li + 48)
        bsum[nt] += __shfl_xor(bsum[nt], 16, 64);
        bsum[nt] += __shfl_xor(bsum[nt], 32, 64);
    }
    auto put = [&](int slot_wave) {
        float4* d = xch + (size_t)slot_wave * (NTILE + 1) * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                d[(mt * NT + nt) * 64] = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
        d[NTILE * 64] = make_float4(bsum[0], NT > 1 ? bsum[NT - 1] : 0.f, 0.f, 0.f);
    };
    auto add = [&](int slot_wave) {
        const float4* d = xch + (size_t)slot_wave * (NTILE + 1) * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 v = d[(mt * NT + nt) * 64];
                acc[mt][nt][0] += v.x; acc[mt][nt][1] += v.y; acc[mt][nt][2] += v.z; acc[mt][nt][3] += v.w;
            }
        const float4 bb = d[NTILE * 64];
        bsum[0] += bb.x;
        if (NT > 1) bsum[NT - 1] += bb.y;
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
    if (wave == 0) add(0);
    __syncthreads();
    const int combos_max = a.combos_max;
    const int img_floats = NCHUNK * combos_max * RS;
    float* img = smem;
    for (int i = threadIdx.x; i < (img_floats >> 2); i += 256) reinterpret_cast<float4*>(img)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = nt * 16 + li;
                const int cot = co >> 3, qq = co & 7;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mt * 16 + 4 * lk + r;
                    const int tap = m / CIN, ci = m - tap * CIN;
                    const int slot = tap * CO_T + qq;
                    if (co < cob && m < M) img[((slot / RS) * combos_max + ci * n_cot + cot) * RS + slot % RS] = acc[mt][nt][r];
                }
            }
        if (a.with_bias && lk == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = nt * 16 + li;
                const int slot = K * K * CO_T + (co & 7);
                if (co < cob) img[((slot / RS) * combos_max + (co >> 3)) * RS + slot % RS] = bsum[nt];
            }
        }
    }
    __syncthreads();
    float* out = a.slab + (size_t)bp.bx * img_floats;  // one part per grid.x block (grid.y == grid.z == 1)
    for (int i = threadIdx.x; i < (img_floats >> 2); i += 256)
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(img)[i];
}

template <int K, int CO_T>
__global__ __launch_bounds__(256, 2) void wgrad_s1_lds(WgradArgs a) {
    wgrad_body<K, CO_T>(a, BlockPos{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y,
                                    (int)gridDim.z});
}

// The kernel gradients of several layers in ONE launch: workgroup ranges [first[j], first[j + 1]) belong to job j.
// The layers are independent once every dy exists, so their workgroups share the chip and the staging phases of
// one layer overlap the arithmetic of another instead of every launch idling through its own.
constexpr int kMaxWgradJobs = 8;
struct WgradMultiArgs {
    int n_jobs;
    int first[kMaxWgradJobs + 1];
    int gx[kMaxWgradJobs], gy[kMaxWgradJobs], gz[kMaxWgradJobs], co_t[kMaxWgradJobs];
    WgradArgs job[kMaxWgradJobs];
};
template <int K>
__global__ __launch_bounds__(256, 2) void wgrad_s1_multi(WgradMultiArgs m) {
    int j = 0;
    while (j + 1 < m.n_jobs && (int)blockIdx.x >= m.first[j + 1]) ++j;
    const int b = blockIdx.x - m.first[j];
    BlockPos bp;
    bp.gx = m.gx[j]; bp.gy = m.gy[j]; bp.gz = m.gz[j];
    bp.bx = b % bp.gx;
    const int r = b / bp.gx;
    bp.by = r % bp.gy;
    bp.bz = r / bp.gy;
    if (m.co_t[j] == 8) wgrad_body<K, 8>(m.job[j], bp);
    else wgrad_body<K, 4>(m.job[j], bp);
}

struct WgradPlan {
    bool ok;
    int mfma;  // the matrix-core body applies (see wgrad_body_mfma)
    int S, cib, cob, PG, co_t, xplane, xwp, zplane, nblk, nchunk, combos_max;
    size_t lds_bytes, slab_floats;  // slab_floats: per grid.x block
    dim3 grid;
};

// tb: workgroups the launch should have per layer (512 = two per CU for a launch of its own; a batched launch, where
// several layers share the chip, asks for fewer, longer-lived workgroups: less fold / slab work per sample)
WgradPlan plan_wgrad(int N, int Hs, int Ws, int Cin, int Cout, int K, int UP, int tb = 512) {
    if (tb < 1) tb = 512;
    WgradPlan p{};
    p.ok = false;
    const int Ho = Hs * UP, Wo = Ws * UP;
    if (Wo & 3) return p;
    const int padl = (K - 1) / 2;
    p.co_t = (K == 3 && Cout > 4) ? 8 : 4;
    p.xwp = (Wo + 2 * padl + 3 + 3) & ~3;  // a strip reads up to col ow0 + K + 2 <= Wo - 4 + K + 2
    p.xplane = (Ho + 2 * padl) * p.xwp + 4;
    p.zplane = Ho * Wo + 4;
    int cob = ((Cout + p.co_t - 1) / p.co_t) * p.co_t;
    if (cob > 32) cob = 32;
    int cib = Cin;
    while (cib * (cob / p.co_t) > 256) cib = (cib + 1) / 2;
    auto bytes = [&](int S, int cib_, int cob_) {
        return ((size_t)S * (cib_ * p.xplane + cob_ * p.zplane)) * sizeof(float);
    };
    // shrink whichever side holds more LDS until one sample fits
    while (bytes(1, cib, cob) > kLdsBudget) {
        const bool x_bigger = (size_t)cib * p.xplane >= (size_t)cob * p.zplane;
        if (x_bigger && cib > 1) cib = (cib + 1) / 2;
        else if (cob > p.co_t) cob = ((cob / 2 + p.co_t - 1) / p.co_t) * p.co_t;
        else if (cib > 1) cib = (cib + 1) / 2;
        else return p;
    }
    const int combos = cib * (cob / p.co_t);
    int PG = 1;
    while (PG * 2 * combos <= 256 && PG < 64) PG *= 2;
    // enough strips per pass to feed PG groups a few strips each
    const int strips = Ho * (Wo / 4);
    const int ny0 = (Cin + cib - 1) / cib, nz0 = (Cout + cob - 1) / cob;
    int S = (4 * PG + strips - 1) / strips;
    if (S < 1) S = 1;
    while (S > 1 && (int64_t)((N + S - 1) / S) * ny0 * nz0 < tb) S >>= 1;
    // One pass per workgroup when the LDS allows it: the loads of all its samples are then in flight together
    // instead of one exposed global round trip per pass (512 workgroups = two per CU at this LDS budget).
    const int one_pass = (int)(((int64_t)N * ny0 * nz0 + tb - 1) / tb);
    if (one_pass > S && one_pass <= 8 && bytes(one_pass, cib, cob) <= kLdsBudget) S = one_pass;
    while (S > 1 && bytes(S, cib, cob) > kLdsBudget) --S;
    if (S > N) S = N;
    p.S = S; p.cib = cib; p.cob = cob; p.PG = PG;
    p.lds_bytes = bytes(S, cib, cob);
    if (p.lds_bytes < (size_t)256 * (kWgradRS + 1) * sizeof(float)) p.lds_bytes = (size_t)256 * (kWgradRS + 1) * sizeof(float);
    p.lds_bytes = (p.lds_bytes + 15) & ~(size_t)15;
    const int passes = (N + S - 1) / S;
    const int ny = (Cin + cib - 1) / cib, nz = (Cout + cob - 1) / cob;
    int gx = tb / (ny * nz);
    if (gx < 1) gx = 1;
    if (gx > passes) gx = passes;
    p.nblk = gx;
    p.grid = dim3((unsigned)gx, (unsigned)ny, (unsigned)nz);
    const int nslot = K * K * p.co_t + p.co_t;
    p.nchunk = (nslot + kWgradRS - 1) / kWgradRS;
    p.combos_max = combos;
    p.slab_floats = (size_t)ny * nz * p.nchunk * combos * kWgradRS;
    // matrix-core body: 3x3, all channels in one workgroup, and 16 x 32 or 32 x 16 channels: measured at batch 1024 it
    // is 1.4 - 1.5x faster than the vector body there (21 vs 32 us per layer); with 8 or 16 x 8 channels the 16-wide
    // tiles are half empty or the loop too short to hide the operand reads, and the vector body is as fast.  The slab
    // image is summed in LDS (over the dead staging buffers) before it is written, so the LDS request covers it.
    p.mfma = K == 3 && p.co_t == 8 && ny == 1 && nz == 1 && ((Cin == 16 && Cout == 32) || (Cin == 32 && Cout == 16)) &&
             (p.slab_floats & 3) == 0 && p.slab_floats * sizeof(float) <= kLdsBudget &&
             opt(OPT_WGRAD_VALU) != 1;
    if (p.mfma) {
        const size_t tiles = (size_t)((9 * Cin + 15) / 16) * (Cout > 16 ? 2 : 1);
        size_t need = 2 * (tiles + 1) * 64 * sizeof(float4);  // two waves' accumulator tiles in flight between the waves
        if (need < p.slab_floats * sizeof(float)) need = p.slab_floats * sizeof(float);
        if (need > p.lds_bytes) p.lds_bytes = (need + 15) & ~(size_t)15;
    }
    p.ok = true;
    return p;
}

size_t tiled_wgrad_workspace(int N, int Hs, int Ws, int Cin, int Cout, int K, int UP) {
    if (!((K == 3 || K == 5) && (UP == 1 || UP == 2))) return 0;
    WgradPlan p = plan_wgrad(N, Hs, Ws, Cin, Cout, K, UP);
    if (!p.ok) return 0;
    return (size_t)p.nblk * p.slab_floats * sizeof(float);
}

// Sums the slabs in a fixed order and scatters each sum to its place in dw / db.  One thread column per slab
// element (coalesced reads); 16 groups of a block split the slabs and fold through LDS.
struct ReduceArgs {
    const float* part;
    float* dw;
    float* db;
    int64_t slab_floats;
    int n_split, accumulate;
    int K2, co_t, nchunk, combos_max, ny, cib, cob, Cin, Cout;
};
__device__ __forceinline__ void wgrad_reduce_body(const ReduceArgs& a, int block, float (*red)[64]) {
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t e = (int64_t)block * 64 + lane;
    float s = 0.f;
    if (e < a.slab_floats) {
        int sp = grp;
        for (; sp + 7 * 16 < a.n_split; sp += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a.part[(int64_t)(sp + u * 16) * a.slab_floats + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; sp < a.n_split; sp += 16) s += a.part[(int64_t)sp * a.slab_floats + e];
    }
    red[grp][lane] = s;
    __syncthreads();
    if (grp != 0 || e >= a.slab_floats) return;
    // decode the slab position
    const int RS = kWgradRS;
    int r = (int)e;
    const int j = r % RS; r /= RS;
    const int combo = r % a.combos_max; r /= a.combos_max;
    const int ch = r % a.nchunk;
    const int part = r / a.nchunk;
    const int by = part % a.ny, bz = part / a.ny;
    const int ci_base = by * a.cib, co_base = bz * a.cob;
    const int cib = min(a.cib, a.Cin - ci_base), cob = min(a.cob, a.Cout - co_base);
    const int n_cot = (cob + a.co_t - 1) / a.co_t;
    if (combo >= cib * n_cot) return;
    const int cil = combo / n_cot, ct = combo - cil * n_cot;
    const int slot = ch * RS + j;
    float* o;
    if (slot < a.K2 * a.co_t) {
        const int tap = slot / a.co_t, q = slot - tap * a.co_t;
        const int co = co_base + ct * a.co_t + q;
        if (co >= a.Cout) return;
        o = a.dw + ((size_t)tap * a.Cin + ci_base + cil) * a.Cout + co;
    } else if (slot < a.K2 * a.co_t + a.co_t) {
        const int co = co_base + ct * a.co_t + (slot - a.K2 * a.co_t);
        if (a.db == nullptr || cil != 0 || by != 0 || co >= a.Cout) return;
        o = a.db + co;
    } else {
        return;
    }
    float t = a.accumulate ? *o : 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][lane];
    *o = t;
}

__global__ __launch_bounds__(1024) void wgrad_reduce3(ReduceArgs a) {
    __shared__ float red[16][64];
    wgrad_reduce_body(a, blockIdx.x, red);
}

// the same reduction for several layers in one launch: block ranges [first[j], first[j + 1]) belong to job j
constexpr int kMaxReduceJobs = 8;
struct ReduceMultiArgs {
    int n_jobs;
    int first[kMaxReduceJobs + 1];
    ReduceArgs job[kMaxReduceJobs];
};
__global__ __launch_bounds__(1024) void wgrad_reduce_multi(ReduceMultiArgs m) {
    __shared__ float red[16][64];
    int j = 0;
    while (j + 1 < m.n_jobs && (int)blockIdx.x >= m.first[j + 1]) ++j;
    wgrad_reduce_body(m.job[j], blockIdx.x - m.first[j], red);
}

bool tiled_wgrad_supported(int N, int Hs, int Ws, int Cin, int Cout, int K, int UP) {
    if (!((K == 3 || K == 5) && (UP == 1 || UP == 2))) return false;
    return plan_wgrad(N, Hs, Ws, Cin, Cout, K, UP).ok;
}

// Returns 1 when the shape is not covered, 0 on success, <0 on error.  db != null adds the bias gradient.
static void fill_reduce_args(ReduceArgs& r, const WgradPlan& p, const void* ws, float* dw, float* db, int accumulate, int K,
                             int Cin, int Cout) {
    r.part = (const float*)ws; r.dw = dw; r.db = db; r.slab_floats = (int64_t)p.slab_floats; r.n_split = p.nblk;
    r.accumulate = accumulate; r.K2 = K * K; r.co_t = p.co_t; r.nchunk = p.nchunk; r.combos_max = p.combos_max;
    r.ny = (int)p.grid.y; r.cib = p.cib; r.cob = p.cob; r.Cin = Cin; r.Cout = Cout;
}

// One launch that finishes up to kMaxReduceJobs deferred kernel-gradient reductions (slabs written with defer = 1).
int launch_wgrad_finish(const sv_wgrad_job* jobs, int n_jobs, hipStream_t st) {
    ReduceMultiArgs m{};
    m.n_jobs = n_jobs;
    int blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const sv_conv_geom& g = jobs[j].g;
        WgradPlan p = plan_wgrad(g.N, g.H / g.up_h, g.W / g.up_w, g.Cin, g.Cout, g.KH, g.up_h, jobs[j].blocks);
        if (!p.ok) return fail(SV_E_UNSUPPORTED, "conv2d_wgrad_finish: job %d is outside the tiled family", j);
        fill_reduce_args(m.job[j], p, jobs[j].workspace, (float*)jobs[j].dw, (float*)jobs[j].db, jobs[j].accumulate, g.KH,
                         g.Cin, g.Cout);
        m.first[j] = blocks;
        blocks += (int)ceil_div((int64_t)p.slab_floats, 64);
    }
    m.first[n_jobs] = blocks;
    hipLaunchKernelGGL(wgrad_reduce_multi, dim3((unsigned)blocks), dim3(1024), 0, st, m);
    return check_launch("wgrad_reduce_multi");
}

static void fill_wgrad_args(WgradArgs& a, const WgradPlan& p, const float* x, const float* dz, void* ws, int N, int Hs,
                            int Ws, int Cin, int Cout, int UP, int zmode, const float* zy, const uint8_t* zidx, int zact,
                            float zalpha, int with_bias) {
    a.x = x; a.dz = dz; a.slab = (float*)ws;
    a.N = N; a.Hs = Hs; a.Ws = Ws; a.Cin = Cin; a.Cout = Cout; a.Ho = Hs * UP; a.Wo = Ws * UP; a.up = UP;
    a.S = p.S; a.cib = p.cib; a.cob = p.cob; a.PG = p.PG; a.xplane = p.xplane; a.xwp = p.xwp; a.zplane = p.zplane;
    a.zmode = zmode; a.zy = zy; a.zidx = zidx; a.zk = act_grad_coef(zact, zalpha); a.with_bias = with_bias;
    a.combos_max = p.combos_max;
    a.mfma = p.mfma;
}

// Kernel (+ bias) gradients of up to kMaxWgradJobs layers: one launch for the partial sums, one for the reductions.
int launch_wgrad_nhwc_batch(const sv_wgrad_batch_job* jobs, int n_jobs, hipStream_t st, sv_grad_slabs* slabs_out);  // wgrad_nhwc.hip

int launch_wgrad_batch(const sv_wgrad_batch_job* jobs, int n_jobs, int reduce, hipStream_t st) {
    if (reduce) {  // the channel-last matrix-core kernel when every layer of the batch is in its family
        const int rc = launch_wgrad_nhwc_batch(jobs, n_jobs, st, nullptr);
        if (rc <= 0) return rc;
    }
    WgradMultiArgs m{};
    ReduceMultiArgs rm{};
    m.n_jobs = rm.n_jobs = n_jobs;
    int blocks = 0, rblocks = 0;
    size_t lds = 0;
    // Workgroups are dispatched in block order and a layer's workgroups live longer the larger its images are (staging
    // dominates), so the layers go into the table longest first: the launch then ends on short workgroups instead of
    // waiting for a last round of long ones.
    int ord[kMaxWgradJobs];
    for (int j = 0; j < n_jobs; ++j) ord[j] = j;
    for (int a = 1; a < n_jobs; ++a)
        for (int b = a; b > 0; --b) {
            const sv_conv_geom &ga = jobs[ord[b]].g, &gb = jobs[ord[b - 1]].g;
            const int64_t ka = (int64_t)ga.H * ga.W * 64 + ga.Cin + ga.Cout, kb = (int64_t)gb.H * gb.W * 64 + gb.Cin + gb.Cout;
            if (ka <= kb) break;
            const int t = ord[b]; ord[b] = ord[b - 1]; ord[b - 1] = t;
        }
    for (int jj = 0; jj < n_jobs; ++jj) {
        const int j = jj;  // table slot
        const sv_wgrad_batch_job& q = jobs[ord[jj]];
        const sv_conv_geom& g = q.g;
        if (g.KH != 3) return fail(SV_E_UNSUPPORTED, "conv2d_wgrad_batch: job %d: 3x3 kernels only", ord[jj]);
        const int Hs = g.H / g.up_h, Ws = g.W / g.up_w;
        WgradPlan p = plan_wgrad(g.N, Hs, Ws, g.Cin, g.Cout, g.KH, g.up_h, q.blocks);
        if (!p.ok) return fail(SV_E_UNSUPPORTED, "conv2d_wgrad_batch: job %d is outside the tiled family", j);
        const size_t need = (size_t)p.nblk * p.slab_floats * sizeof(float);
        if (q.workspace == nullptr || q.workspace_bytes < need)
            return fail(SV_E_WORKSPACE, "conv2d_wgrad_batch: job %d: workspace %zu bytes < required %zu", j,
                        (size_t)q.workspace_bytes, need);
        fill_wgrad_args(m.job[j], p, (const float*)q.x, (const float*)q.dy, q.workspace, g.N, Hs, Ws, g.Cin, g.Cout, g.up_h,
                        q.idx != nullptr ? 2 : 1, (const float*)q.y, q.idx, q.act, q.alpha, q.db != nullptr);
        m.first[j] = blocks;
        m.gx[j] = (int)p.grid.x; m.gy[j] = (int)p.grid.y; m.gz[j] = (int)p.grid.z; m.co_t[j] = p.co_t;
        blocks += (int)(p.grid.x * p.grid.y * p.grid.z);
        if (p.lds_bytes > lds) lds = p.lds_bytes;
        fill_reduce_args(rm.job[j], p, q.workspace, (float*)q.dw, (float*)q.db, q.accumulate, g.KH, g.Cin, g.Cout);
        rm.first[j] = rblocks;
        rblocks += (int)ceil_div((int64_t)p.slab_floats, 64);
    }
    m.first[n_jobs] = blocks;
    rm.first[n_jobs] = rblocks;
    hipLaunchKernelGGL((wgrad_s1_multi<3>), dim3((unsigned)blocks), dim3(256), lds, st, m);
    int rc = check_launch("wgrad_s1_multi");
    if (rc || !reduce) return rc;
    hipLaunchKernelGGL(wgrad_reduce_multi, dim3((unsigned)rblocks), dim3(1024), 0, st, rm);
    return check_launch("wgrad_reduce_multi");
}

int launch_tiled_wgrad(const float* x, const float* dz, float* dw, float* db, int accumulate, void* ws, size_t ws_bytes,
                       int N, int Hs, int Ws, int Cin, int Cout, int K, int UP, hipStream_t st, int zmode,
                       const float* zy, const uint8_t* zidx, int zact, float zalpha, int defer, int with_bias) {
    if (!((K == 3 || K == 5) && (UP == 1 || UP == 2))) return 1;
    WgradPlan p = plan_wgrad(N, Hs, Ws, Cin, Cout, K, UP);
    if (!p.ok) return 1;
    if (zmode == 2 && (((Hs * UP) | (Ws * UP)) & 1)) return 1;
    const size_t need = (size_t)p.nblk * p.slab_floats * sizeof(float);
    if (ws == nullptr || ws_bytes < need)
        return fail(SV_E_WORKSPACE, "conv2d_wgrad: workspace %zu bytes < required %zu", ws_bytes, need);
    WgradArgs a{};
    a.x = x; a.dz = dz; a.slab = (float*)ws;
    a.N = N; a.Hs = Hs; a.Ws = Ws; a.Cin = Cin; a.Cout = Cout; a.Ho = Hs * UP; a.Wo = Ws * UP; a.up = UP;
    a.S = p.S; a.cib = p.cib; a.cob = p.cob; a.PG = p.PG; a.xplane = p.xplane; a.xwp = p.xwp; a.zplane = p.zplane;
    a.zmode = zmode; a.zy = zy; a.zidx = zidx; a.zk = act_grad_coef(zact, zalpha); a.with_bias = db != nullptr || with_bias;
    a.combos_max = p.combos_max;
    if (K == 3) {
        if (p.co_t == 8)
            hipLaunchKernelGGL((wgrad_s1_lds<3, 8>), p.grid, dim3(256), p.lds_bytes, st, a);
        else
            hipLaunchKernelGGL((wgrad_s1_lds<3, 4>), p.grid, dim3(256), p.lds_bytes, st, a);
    } else {
        hipLaunchKernelGGL((wgrad_s1_lds<5, 4>), p.grid, dim3(256), p.lds_bytes, st, a);
    }
    int rc = check_launch("wgrad_s1_lds");
    if (rc || defer) return rc;
    ReduceArgs r{};
    fill_reduce_args(r, p, ws, dw, db, accumulate, K, Cin, Cout);
    hipLaunchKernelGGL(wgrad_reduce3, dim3((unsigned)ceil_div((int64_t)p.slab_floats, 64)), dim3(1024), 0, st, r);
    return check_launch("wgrad_reduce3");
}

}  // namespace sv
