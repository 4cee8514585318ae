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
#include "common.h"

namespace sv {

enum { EPI_STORE = 0, EPI_MAXPOOL = 1, EPI_SUM4 = 2 };

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
    int lgW, lgH;  // log2 of Ws / Hs when they are powers of two, else -1
    int ks;        // input channels of one output tile are split over ks adjacent lanes (1, 2 or 4) and summed by shuffles
};

// n / d and n % d for a divisor that is usually a power of two (image extents, channel counts): lg >= 0 selects
// the shift form, lg < 0 the general division.
__device__ __forceinline__ int fdiv(int n, int d, int lg) { return lg >= 0 ? (n >> lg) : n / d; }
inline int ilog2_exact(int v) { return (v > 0 && (v & (v - 1)) == 0) ? __builtin_ctz(v) : -1; }

__host__ __device__ constexpr int floordiv_c(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// stored-patch row used by logical patch row q (q = pixel offset + tap) for pad PADL and upsampling UP
template <int UP, int PADL>
__host__ __device__ constexpr int patch_row(int q) { return floordiv_c(q - PADL, UP) - floordiv_c(-PADL, UP); }

// ---- staging helpers (nt = threads in the block) -------------------------------------------------------------
// Input chunk [ci0, ci0+cic) of samples [n0, n0+ns) -> planar LDS image inside the zero halo.
__device__ __forceinline__ void stage_input(const TiledArgs& a, float* in_lds, int n0, int ns, int ci0, int cic,
                                            int nt) {
    const int Hs = a.Hs, Ws = a.Ws;
    const int tid = threadIdx.x;
    const int C4 = cic >> 2;
    const bool vec = (cic & 3) == 0 && (a.Cin & 3) == 0 && C4 > 0 && (C4 & (C4 - 1)) == 0 && C4 <= nt;
    if (a.zmode != 2) {
        if (vec) {  // one float4 = 4 channels of one pixel
            const int cl = tid & (C4 - 1), pl = tid / C4, PP = nt / C4;
            const int HW = Hs * Ws;
            for (int p = pl; p < ns * HW; p += PP) {
                const int r = fdiv(p, Ws, a.lgW), col = p - r * Ws;
                const int ss = fdiv(r, Hs, a.lgH), row = r - ss * Hs;
                const size_t gi = ((size_t)n0 * HW + p) * a.Cin + ci0 + cl * 4;
                float4 v = *reinterpret_cast<const float4*>(a.x + gi);
                if (a.zmode == 1) {
                    const float4 yv = *reinterpret_cast<const float4*>(a.zy + gi);
                    v.x *= act_grad_q(yv.x, a.zk);
                    v.y *= act_grad_q(yv.y, a.zk);
                    v.z *= act_grad_q(yv.z, a.zk);
                    v.w *= act_grad_q(yv.w, a.zk);
                }
                float* q = in_lds + (ss * a.cic + cl * 4) * a.plane + (row + a.halo) * a.wp + col + a.halo;
                q[0] = v.x; q[a.plane] = v.y; q[2 * a.plane] = v.z; q[3 * a.plane] = v.w;
            }
        } else {
            const int elems = Hs * Ws * cic;
            for (int i = tid; i < ns * elems; i += nt) {
                const int ss = i / elems;
                const int e = i - ss * elems;
                const int p = e / cic, c = e - p * cic;
                const int row = p / Ws, col = p - row * Ws;
                const size_t gi = (((size_t)(n0 + ss) * Hs + row) * Ws + col) * a.Cin + ci0 + c;
                float v = a.x[gi];
                if (a.zmode == 1) v *= act_grad_q(a.zy[gi], a.zk);
                in_lds[(ss * a.cic + c) * a.plane + (row + a.halo) * a.wp + col + a.halo] = v;
            }
        }
    } else {  // pooled gradient: x, zy, zidx at (Hs/2, Ws/2); scatter to the argmax slot of each 2x2 window
        const int Hq = Hs >> 1, Wq = Ws >> 1;
        const int elems = Hq * Wq * cic;
        for (int i = tid; i < ns * elems; i += nt) {
            const int ss = i / elems;
            const int e = i - ss * elems;
            const int p = e / cic, c = e - p * cic;
            const int row = p / Wq, col = p - row * Wq;
            const size_t gi = (((size_t)(n0 + ss) * Hq + row) * Wq + col) * a.Cin + ci0 + c;
            const float v = a.x[gi] * act_grad_q(a.zy[gi], a.zk);
            const int k = a.zidx[gi];
            float* q = in_lds + (ss * a.cic + c) * a.plane + (2 * row + a.halo) * a.wp + 2 * col + a.halo;
            *reinterpret_cast<float2*>(q) = make_float2(k == 0 ? v : 0.f, k == 1 ? v : 0.f);
            *reinterpret_cast<float2*>(q + a.wp) = make_float2(k == 2 ? v : 0.f, k == 3 ? v : 0.f);
        }
    }
}

// Filter chunk -> w_lds[tap][c][cop] (cop = channel tiles * CO_T, zero padded).
template <int K>
__device__ __forceinline__ void stage_weights(const TiledArgs& a, float* w_lds, int ci0, int cic, int co_base, int coc,
                                              int cop, int nt) {
    const int cnt = K * K * cic * cop;
    for (int i = threadIdx.x; i < cnt; i += nt) {
        const int co = i % cop;
        const int r = i / cop;
        const int c = r % cic, tap = r / cic;
        float v = 0.f;
        if (co < coc) {
            const int kh = tap / K, kw = tap - kh * K;
            if (a.wflip)  // stored [K][K][Cout_launch][Cin_launch], taps mirrored
                v = a.w[((size_t)((K - 1 - kh) * K + (K - 1 - kw)) * a.Cout + co_base + co) * a.Cin + ci0 + c];
            else
                v = a.w[((size_t)tap * a.Cin + ci0 + c) * a.Cout + co_base + co];
        }
        w_lds[(tap * a.cic + c) * cop + co] = v;
    }
}

template <int K, int UP, int CO_T, int EPI>
__global__ __launch_bounds__(256) void conv_s1_lds(TiledArgs a) {
    constexpr int PADL = (K - 1) / 2;
    constexpr int NPR = patch_row<UP, PADL>(K) + 1;  // patch rows == patch cols
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

    // zero once: the halo, the row padding and unused planes are never overwritten afterwards
    for (int i = threadIdx.x; i < a.S * a.cic * a.plane; i += nt) in_lds[i] = 0.f;
    if (one_chunk) stage_weights<K>(a, w_lds, 0, a.Cin, co_base, coc, cop, nt);

    for (int n0 = blockIdx.x * a.S; n0 < a.N; n0 += gridDim.x * a.S) {
        const int ns = min(a.S, a.N - n0);
        // A thread owns one work item per pass of nt; its accumulators live across the ci-chunk loop.  When a
        // sample pass has more than nt items the passes are looped and each re-walks (re-stages) the chunks.
        for (int item0 = 0; item0 < total; item0 += nt) {
            const int item = item0 + threadIdx.x;
            const bool live = item < total;
            int cot = 0, s = 0, ty = 0, tx = 0, kl = 0;
            if (live) {
                cot = item / per_cot;
                int r = item - cot * per_cot;
                kl = r % ks;  // which slice of the input channels this lane sums
                r /= ks;
                s = r / T;
                const int t = r - s * T;
                ty = t / tiles_x;
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
                __syncthreads();  // previous chunk / item pass is done with LDS (and the zero fill has landed)
                if (item0 == 0 || !one_chunk) stage_input(a, in_lds, n0, ns, ci0, cic, nt);
                if (!one_chunk) stage_weights<K>(a, w_lds, ci0, cic, co_base, coc, cop, nt);
                __syncthreads();
                if (work) {
                    const float* img = in_lds + (size_t)s * a.cic * a.plane;
                    // patch origin in the padded stored image (even column when UP == 1: 8-byte aligned rows)
                    const int r0 = (UP == 1 ? 2 * ty : ty) + a.halo + floordiv_c(-PADL, UP);
                    const int c0 = (UP == 1 ? 2 * tx : tx) + a.halo + floordiv_c(-PADL, UP);
                    const float* wbase = w_lds + cot * CO_T;
                    for (int c = kl; c < cic; c += ks) {
                        float patch[NPR][NPR + 1];
                        const float* pl = img + c * a.plane + r0 * a.wp + c0;
#pragma unroll
                        for (int pr = 0; pr < NPR; ++pr) {
                            if (UP == 1) {  // NPR = K + 1 is even
#pragma unroll
                                for (int pc = 0; pc < NPR; pc += 2) {
                                    const float2 v2 = *reinterpret_cast<const float2*>(pl + pr * a.wp + pc);
                                    patch[pr][pc] = v2.x;
                                    patch[pr][pc + 1] = v2.y;
                                }
                            } else {
#pragma unroll
                                for (int pc = 0; pc < NPR; ++pc) patch[pr][pc] = pl[pr * a.wp + pc];
                            }
                        }
#pragma unroll
                        for (int kh = 0; kh < K; ++kh)
#pragma unroll
                            for (int kw = 0; kw < K; ++kw) {
                                float wv[CO_T];
                                const float* wp = wbase + ((kh * K + kw) * a.cic + c) * cop;
#pragma unroll
                                for (int q = 0; q < CO_T; q += 4) {
                                    const float4 w4 = *reinterpret_cast<const float4*>(wp + q);
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
            }
            // ---- fold the ks channel slices (adjacent lanes of one wave; dead lanes hold zeros)
            for (int o = 1; o < ks; o <<= 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < CO_T; ++q) acc[i][j][q] += __shfl_xor(acc[i][j][q], o, 64);
            }
            // ---- epilogue
            if (work && kl == 0) {
                const int n = n0 + s;
                const int cg = co_base + cot * CO_T;  // first global output channel of this thread
                if (EPI == EPI_STORE) {
                    const bool vec = (a.Cout & 3) == 0;  // cg is a multiple of 4: 16-byte aligned rows of channels
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            float* yp = a.y + (((size_t)n * a.Ho + 2 * ty + i) * a.Wo + 2 * tx + j) * a.Cout + cg;
                            float v[CO_T];
#pragma unroll
                            for (int q = 0; q < CO_T; ++q) {
                                const float b = (a.bias != nullptr && cg + q < a.Cout) ? a.bias[cg + q] : 0.f;
                                v[q] = act_apply(acc[i][j][q] + b, a.act, a.alpha);
                            }
                            if (vec) {
#pragma unroll
                                for (int q = 0; q < CO_T; q += 4)
                                    if (cg + q < a.Cout)
                                        *reinterpret_cast<float4*>(yp + q) = make_float4(v[q], v[q + 1], v[q + 2], v[q + 3]);
                            } else {
#pragma unroll
                                for (int q = 0; q < CO_T; ++q)
                                    if (cg + q < a.Cout) yp[q] = v[q];
                            }
                        }
                } else if (EPI == EPI_MAXPOOL) {
                    // max of the pre-activations first (every activation here is non-decreasing, so the maximum and -
                    // wherever the activation is strictly increasing - its position are those of the activated
                    // values), then ONE activation per window.  relu flattens negatives into ties at 0 whose "first
                    // maximum" is slot 0 in the unfused order; no gradient flows there (relu'(0) = 0), so only the
                    // recorded slot could differ, and it is made to agree below.
                    const size_t o = (((size_t)n * tiles_y + ty) * tiles_x + tx) * a.Cout + cg;
#pragma unroll
                    for (int q = 0; q < CO_T; ++q)
                        if (cg + q < a.Cout) {
                            const float b = a.bias != nullptr ? a.bias[cg + q] : 0.f;
                            float best = acc[0][0][q];
                            int bi = 0;
                            if (acc[0][1][q] > best) { best = acc[0][1][q]; bi = 1; }
                            if (acc[1][0][q] > best) { best = acc[1][0][q]; bi = 2; }
                            if (acc[1][1][q] > best) { best = acc[1][1][q]; bi = 3; }
                            const float v = act_apply(best + b, a.act, a.alpha);
                            if (a.act == SV_ACT_RELU && v <= 0.f) bi = 0;
                            a.y[o + q] = v;
                            if (a.idx != nullptr) a.idx[o + q] = (uint8_t)bi;
                        }
                } else {  // EPI_SUM4
                    const size_t o = (((size_t)n * tiles_y + ty) * tiles_x + tx) * a.Cout + cg;
#pragma unroll
                    for (int q = 0; q < CO_T; ++q)
                        if (cg + q < a.Cout) a.y[o + q] = (acc[0][0][q] + acc[0][1][q]) + (acc[1][0][q] + acc[1][1][q]);
                }
            }
        }
    }
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
    a.lgW = ilog2_exact(Ws); a.lgH = ilog2_exact(Hs); a.ks = p.ks;
    if (K == 3) return UP == 1 ? launch_cot<3, 1>(a, epi, p, st) : launch_cot<3, 2>(a, epi, p, st);
    return UP == 1 ? launch_cot<5, 1>(a, epi, p, st) : launch_cot<5, 2>(a, epi, p, st);
}

}  // namespace sv

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
    float* slab;      // [gridDim.x][K*K*Cin*Cout]
    int N, Hs, Ws, Cin, Cout, Ho, Wo, up;
    int S, cib, cob;  // samples per pass; input channels per block (grid.y); output channels per block (grid.z)
    int PG;           // pixel groups (power of two <= 64)
    int xplane, xwp, zplane;
    // dz formed on the fly, as in TiledArgs: 0 = dz given; 1 = dz*act'(zy); 2 = pooled (dz, zy, zidx at Ho/2 x Wo/2)
    const float* zy;
    const uint8_t* zidx;
    int zmode;
    ActGradCoef zk;
    int lgWs, lgHs, lgHWo;  // log2 of Ws, Hs, Ho*Wo when powers of two, else -1
    int with_bias;    // also emit sum(dz) per output channel at slab offset K*K*Cin*Cout
};

template <int K, int CO_T>
__global__ __launch_bounds__(256) void wgrad_s1_lds(WgradArgs a) {
    constexpr int PADL = (K - 1) / 2;
    constexpr int XC = K + 3;  // x columns needed by a strip of 4 output pixels
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ci_base = blockIdx.y * a.cib;
    const int cib = min(a.cib, a.Cin - ci_base);
    const int co_base = blockIdx.z * a.cob;
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

    float acc[K * K][CO_T];
    float accb[CO_T];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int q = 0; q < CO_T; ++q) acc[t][q] = 0.f;
#pragma unroll
    for (int q = 0; q < CO_T; ++q) accb[q] = 0.f;
    const bool do_bias = a.with_bias && live && cil == 0 && blockIdx.y == 0;

    // zero once: halo, row padding and padded channels never get overwritten
    const int lds_floats = a.S * (a.cib * a.xplane + n_cot * CO_T * a.zplane);
    for (int i = threadIdx.x; i < lds_floats; i += 256) smem[i] = 0.f;

    for (int n0 = blockIdx.x * a.S; n0 < a.N; n0 += gridDim.x * a.S) {
        const int ns = min(a.S, a.N - n0);
        __syncthreads();
        {  // stage x (replicated to the logical resolution when up > 1)
            const int tid = threadIdx.x;
            const int Hl = a.Hs * a.up, Wl = a.Ws * a.up;
            const int C4 = cib >> 2;
            if ((cib & 3) == 0 && (a.Cin & 3) == 0 && C4 > 0 && (C4 & (C4 - 1)) == 0) {
                const int cl = tid & (C4 - 1), pl = tid / C4, PP = 256 / C4;
                const int HW = a.Hs * a.Ws;
                for (int p = pl; p < ns * HW; p += PP) {
                    const int r = fdiv(p, a.Ws, a.lgWs), col = p - r * a.Ws;
                    const int ss = fdiv(r, a.Hs, a.lgHs), row = r - ss * a.Hs;
                    const float4 v = *reinterpret_cast<const float4*>(a.x + ((size_t)n0 * HW + p) * a.Cin + ci_base + cl * 4);
                    float* q = x_lds + (ss * a.cib + cl * 4) * a.xplane + (row * a.up + PADL) * a.xwp + col * a.up + PADL;
                    for (int u = 0; u < a.up; ++u)
                        for (int t = 0; t < a.up; ++t) {
                            float* qq = q + u * a.xwp + t;
                            qq[0] = v.x; qq[a.xplane] = v.y; qq[2 * a.xplane] = v.z; qq[3 * a.xplane] = v.w;
                        }
                }
            } else {
                const int elems = Hl * Wl * cib;
                for (int i = tid; i < ns * elems; i += 256) {
                    const int ss = i / elems;
                    const int e = i - ss * elems;
                    const int p = e / cib, c = e - p * cib;
                    const int row = p / Wl, col = p - row * Wl;
                    const float v = a.x[(((size_t)(n0 + ss) * a.Hs + row / a.up) * a.Ws + col / a.up) * a.Cin + ci_base + c];
                    x_lds[(ss * a.cib + c) * a.xplane + (row + PADL) * a.xwp + col + PADL] = v;
                }
            }
            // dz -> planar [co][Ho*Wo]
            const int Z4 = cob >> 2;
            const bool zvec = (cob & 3) == 0 && (a.Cout & 3) == 0 && Z4 > 0 && (Z4 & (Z4 - 1)) == 0;
            const int zs = n_cot * CO_T;  // planes per sample
            if (a.zmode != 2) {
                const int HW = a.Ho * a.Wo;
                if (zvec) {
                    const int cl = tid & (Z4 - 1), pl = tid / Z4, PP = 256 / Z4;
                    for (int p = pl; p < ns * HW; p += PP) {
                        const int ss = fdiv(p, HW, a.lgHWo), pp = p - ss * HW;
                        const size_t gi = ((size_t)n0 * HW + p) * a.Cout + co_base + cl * 4;
                        float4 v = *reinterpret_cast<const float4*>(a.dz + gi);
                        if (a.zmode == 1) {
                            const float4 yv = *reinterpret_cast<const float4*>(a.zy + gi);
                            v.x *= act_grad_q(yv.x, a.zk);
                            v.y *= act_grad_q(yv.y, a.zk);
                            v.z *= act_grad_q(yv.z, a.zk);
                            v.w *= act_grad_q(yv.w, a.zk);
                        }
                        float* q = z_lds + (ss * zs + cl * 4) * a.zplane + pp;
                        q[0] = v.x; q[a.zplane] = v.y; q[2 * a.zplane] = v.z; q[3 * a.zplane] = v.w;
                    }
                } else {
                    const int zel = HW * cob;
                    for (int i = tid; i < ns * zel; i += 256) {
                        const int ss = i / zel;
                        const int e = i - ss * zel;
                        const int p = e / cob, c = e - p * cob;
                        const size_t gi = ((size_t)(n0 + ss) * HW + p) * a.Cout + co_base + c;
                        float v = a.dz[gi];
                        if (a.zmode == 1) v *= act_grad_q(a.zy[gi], a.zk);
                        z_lds[(ss * zs + c) * a.zplane + p] = v;
                    }
                }
            } else {
                const int Hq = a.Ho >> 1, Wq = a.Wo >> 1;
                const int zel = Hq * Wq * cob;
                for (int i = tid; i < ns * zel; i += 256) {
                    const int ss = i / zel;
                    const int e = i - ss * zel;
                    const int p = e / cob, c = e - p * cob;
                    const int row = p / Wq, col = p - row * Wq;
                    const size_t gi = ((size_t)(n0 + ss) * Hq * Wq + p) * a.Cout + co_base + c;
                    const float v = a.dz[gi] * act_grad_q(a.zy[gi], a.zk);
                    const int k = a.zidx[gi];
                    float* q = z_lds + (ss * zs + c) * a.zplane + (2 * row) * a.Wo + 2 * col;
                    *reinterpret_cast<float2*>(q) = make_float2(k == 0 ? v : 0.f, k == 1 ? v : 0.f);
                    *reinterpret_cast<float2*>(q + a.Wo) = make_float2(k == 2 ? v : 0.f, k == 3 ? v : 0.f);
                }
            }
        }
        __syncthreads();
        if (live) {
            for (int w = pg; w < ns * strips; w += PG) {
                const int ss = w / strips;
                const int st = w - ss * strips;
                const int oh = st / strips_x;
                const int ow0 = (st - oh * strips_x) << 2;
                // x rows oh .. oh+K-1 and cols ow0 .. ow0+K+2 of the padded image
                const float* xp = x_lds + (ss * a.cib + cil) * a.xplane + oh * a.xwp + ow0;
                float xv[K][XC];
#pragma unroll
                for (int r = 0; r < K; ++r) {
                    const float4 v4 = *reinterpret_cast<const float4*>(xp + r * a.xwp);
                    xv[r][0] = v4.x; xv[r][1] = v4.y; xv[r][2] = v4.z; xv[r][3] = v4.w;
#pragma unroll
                    for (int c = 4; c < XC; ++c) xv[r][c] = xp[r * a.xwp + c];
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
    // ---- fold the PG partial sums of every (ci, co tile) through LDS (the staging buffers are dead now) and
    // write this workgroup's slab.  Slots = K*K*CO_T kernel-gradient sums (+ CO_T bias sums), folded in chunks of
    // RS slots so the scratch stays within the staging footprint: red[thread][RS + 1].
    constexpr int NSLOT = K * K * CO_T + CO_T;
    constexpr int RS = 20;
    constexpr int NCHUNK = (NSLOT + RS - 1) / RS;
    const size_t kc = (size_t)K * K * a.Cin * a.Cout;
    const size_t slab_stride = kc + (a.with_bias ? a.Cout : 0);
    float* out = a.slab + (size_t)blockIdx.x * slab_stride;
    float* red = smem;  // 256 * (RS + 1) floats = 21 KiB <= any staging footprint the planner accepts
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
            const int slot = ch * RS + j;
            if (slot >= NSLOT) continue;
            float sum = 0.f;
            for (int g = 0; g < PG; ++g) sum += red[(cb * PG + g) * (RS + 1) + j];
            const int ci_l = cb / n_cot, ct = cb - ci_l * n_cot;
            if (slot < K * K * CO_T) {
                const int tap = slot / CO_T, q = slot - tap * CO_T;
                const int co = co_base + ct * CO_T + q;
                if (co < a.Cout) out[((size_t)tap * a.Cin + ci_base + ci_l) * a.Cout + co] = sum;
            } else if (a.with_bias && ci_l == 0 && blockIdx.y == 0) {
                const int co = co_base + ct * CO_T + (slot - K * K * CO_T);
                if (co < a.Cout) out[kc + co] = sum;
            }
        }
    }
}

struct WgradPlan {
    bool ok;
    int S, cib, cob, PG, co_t, xplane, xwp, zplane, nblk;
    size_t lds_bytes;
    dim3 grid;
};

WgradPlan plan_wgrad(int N, int Hs, int Ws, int Cin, int Cout, int K, int UP) {
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
    while (S > 1 && (int64_t)((N + S - 1) / S) * ny0 * nz0 < 512) S >>= 1;
    while (S > 1 && bytes(S, cib, cob) > kLdsBudget) --S;
    if (S > N) S = N;
    p.S = S; p.cib = cib; p.cob = cob; p.PG = PG;
    p.lds_bytes = bytes(S, cib, cob);
    if (p.lds_bytes < (size_t)256 * 21 * sizeof(float)) p.lds_bytes = (size_t)256 * 21 * sizeof(float);  // fold scratch
    const int passes = (N + S - 1) / S;
    const int ny = (Cin + cib - 1) / cib, nz = (Cout + cob - 1) / cob;
    int gx = 512 / (ny * nz);
    if (gx < 1) gx = 1;
    if (gx > passes) gx = passes;
    p.nblk = gx;
    p.grid = dim3((unsigned)gx, (unsigned)ny, (unsigned)nz);
    p.ok = true;
    return p;
}

size_t tiled_wgrad_workspace(int N, int Hs, int Ws, int Cin, int Cout, int K, int UP) {
    if (!((K == 3 || K == 5) && (UP == 1 || UP == 2))) return 0;
    WgradPlan p = plan_wgrad(N, Hs, Ws, Cin, Cout, K, UP);
    if (!p.ok) return 0;
    return (size_t)p.nblk * ((size_t)K * K * Cin * Cout + Cout) * sizeof(float);
}

// out[i] (+)= sum_sp part[sp][i] for i < kc (dw) and, when db != null, db[i - kc] for kc <= i < kc + ncb.
__global__ __launch_bounds__(1024) void wgrad_reduce2(const float* __restrict__ part, float* __restrict__ dw,
                                                       float* __restrict__ db, int64_t kc, int ncb, int n_split,
                                                       int accumulate) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t stride = kc + ncb;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (i < stride) {
        int sp = grp;
        for (; sp + 7 * 16 < n_split; sp += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(sp + u * 16) * stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; sp < n_split; sp += 16) s += part[(int64_t)sp * stride + i];
    }
    red[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && i < stride) {
        float* o = i < kc ? dw + i : db + (i - kc);
        float t = accumulate ? *o : 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][lane];
        *o = t;
    }
}

bool tiled_wgrad_supported(int N, int Hs, int Ws, int Cin, int Cout, int K, int UP) {
    if (!((K == 3 || K == 5) && (UP == 1 || UP == 2))) return false;
    return plan_wgrad(N, Hs, Ws, Cin, Cout, K, UP).ok;
}

// Returns 1 when the shape is not covered, 0 on success, <0 on error.  db != null adds the bias gradient.
int launch_tiled_wgrad(const float* x, const float* dz, float* dw, float* db, int accumulate, void* ws, size_t ws_bytes,
                       int N, int Hs, int Ws, int Cin, int Cout, int K, int UP, hipStream_t st, int zmode,
                       const float* zy, const uint8_t* zidx, int zact, float zalpha) {
    if (!((K == 3 || K == 5) && (UP == 1 || UP == 2))) return 1;
    WgradPlan p = plan_wgrad(N, Hs, Ws, Cin, Cout, K, UP);
    if (!p.ok) return 1;
    const size_t kc = (size_t)K * K * Cin * Cout;
    const int ncb = db != nullptr ? Cout : 0;
    const size_t need = (size_t)p.nblk * (kc + ncb) * sizeof(float);
    if (ws == nullptr || ws_bytes < need)
        return fail(SV_E_WORKSPACE, "conv2d_wgrad: workspace %zu bytes < required %zu", ws_bytes, need);
    WgradArgs a{};
    a.x = x; a.dz = dz; a.slab = (float*)ws;
    a.N = N; a.Hs = Hs; a.Ws = Ws; a.Cin = Cin; a.Cout = Cout; a.Ho = Hs * UP; a.Wo = Ws * UP; a.up = UP;
    a.S = p.S; a.cib = p.cib; a.cob = p.cob; a.PG = p.PG; a.xplane = p.xplane; a.xwp = p.xwp; a.zplane = p.zplane;
    a.zmode = zmode; a.zy = zy; a.zidx = zidx; a.zk = act_grad_coef(zact, zalpha); a.with_bias = db != nullptr;
    a.lgWs = ilog2_exact(Ws); a.lgHs = ilog2_exact(Hs); a.lgHWo = ilog2_exact(a.Ho * a.Wo);
    if (K == 3) {
        if (p.co_t == 8)
            hipLaunchKernelGGL((wgrad_s1_lds<3, 8>), p.grid, dim3(256), p.lds_bytes, st, a);
        else
            hipLaunchKernelGGL((wgrad_s1_lds<3, 4>), p.grid, dim3(256), p.lds_bytes, st, a);
    } else {
        hipLaunchKernelGGL((wgrad_s1_lds<5, 4>), p.grid, dim3(256), p.lds_bytes, st, a);
    }
    int rc = check_launch("wgrad_s1_lds");
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_reduce2, dim3((unsigned)ceil_div((int64_t)(kc + ncb), 64)), dim3(1024), 0, st,
                       (const float*)ws, dw, db, (int64_t)kc, ncb, p.nblk, accumulate);
    return check_launch("wgrad_reduce2");
}

}  // namespace sv
