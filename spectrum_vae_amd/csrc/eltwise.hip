// Bandwidth-bound kernels: activations, pooling, resampling, batch normalisation, concat, dropout,
// batching helpers.  All f32, NHWC, grid-stride with 16-byte accesses where the channel count allows.
#include <stdarg.h>

#include <initializer_list>

#include "common.h"

namespace sv {

// ------------------------------------------------------------------------------------------------ activations
// 16 bytes of activations: 4 floats or 8 bf16
template <typename T>
struct Pack16 {
    static constexpr int N = 16 / sizeof(T);
    T v[N];
};

template <typename T>
__global__ void act_fwd_k(const T* __restrict__ x, T* __restrict__ y, int64_t n, int act, float alpha) {
    constexpr int VN = Pack16<T>::N;
    const int64_t nv = n / VN;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        Pack16<T> v = reinterpret_cast<const Pack16<T>*>(x)[i];
#pragma unroll
        for (int e = 0; e < VN; ++e) v.v[e] = (T)act_apply((float)v.v[e], act, alpha);
        reinterpret_cast<Pack16<T>*>(y)[i] = v;
    }
    for (int64_t i = nv * VN + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = (T)act_apply((float)x[i], act, alpha);
}

// ---- vectorised row sweeps (the column-reduction kernels below when C is a multiple of a 16-byte piece) -------------
// A thread owns VN = 16 / sizeof(T) consecutive channels - one 16-byte piece per row - of every rif-th row of the block's
// row range (rif = 256 / (C / VN) rows in flight); NQ running sums per channel are folded over the row phases through
// LDS (NQ * 256 * VN floats) and leave as part[(block * NQ + q) * C + c].  Two bytes per lane (bf16 through the scalar
// paths) leave most of a wave's load width unused; this form moves 16 bytes per lane for both storage types.
template <typename T>
__device__ __forceinline__ void ld16(const T* __restrict__ p, float (&v)[16 / sizeof(T)]) {
    const Pack16<T> q = *reinterpret_cast<const Pack16<T>*>(p);
#pragma unroll
    for (int e = 0; e < Pack16<T>::N; ++e) v[e] = (float)q.v[e];
}

template <typename T, int NQ, typename Op>
__device__ __forceinline__ void row_sweep_vec(Op& op, float* __restrict__ part, int64_t r0, int64_t r1, int C,
                                              float* __restrict__ sm) {
    constexpr int VN = 16 / sizeof(T);
    const int cgs = C / VN, rif = blockDim.x / cgs;
    const int cg = threadIdx.x % cgs, rr = threadIdx.x / cgs;
    float acc[NQ][VN];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[q][e] = 0.f;
    if (rr < rif) {
        op.begin(cg * VN);
#pragma unroll 2
        for (int64_t r = r0 + rr; r < r1; r += rif) op.row(r * C + cg * VN, acc);
    }
    if (part == nullptr) return;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < VN; ++e) sm[((int64_t)q * blockDim.x + threadIdx.x) * VN + e] = acc[q][e];  // 0 for idle lanes
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / VN, e = c - g * VN;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float t = 0.f;
            for (int k = 0; k < rif; ++k) t += sm[((int64_t)q * blockDim.x + k * cgs + g) * VN + e];
            part[((int64_t)blockIdx.x * NQ + q) * C + c] = t;
        }
    }
}

template <typename T>
struct ActBwdOp {
    static constexpr int VN = 16 / sizeof(T);
    const T* dy;
    const T* y;
    T* dz;
    int act;
    float alpha;
    __device__ __forceinline__ void begin(int) {}
    __device__ __forceinline__ void row(int64_t i, float (&acc)[1][VN]) {
        float g[VN], yv[VN];
        ld16(dy + i, g);
        ld16(y + i, yv);
        Pack16<T> o;
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            o.v[e] = (T)(g[e] * act_grad_from_y(yv[e], act, alpha));
            acc[0][e] += (float)o.v[e];
        }
        *reinterpret_cast<Pack16<T>*>(dz + i) = o;
    }
};

template <typename T>
struct ColsumOp {
    static constexpr int VN = 16 / sizeof(T);
    const T* z;
    __device__ __forceinline__ void begin(int) {}
    __device__ __forceinline__ void row(int64_t i, float (&acc)[1][VN]) {
        float v[VN];
        ld16(z + i, v);
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[0][e] += v[e];
    }
};

// dz = dy * act'(y); optional per-channel column sums of dz into part[block][C] (C <= 1024).
// Rows are dealt to blocks in contiguous chunks; thread t of a block owns channel (t % C) of rows t/C, ...
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_k(const T* __restrict__ dy, const T* __restrict__ y,
                                                  T* __restrict__ dz, float* __restrict__ part, int64_t rows,
                                                  int C, int act, float alpha, int64_t rows_per_block, int vec) {
    extern __shared__ float sm[];  // [rows_in_flight][C] partial sums when part != nullptr
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    if (vec) {
        ActBwdOp<T> op{dy, y, dz, act, alpha};
        row_sweep_vec<T, 1>(op, part, r0, r1, C, sm);
        return;
    }
    const int rif = blockDim.x / C > 0 ? blockDim.x / C : 1;  // rows handled per sweep (C <= 256) else 1
    if (C <= (int)blockDim.x) {
        const int c = threadIdx.x % C;
        const int rr = threadIdx.x / C;
        float acc = 0.f;
        if (rr < rif) {
#pragma unroll 4
            for (int64_t r = r0 + rr; r < r1; r += rif) {
                const int64_t i = r * C + c;
                const T gt = (T)((float)dy[i] * act_grad_from_y((float)y[i], act, alpha));
                dz[i] = gt;
                acc += (float)gt;  // the bias gradient sums what the next kernel will read
            }
        }
        if (part != nullptr) {
            sm[threadIdx.x] = (rr < rif) ? acc : 0.f;
            __syncthreads();
            if (threadIdx.x < C) {
                float s = 0.f;
                for (int k = 0; k < rif; ++k) s += sm[k * C + threadIdx.x];
                part[(int64_t)blockIdx.x * C + threadIdx.x] = s;
            }
        }
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float acc = 0.f;
#pragma unroll 4
            for (int64_t r = r0; r < r1; ++r) {
                const int64_t i = r * C + c;
                const T gt = (T)((float)dy[i] * act_grad_from_y((float)y[i], act, alpha));
                dz[i] = gt;
                acc += (float)gt;
            }
            if (part != nullptr) part[(int64_t)blockIdx.x * C + c] = acc;
        }
    }
}

// column sums only
template <typename T>
__global__ __launch_bounds__(256) void colsum_k(const T* __restrict__ z, float* __restrict__ part, int64_t rows,
                                                 int C, int64_t rows_per_block, int vec) {
    extern __shared__ float sm[];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    if (vec) {
        ColsumOp<T> op{z};
        row_sweep_vec<T, 1>(op, part, r0, r1, C, sm);
        return;
    }
    if (C <= (int)blockDim.x) {
        const int rif = blockDim.x / C;
        const int c = threadIdx.x % C, rr = threadIdx.x / C;
        float acc = 0.f;
        if (rr < rif)
#pragma unroll 4
            for (int64_t r = r0 + rr; r < r1; r += rif) acc += (float)z[r * C + c];
        sm[threadIdx.x] = (rr < rif) ? acc : 0.f;
        __syncthreads();
        if (threadIdx.x < C) {
            float s = 0.f;
            for (int k = 0; k < rif; ++k) s += sm[k * C + threadIdx.x];
            part[(int64_t)blockIdx.x * C + threadIdx.x] = s;
        }
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float acc = 0.f;
#pragma unroll 4
            for (int64_t r = r0; r < r1; ++r) acc += (float)z[r * C + c];
            part[(int64_t)blockIdx.x * C + c] = acc;
        }
    }
}

// Column sums of per-block partials, part[(b * NV + v) * C + c]: a workgroup of 1024 owns 64 channels, its 16 waves walk
// every 16th block (8 loads in flight each) and the sixteen sums are combined in a fixed order - latency-parallel (the
// walk is a chain of dependent round trips, so its length is what a fold costs) and bitwise reproducible.  Returns true for
// the lanes (wave 0, c < C) that hold the result in out[].
constexpr int kFoldWaves = 16;
template <int NV, typename T>
__device__ __forceinline__ bool fold_cols(const float* __restrict__ part, int nblocks, int C, int& c, T (&out)[NV]) {
    __shared__ T red[kFoldWaves - 1][NV][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    c = blockIdx.x * 64 + lane;
    T s[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) s[v] = (T)0;
    if (c < C) {
        int b = grp;
        for (; b + 7 * kFoldWaves < nblocks; b += 8 * kFoldWaves) {
            float t[8][NV];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int v = 0; v < NV; ++v) t[u][v] = part[((int64_t)(b + kFoldWaves * u) * NV + v) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int v = 0; v < NV; ++v) s[v] += (T)t[u][v];
        }
        for (; b < nblocks; b += kFoldWaves)
#pragma unroll
            for (int v = 0; v < NV; ++v) s[v] += (T)part[((int64_t)b * NV + v) * C + c];
    }
    if (grp > 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) red[grp - 1][v][lane] = s[v];
    }
    __syncthreads();
    if (grp != 0 || c >= C) return false;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        T t = s[v];
#pragma unroll
        for (int g = 0; g < kFoldWaves - 1; ++g) t += red[g][v][lane];
        out[v] = t;
    }
    return true;
}

// out[c] (+)= sum_b part[b][c] (fixed order: deterministic)
__global__ __launch_bounds__(1024) void fold_partials_k(const float* __restrict__ part, float* __restrict__ out, int nblocks,
                                                       int C, int accumulate) {
    int c;
    float s[1];
    if (!fold_cols<1, float>(part, nblocks, C, c, s)) return;
    out[c] = (accumulate ? out[c] : 0.f) + s[0];
}

// ------------------------------------------------------------------------------------------------ max pooling
__global__ void maxpool_fwd_k(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx, int N,
                              int H, int W, int C, int ph, int pw, int pt, int pl, int Ho, int Wo) {
    const int64_t total = (int64_t)N * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int ow = (int)(p % Wo);
        p /= Wo;
        const int oh = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float best = -INFINITY;
        int bi = 0;
        for (int a = 0; a < ph; ++a) {
            const int ih = oh * ph + a - pt;
            if (ih < 0 || ih >= H) continue;
            for (int b = 0; b < pw; ++b) {
                const int iw = ow * pw + b - pl;
                if (iw < 0 || iw >= W) continue;
                const float v = x[(((int64_t)n * H + ih) * W + iw) * C + c];
                if (v > best) {
                    best = v;
                    bi = a * pw + b;
                }
            }
        }
        y[i] = best;
        if (idx != nullptr) idx[i] = (uint8_t)bi;
    }
}

// one thread per INPUT element: non-overlapping windows, so every input belongs to exactly one window
__global__ void maxpool_bwd_k(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx,
                              int N, int H, int W, int C, int ph, int pw, int pt, int pl, int Ho, int Wo) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int iw = (int)(p % W);
        p /= W;
        const int ih = (int)(p % H);
        const int n = (int)(p / H);
        const int oh = (ih + pt) / ph, ow = (iw + pl) / pw;
        float g = 0.f;
        if (oh < Ho && ow < Wo) {
            const int64_t o = (((int64_t)n * Ho + oh) * Wo + ow) * C + c;
            const int a = ih + pt - oh * ph, b = iw + pl - ow * pw;
            if ((int)idx[o] == a * pw + b) g = dy[o];
        }
        dx[i] = g;
    }
}

// ------------------------------------------------------------------------------------------------ nearest upsampling
__global__ void upsample_fwd_k(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C,
                               int uh, int uw) {
    const int Ho = H * uh, Wo = W * uw;
    const int64_t total = (int64_t)N * Ho * Wo * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int ow = (int)(p % Wo);
        p /= Wo;
        const int oh = (int)(p % Ho);
        const int n = (int)(p / Ho);
        y[i] = x[(((int64_t)n * H + oh / uh) * W + ow / uw) * C + c];
    }
}

__global__ void upsample_bwd_k(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C,
                               int uh, int uw) {
    const int Ho = H * uh, Wo = W * uw;
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t p = i / C;
        const int iw = (int)(p % W);
        p /= W;
        const int ih = (int)(p % H);
        const int n = (int)(p / H);
        float s = 0.f;
        for (int a = 0; a < uh; ++a)
            for (int b = 0; b < uw; ++b) s += dy[(((int64_t)n * Ho + ih * uh + a) * Wo + iw * uw + b) * C + c];
        dx[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------ batch norm
// pass 1: per-block partial (sum, sum of squares about a per-channel pivot) -> part[block][2][C].
// The pivot (first row of the tensor) keeps the one-pass variance well conditioned.
template <typename T>
struct BnStatsOp {
    static constexpr int VN = 16 / sizeof(T);
    const T* x;
    float piv[VN];
    __device__ __forceinline__ void begin(int c0) { ld16(x + c0, piv); }
    __device__ __forceinline__ void row(int64_t i, float (&acc)[2][VN]) {
        float v[VN];
        ld16(x + i, v);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float d = v[e] - piv[e];
            acc[0][e] += d;
            acc[1][e] = fmaf(d, d, acc[1][e]);
        }
    }
};

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_k(const T* __restrict__ x, float* __restrict__ part,
                                                   int64_t rows, int C, int64_t rows_per_block, int vec) {
    extern __shared__ float sm[];  // 2 * blockDim.x
    // blockIdx.y = statistics group: `rows` consecutive rows with statistics (and a pivot row) of their own
    x += (int64_t)blockIdx.y * rows * C;
    part += (int64_t)blockIdx.y * gridDim.x * 2 * C;
    if (vec) {
        const int64_t v0 = (int64_t)blockIdx.x * rows_per_block;
        BnStatsOp<T> op{x};
        row_sweep_vec<T, 2>(op, part, v0, v0 + rows_per_block < rows ? v0 + rows_per_block : rows, C, sm);
        return;
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    if (C <= (int)blockDim.x) {
        const int rif = blockDim.x / C;
        const int c = threadIdx.x % C, rr = threadIdx.x / C;
        float s = 0.f, q = 0.f;
        if (rr < rif) {
            const float piv = (float)x[c];
#pragma unroll 4
            for (int64_t r = r0 + rr; r < r1; r += rif) {
                const float d = (float)x[r * C + c] - piv;
                s += d;
                q = fmaf(d, d, q);
            }
        }
        sm[threadIdx.x] = (rr < rif) ? s : 0.f;
        sm[blockDim.x + threadIdx.x] = (rr < rif) ? q : 0.f;
        __syncthreads();
        if (threadIdx.x < C) {
            float ss = 0.f, qq = 0.f;
            for (int k = 0; k < rif; ++k) {
                ss += sm[k * C + threadIdx.x];
                qq += sm[blockDim.x + k * C + threadIdx.x];
            }
            part[((int64_t)blockIdx.x * 2 + 0) * C + threadIdx.x] = ss;
            part[((int64_t)blockIdx.x * 2 + 1) * C + threadIdx.x] = qq;
        }
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float piv = (float)x[c];
            float s = 0.f, q = 0.f;
#pragma unroll 4
            for (int64_t r = r0; r < r1; ++r) {
                const float d = (float)x[r * C + c] - piv;
                s += d;
                q = fmaf(d, d, q);
            }
            part[((int64_t)blockIdx.x * 2 + 0) * C + c] = s;
            part[((int64_t)blockIdx.x * 2 + 1) * C + c] = q;
        }
    }
}

// pass 2: fold partials -> mean, invstd; update the moving statistics.
template <typename T>
__global__ __launch_bounds__(1024) void bn_finalize_k(const T* __restrict__ x, const float* __restrict__ part, int nblocks,
                              float* __restrict__ save_mean, float* __restrict__ save_invstd,
                              float* __restrict__ moving_mean, float* __restrict__ moving_var, int64_t rows, int C,
                              float eps, float momentum, int unbiased, int groups) {
    // groups in order: the moving statistics receive one update per group, as separate calls on the groups would give
    for (int g = 0; g < groups; ++g) {
        int c;
        double sq[2];
        const bool mine = fold_cols<2, double>(part + (int64_t)g * nblocks * 2 * C, nblocks, C, c, sq);
        __syncthreads();  // fold_cols' LDS staging is reused by the next group
        if (!mine) continue;
        const double s = sq[0], q = sq[1];
        const double m = (double)rows;
        const double dmean = s / m;
        double var = q / m - dmean * dmean;
        if (var < 0.0) var = 0.0;
        const float mean = (float)((double)(float)x[(int64_t)g * rows * C + c] + dmean);
        save_mean[g * C + c] = mean;
        save_invstd[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
        if (moving_mean != nullptr) {
            const double mv = unbiased && rows > 1 ? var * m / (m - 1.0) : var;
            moving_mean[c] = momentum * moving_mean[c] + (1.f - momentum) * mean;
            moving_var[c] = momentum * moving_var[c] + (1.f - momentum) * (float)mv;
        }
    }
}

// The fold of pass 1 for statistics that are combined ACROSS replicas afterwards: per group and channel the count, the
// mean and the sum of squared deviations about that mean, in double (the combination adds the between-replica term
// count * (mean - global mean)^2, which needs the means to full precision).
template <typename T>
__global__ __launch_bounds__(1024) void bn_local_moments_k(const T* __restrict__ x, const float* __restrict__ part, int nblocks,
                                                            double* __restrict__ moments, int64_t rows, int C, int groups) {
    for (int g = 0; g < groups; ++g) {
        int c;
        double sq[2];
        const bool mine = fold_cols<2, double>(part + (int64_t)g * nblocks * 2 * C, nblocks, C, c, sq);
        __syncthreads();
        if (!mine) continue;
        const double m = (double)rows, dmean = sq[0] / m;
        double m2 = sq[1] - sq[0] * dmean;
        if (m2 < 0.0) m2 = 0.0;
        moments[((int64_t)g * 3 + 0) * C + c] = m;
        moments[((int64_t)g * 3 + 1) * C + c] = (double)(float)x[(int64_t)g * rows * C + c] + dmean;
        moments[((int64_t)g * 3 + 2) * C + c] = m2;
    }
}

// y = (x - mean) * invstd * gamma + beta   (mean/invstd either saved batch stats or derived from moving stats)
// A following activation layer may ride along (act != ACT_NONE): it sees the normalised value in its storage type, as it
// would coming from a separate launch.
template <typename T>
__global__ void bn_apply_k(const T* __restrict__ x, const float* __restrict__ gamma,
                           const float* __restrict__ beta, const float* __restrict__ mean,
                           const float* __restrict__ var_or_invstd, int is_var, float eps, T* __restrict__ y,
                           int64_t total, int C, int act, float alpha, int64_t group_elems, int vec) {
    if (vec) {  // 16-byte pieces (C is a multiple of the piece, so a piece never straddles two rows)
        constexpr int VN = 16 / sizeof(T);
        const int64_t nv = total / VN;
        const int64_t stride = (int64_t)gridDim.x * blockDim.x;
        int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if ((stride * VN) % C == 0) {
            // a thread's pieces all start at the same channel: the per-channel values stay in registers (reloaded when
            // the walk enters another statistics group) and the loop is one load, VN fused operations, one store
            const int c0 = (int)((i * VN) % C);
            float mu[VN], iv[VN], ga[VN], be[VN];
#pragma unroll
            for (int e = 0; e < VN; ++e) { ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e]; mu[e] = 0.f; iv[e] = 0.f; }
            int64_t g_end = 0;  // first element past the group whose statistics are loaded
            for (; i < nv; i += stride) {
                const int64_t b = i * VN;
                if (b >= g_end) {
                    const int64_t g = group_elems >= total ? 0 : b / group_elems;
                    g_end = group_elems >= total ? total : (g + 1) * group_elems;
#pragma unroll
                    for (int e = 0; e < VN; ++e) {
                        const float s = var_or_invstd[g * C + c0 + e];
                        iv[e] = is_var ? rsqrtf(s + eps) : s;
                        mu[e] = mean[g * C + c0 + e];
                    }
                }
                float v[VN];
                ld16(x + b, v);
                Pack16<T> o;
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    const T t = (T)((v[e] - mu[e]) * iv[e] * ga[e] + be[e]);
                    o.v[e] = act == SV_ACT_NONE ? t : (T)act_apply((float)t, act, alpha);
                }
                *reinterpret_cast<Pack16<T>*>(y + b) = o;
            }
            return;
        }
        for (; i < nv; i += stride) {
            const int64_t b = i * VN;
            const int c0 = (int)(b % C);
            const int s0 = group_elems >= total ? c0 : (int)(b / group_elems) * C + c0;
            float v[VN];
            ld16(x + b, v);
            Pack16<T> o;
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                const float inv = is_var ? rsqrtf(var_or_invstd[s0 + e] + eps) : var_or_invstd[s0 + e];
                const T t = (T)((v[e] - mean[s0 + e]) * inv * gamma[c0 + e] + beta[c0 + e]);
                o.v[e] = act == SV_ACT_NONE ? t : (T)act_apply((float)t, act, alpha);
            }
            *reinterpret_cast<Pack16<T>*>(y + b) = o;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int sc = group_elems >= total ? c : (int)(i / group_elems) * C + c;  // the row's statistics group
        const float inv = is_var ? rsqrtf(var_or_invstd[sc] + eps) : var_or_invstd[sc];
        const T v = (T)(((float)x[i] - mean[sc]) * inv * gamma[c] + beta[c]);
        y[i] = act == SV_ACT_NONE ? v : (T)act_apply((float)v, act, alpha);
    }
}

// The gradient arriving at the normalised value: dy itself, or - when the activation layer that followed rode along in
// the forward launch (yact = its output) - dy * act'(yact), rounded to the storage type as a separate launch would have.
template <typename T>
__device__ __forceinline__ float bn_incoming_grad(const T* __restrict__ dy, const T* __restrict__ yact, int64_t i,
                                                  int act_out, float alpha_out) {
    const float g = (float)dy[i];
    if (yact == nullptr) return g;
    return (float)(T)(g * act_grad_from_y((float)yact[i], act_out, alpha_out));
}

template <typename T>
struct BnBwdStatsOp {
    static constexpr int VN = 16 / sizeof(T);
    const T* x;
    const T* dy;
    const T* yact;
    const float* mean;
    const float* invstd;
    int act_out;
    float alpha_out;
    float mu[VN], inv[VN];
    __device__ __forceinline__ void begin(int c0) {
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            mu[e] = mean[c0 + e];
            inv[e] = invstd[c0 + e];
        }
    }
    __device__ __forceinline__ void row(int64_t i, float (&acc)[2][VN]) {
        float xv[VN], g[VN];
        ld16(x + i, xv);
        ld16(dy + i, g);
        if (yact != nullptr) {
            float yv[VN];
            ld16(yact + i, yv);
#pragma unroll
            for (int e = 0; e < VN; ++e) g[e] = (float)(T)(g[e] * act_grad_from_y(yv[e], act_out, alpha_out));
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            acc[0][e] += g[e];
            acc[1][e] = fmaf(g[e], (xv[e] - mu[e]) * inv[e], acc[1][e]);
        }
    }
};

// backward pass 1: per-block partial (sum dy, sum dy*xhat) -> part[block][2][C]
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_stats_k(const T* __restrict__ x, const T* __restrict__ dy,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, float* __restrict__ part,
                                                       int64_t rows, int C, int64_t rows_per_block,
                                                       const T* __restrict__ yact, int act_out, float alpha_out,
                                                       int vec) {
    extern __shared__ float sm[];
    {  // blockIdx.y = statistics group
        const int64_t go = (int64_t)blockIdx.y * rows * C;
        x += go;
        dy += go;
        if (yact != nullptr) yact += go;
        mean += blockIdx.y * C;
        invstd += blockIdx.y * C;
        part += (int64_t)blockIdx.y * gridDim.x * 2 * C;
    }
    if (vec) {
        const int64_t v0 = (int64_t)blockIdx.x * rows_per_block;
        BnBwdStatsOp<T> op{x, dy, yact, mean, invstd, act_out, alpha_out};
        row_sweep_vec<T, 2>(op, part, v0, v0 + rows_per_block < rows ? v0 + rows_per_block : rows, C, sm);
        return;
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    if (C <= (int)blockDim.x) {
        const int rif = blockDim.x / C;
        const int c = threadIdx.x % C, rr = threadIdx.x / C;
        float s = 0.f, q = 0.f;
        if (rr < rif) {
            const float mu = mean[c], inv = invstd[c];
#pragma unroll 4
            for (int64_t r = r0 + rr; r < r1; r += rif) {
                const float g = bn_incoming_grad(dy, yact, r * C + c, act_out, alpha_out);
                s += g;
                q = fmaf(g, ((float)x[r * C + c] - mu) * inv, q);
            }
        }
        sm[threadIdx.x] = (rr < rif) ? s : 0.f;
        sm[blockDim.x + threadIdx.x] = (rr < rif) ? q : 0.f;
        __syncthreads();
        if (threadIdx.x < C) {
            float ss = 0.f, qq = 0.f;
            for (int k = 0; k < rif; ++k) {
                ss += sm[k * C + threadIdx.x];
                qq += sm[blockDim.x + k * C + threadIdx.x];
            }
            part[((int64_t)blockIdx.x * 2 + 0) * C + threadIdx.x] = ss;
            part[((int64_t)blockIdx.x * 2 + 1) * C + threadIdx.x] = qq;
        }
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float mu = mean[c], inv = invstd[c];
            float s = 0.f, q = 0.f;
#pragma unroll 4
            for (int64_t r = r0; r < r1; ++r) {
                const float g = bn_incoming_grad(dy, yact, r * C + c, act_out, alpha_out);
                s += g;
                q = fmaf(g, ((float)x[r * C + c] - mu) * inv, q);
            }
            part[((int64_t)blockIdx.x * 2 + 0) * C + c] = s;
            part[((int64_t)blockIdx.x * 2 + 1) * C + c] = q;
        }
    }
}

// fold -> stats2[0][C] = sum dy (dbeta), stats2[1][C] = sum dy*xhat (dgamma); also emit dgamma/dbeta
__global__ __launch_bounds__(1024) void bn_bwd_fold_k(const float* __restrict__ part, int nblocks, float* __restrict__ stats2,
                              float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, int C, int groups) {
    // stats2[g] = the group's sums (its rows' input gradient uses them); dgamma / dbeta sum over the groups in order
    for (int g = 0; g < groups; ++g) {
        int c;
        float sq[2];
        const bool mine = fold_cols<2, float>(part + (int64_t)g * nblocks * 2 * C, nblocks, C, c, sq);
        __syncthreads();
        if (!mine) continue;
        const float s = sq[0], q = sq[1];
        stats2[(2 * g) * C + c] = s;
        stats2[(2 * g + 1) * C + c] = q;
        dbeta[c] = (accumulate || g > 0 ? dbeta[c] : 0.f) + s;
        dgamma[c] = (accumulate || g > 0 ? dgamma[c] : 0.f) + q;
    }
}

// dx = gamma*invstd/m * (m*dy - sum_dy - xhat*sum_dy_xhat)
template <typename T>
__global__ void bn_bwd_apply_k(const T* __restrict__ x, const T* __restrict__ dy,
                               const float* __restrict__ gamma, const float* __restrict__ mean,
                               const float* __restrict__ invstd, const float* __restrict__ stats2,
                               T* __restrict__ dx, int64_t rows, int C, int groups, int vec, float invm_all) {
    const int64_t total = rows * C * groups, group_elems = rows * C;  // rows: per statistics group
    // invm_all > 0: the sums in stats2 run over the rows of every replica (1 / their number), not over this launch's
    const float invm = invm_all > 0.f ? invm_all : 1.f / (float)rows;
    if (vec) {
        constexpr int VN = 16 / sizeof(T);
        const int64_t nv = total / VN;
        const int64_t stride = (int64_t)gridDim.x * blockDim.x;
        int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if ((stride * VN) % C == 0) {  // per-channel values in registers: see bn_apply_k
            const int c0 = (int)((i * VN) % C);
            float mu[VN], iv[VN], gi[VN], s1[VN], s2[VN];
            int64_t g_end = 0;
            for (; i < nv; i += stride) {
                const int64_t b = i * VN;
                if (b >= g_end) {
                    const int64_t g = groups == 1 ? 0 : b / group_elems;
                    g_end = (g + 1) * group_elems;
#pragma unroll
                    for (int e = 0; e < VN; ++e) {
                        iv[e] = invstd[g * C + c0 + e];
                        mu[e] = mean[g * C + c0 + e];
                        gi[e] = gamma[c0 + e] * iv[e];
                        s1[e] = stats2[2 * g * C + c0 + e];
                        s2[e] = stats2[(2 * g + 1) * C + c0 + e];
                    }
                }
                float xv[VN], gv[VN];
                ld16(x + b, xv);
                ld16(dy + b, gv);
                Pack16<T> o;
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    const float xh = (xv[e] - mu[e]) * iv[e];
                    o.v[e] = (T)(gi[e] * (gv[e] - invm * (s1[e] + xh * s2[e])));
                }
                *reinterpret_cast<Pack16<T>*>(dx + b) = o;
            }
            return;
        }
        for (; i < nv; i += stride) {
            const int64_t b = i * VN;
            const int c0 = (int)(b % C);
            const int g = groups == 1 ? 0 : (int)(b / group_elems);
            float xv[VN], gv[VN];
            ld16(x + b, xv);
            ld16(dy + b, gv);
            Pack16<T> o;
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                const float inv = invstd[g * C + c0 + e];
                const float xh = (xv[e] - mean[g * C + c0 + e]) * inv;
                o.v[e] = (T)(gamma[c0 + e] * inv *
                             (gv[e] - invm * (stats2[2 * g * C + c0 + e] + xh * stats2[(2 * g + 1) * C + c0 + e])));
            }
            *reinterpret_cast<Pack16<T>*>(dx + b) = o;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int g = groups == 1 ? 0 : (int)(i / group_elems);
        const float inv = invstd[g * C + c];
        const float xh = ((float)x[i] - mean[g * C + c]) * inv;
        dx[i] = (T)(gamma[c] * inv * ((float)dy[i] - invm * (stats2[2 * g * C + c] + xh * stats2[(2 * g + 1) * C + c])));
    }
}

template <typename T>
struct BnBwdApplyOp {
    static constexpr int VN = 16 / sizeof(T);
    const T* x;
    const T* dy;
    const T* yact;
    T* dx;
    const float* gamma;
    const float* mean;
    const float* invstd;
    const float* stats2;
    int act_out;
    float alpha_out;
    int act_in;
    float alpha_in, invm;
    int C;
    float mu[VN], inv[VN], ga[VN], s1[VN], s2[VN];
    __device__ __forceinline__ void begin(int c0) {
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            mu[e] = mean[c0 + e];
            inv[e] = invstd[c0 + e];
            ga[e] = gamma[c0 + e];
            s1[e] = stats2[c0 + e];
            s2[e] = stats2[C + c0 + e];
        }
    }
    __device__ __forceinline__ void row(int64_t i, float (&acc)[1][VN]) {
        float xv[VN], g[VN];
        ld16(x + i, xv);
        ld16(dy + i, g);
        if (yact != nullptr) {
            float yv[VN];
            ld16(yact + i, yv);
#pragma unroll
            for (int e = 0; e < VN; ++e) g[e] = (float)(T)(g[e] * act_grad_from_y(yv[e], act_out, alpha_out));
        }
        Pack16<T> o;
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float xh = (xv[e] - mu[e]) * inv[e];
            T d = (T)(ga[e] * inv[e] * (g[e] - invm * (s1[e] + xh * s2[e])));
            if (act_in != SV_ACT_NONE) d = (T)((float)d * act_grad_from_y(xv[e], act_in, alpha_in));
            o.v[e] = d;
            acc[0][e] += (float)d;
        }
        *reinterpret_cast<Pack16<T>*>(dx + i) = o;
    }
};

// The same with the neighbouring activations folded in, one workgroup per block of rows (the layout of the statistics
// pass).  yact: output of the activation layer that followed (its derivative scales the incoming gradient).  act_in: the
// activation of the PRODUCER of x (a conv whose output feeds this layer): dx is multiplied by act_in'(x), i.e. the launch
// emits the gradient at the conv's pre-activation, and partb[block][C] receives its column sums (the conv's bias
// gradient) when not null.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows_k(const T* __restrict__ x, const T* __restrict__ dy,
                                                            const T* __restrict__ yact, int act_out, float alpha_out,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ stats2, T* __restrict__ dx,
                                                            int act_in, float alpha_in, float* __restrict__ partb,
                                                            int64_t rows, int C, int64_t rows_per_block, int vec,
                                                            float invm_all) {
    extern __shared__ float sm[];
    {  // blockIdx.y = statistics group
        const int64_t go = (int64_t)blockIdx.y * rows * C;
        x += go;
        dy += go;
        dx += go;
        if (yact != nullptr) yact += go;
        mean += blockIdx.y * C;
        invstd += blockIdx.y * C;
        stats2 += 2 * blockIdx.y * C;
        if (partb != nullptr) partb += (int64_t)blockIdx.y * gridDim.x * C;
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    const float invm = invm_all > 0.f ? invm_all : 1.f / (float)rows;
    if (vec) {
        BnBwdApplyOp<T> op{x, dy, yact, dx, gamma, mean, invstd, stats2, act_out, alpha_out, act_in, alpha_in, invm, C};
        row_sweep_vec<T, 1>(op, partb, r0, r1, C, sm);
        return;
    }
    auto one = [&](int64_t i, float mu, float inv, float ga, float s1, float s2) -> float {
        const float xv = (float)x[i];
        const float g = bn_incoming_grad(dy, yact, i, act_out, alpha_out);
        const float xh = (xv - mu) * inv;
        T d = (T)(ga * inv * (g - invm * (s1 + xh * s2)));
        if (act_in != SV_ACT_NONE) d = (T)((float)d * act_grad_from_y(xv, act_in, alpha_in));
        dx[i] = d;
        return (float)d;
    };
    if (C <= (int)blockDim.x) {
        const int rif = blockDim.x / C;
        const int c = threadIdx.x % C, rr = threadIdx.x / C;
        float acc = 0.f;
        if (rr < rif) {
            const float mu = mean[c], inv = invstd[c], ga = gamma[c], s1 = stats2[c], s2 = stats2[C + c];
#pragma unroll 4
            for (int64_t r = r0 + rr; r < r1; r += rif) acc += one(r * C + c, mu, inv, ga, s1, s2);
        }
        if (partb != nullptr) {
            sm[threadIdx.x] = (rr < rif) ? acc : 0.f;
            __syncthreads();
            if (threadIdx.x < C) {
                float t = 0.f;
                for (int k = 0; k < rif; ++k) t += sm[k * C + threadIdx.x];
                partb[(int64_t)blockIdx.x * C + threadIdx.x] = t;
            }
        }
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float mu = mean[c], inv = invstd[c], ga = gamma[c], s1 = stats2[c], s2 = stats2[C + c];
            float acc = 0.f;
#pragma unroll 4
            for (int64_t r = r0; r < r1; ++r) acc += one(r * C + c, mu, inv, ga, s1, s2);
            if (partb != nullptr) partb[(int64_t)blockIdx.x * C + c] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------ concat / misc
template <typename T>
__global__ void concat_fwd_k(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y,
                             int64_t rows, int Ca, int Cb) {
    const int Cy = Ca + Cb;
    const int64_t total = rows * Cy;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cy);
        const int64_t r = i / Cy;
        y[i] = c < Ca ? a[r * Ca + c] : b[r * Cb + (c - Ca)];
    }
}

template <typename T>
__global__ void concat_bwd_k(const T* __restrict__ dy, T* __restrict__ da, T* __restrict__ db,
                             int64_t rows, int Ca, int Cb) {
    const int Cy = Ca + Cb;
    const int64_t total = rows * Cy;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cy);
        const int64_t r = i / Cy;
        if (c < Ca)
            da[r * Ca + c] = dy[i];
        else
            db[r * Cb + (c - Ca)] = dy[i];
    }
}

__global__ void dropout_k(const float* __restrict__ x, float* __restrict__ y, int64_t n, float rate, uint64_t seed,
                          uint64_t offset) {
    const float scale = 1.f / (1.f - rate);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = u01(seed, offset + (uint64_t)i) >= rate ? x[i] * scale : 0.f;
}

__global__ void reduce_sum_h_fwd_k(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
    const int64_t total = (int64_t)N * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t wc = i % ((int64_t)W * C);
        const int64_t n = i / ((int64_t)W * C);
        float s = 0.f;
        for (int h = 0; h < H; ++h) s += x[(n * H + h) * (int64_t)W * C + wc];
        y[i] = s;
    }
}

__global__ void reduce_sum_h_bwd_k(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C) {
    const int64_t total = (int64_t)N * H * W * C;
    const int64_t wcn = (int64_t)W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t wc = i % wcn;
        const int64_t n = i / (wcn * H);
        dx[i] = dy[n * wcn + wc];
    }
}

template <typename SRC>
__global__ void gather_rows_k(const SRC* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ dst,
                              int64_t n_rows, int64_t row_elems) {
    const int64_t total = n_rows * row_elems;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / row_elems, e = i % row_elems;
        const int64_t sr = idx != nullptr ? idx[r] : r;
        dst[i] = (float)src[sr * row_elems + e];
    }
}

// f32 rows whose length is a multiple of 4 and 16-byte aligned: 16-byte pieces, one 32-bit division per piece
__global__ void gather_rows_f4_k(const float4* __restrict__ src, const int64_t* __restrict__ idx, float4* __restrict__ dst,
                                 int n_rows, int row_f4) {
    const int total = n_rows * row_f4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / row_f4, e = i - r * row_f4;
        const int64_t sr = idx != nullptr ? idx[r] : r;
        dst[i] = src[sr * row_f4 + e];
    }
}

__global__ void mask_input_k(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ y,
                             int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = x[i] * (1.f - m[i]);
}

__global__ void fill_k(float* __restrict__ p, float v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void axpy_k(float* __restrict__ y, const float* __restrict__ x, float a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = fmaf(a, x[i], y[i]);
}

template <typename T>
__global__ void axpy_t_k(T* __restrict__ y, const T* __restrict__ x, float a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (T)fmaf(a, (float)x[i], (float)y[i]);
}

// y = x * (1 - mask), x and mask float32, y in the model's storage type
// One 64-lane workgroup per image, the procedure of run_context_encoder.py:39-58: the block [seg, 3 seg)^2 with
// seg = W / 8, then seg x seg blocks at uniform positions in [0, W - seg)^2 (rows past H clipped, as NumPy slicing does)
// until at least pct of W^2 pixels are set.  Draw d of image n uses counters offset + (n << 20) + 2d (+1): a pure
// function of its arguments.  The cap on draws is the loop's exit for coverages that take long to reach.
constexpr int kMaskMaxDraws = 1 << 16;
__global__ __launch_bounds__(64) void random_block_mask_k(float* __restrict__ mask, int H, int W, int C, double pct,
                                                          uint64_t seed, uint64_t offset) {
    extern __shared__ unsigned char bm[];
    const int n = blockIdx.x, seg = W / 8, lane = threadIdx.x;
    int cnt = 0;
    for (int i = lane; i < H * W; i += 64) {
        const int r = i / W, c = i - r * W;
        const unsigned char v = r >= seg && r < 3 * seg && c >= seg && c < 3 * seg;
        bm[i] = v;
        cnt += v;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    __syncthreads();
    const double total = (double)W * (double)W;
    const uint64_t base = offset + ((uint64_t)n << 20);
    const uint64_t span = (uint64_t)(W - seg);
    for (int d = 0; d < kMaskMaxDraws && (double)cnt / total < pct; ++d) {
        const int rx = (int)((mix64(seed ^ mix64(base + 2 * (uint64_t)d)) >> 11) % span);
        const int ry = (int)((mix64(seed ^ mix64(base + 2 * (uint64_t)d + 1)) >> 11) % span);
        int add = 0;
        for (int i = lane; i < seg * seg; i += 64) {
            const int r = rx + i / seg, c = ry + i % seg;
            if (r < H && !bm[r * W + c]) {
                bm[r * W + c] = 1;
                ++add;
            }
        }
        for (int o = 32; o > 0; o >>= 1) add += __shfl_xor(add, o);
        cnt += add;
        __syncthreads();
    }
    float* out = mask + (size_t)n * H * W * C;
    for (int i = lane; i < H * W * C; i += 64) out[i] = bm[i / C] ? 1.f : 0.f;
}

template <typename T>
__global__ void mask_input_t_k(const float* __restrict__ x, const float* __restrict__ m, T* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (T)(x[i] * (1.f - m[i]));
}

__global__ void cast_f32_bf16_k(const float* __restrict__ x, __hip_bfloat16* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = __float2bfloat16(x[i]);
}

__global__ void cast_bf16_f32_k(const __hip_bfloat16* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = __bfloat162float(x[i]);
}

}  // namespace sv

// =================================================================================================== C ABI
using namespace sv;

// Row-chunked column reductions share this plan: nblocks blocks of rows_per_block rows, partial slabs in
// a caller-visible scratch.  The scratch for act_bwd/bias_grad/batchnorm lives in a per-device buffer that
// is allocated once (first use) and only grows when a larger reduction shows up.
namespace {
struct ColPlan {
    int nblocks;
    int64_t rows_per_block;
};
ColPlan col_plan(int64_t rows, int C) {
    // A thread's row loop is a chain of dependent-latency loads, so keep it near 8 iterations (a block sweeps
    // 256 / C rows at once when C < 256); at most 512 blocks: the fold that follows is one workgroup per 64 channels
    // whose 4 waves walk the blocks' partial sums, i.e. pure latency that grows with the block count.
    const int64_t sweep = C < 256 ? 256 / C : 1;
    int64_t rpb = 8 * sweep;
    const int64_t floor_rpb = ceil_div(rows, 512);
    if (rpb < floor_rpb) rpb = floor_rpb;
    ColPlan p;
    p.rows_per_block = rpb;
    p.nblocks = (int)ceil_div(rows, rpb);
    return p;
}

constexpr int kMaxC = 4096;

// Whether the vectorised row sweep applies: C a multiple of the 16-byte piece, at most 256 pieces per row, every tensor
// 16-byte aligned.  *lds = dynamic LDS bytes of the launch (the larger of the scalar and the vector form's needs).
static int row_vec(int dtype, int C, int nq, size_t scalar_lds, size_t* lds, std::initializer_list<const void*> ptrs) {
    const int vn = dtype == SV_F32 ? 4 : 8;
    *lds = scalar_lds;
    if (C % vn != 0 || C / vn > 256) return 0;
    for (const void* p : ptrs)
        if (p != nullptr && (reinterpret_cast<uintptr_t>(p) & 15) != 0) return 0;
    const size_t need = (size_t)nq * 256 * vn * sizeof(float);
    if (need > *lds) *lds = need;
    return 1;
}
// Workspace layout of a column reduction: [nblocks][2][C] partial slabs followed by [2][C] folded sums.
size_t col_ws_bytes(int64_t rows, int C) {
    ColPlan p = col_plan(rows, C);
    return ((size_t)p.nblocks * 2 * C + 2 * (size_t)C) * sizeof(float);
}
int col_ws(void* ws, size_t ws_bytes, int64_t rows, int C, float** part, float** folded, const char* who) {
    const size_t need = col_ws_bytes(rows, C);
    if (ws == nullptr || ws_bytes < need)
        return fail(SV_E_WORKSPACE, "%s: workspace %zu bytes < required %zu", who, ws_bytes, need);
    ColPlan p = col_plan(rows, C);
    *part = (float*)ws;
    *folded = (float*)ws + (size_t)p.nblocks * 2 * C;
    return SV_OK;
}
}  // namespace

// Runs BODY with T = the storage type `dtype` names (float or bf16).
#define SV_WITH_T(dtype, BODY)        \
    do {                              \
        if ((dtype) == SV_F32) {      \
            typedef float T;          \
            BODY                      \
        } else {                      \
            typedef __bf16 T;         \
            BODY                      \
        }                             \
    } while (0)

extern "C" {

int sv_act_fwd(const void* x, void* y, int64_t n, int act, float alpha, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_act_fwd");
    SV_REQUIRE(x && y && n >= 0, "sv_act_fwd: bad arguments");
    if (n == 0) return SV_OK;
    SV_WITH_T(dtype, hipLaunchKernelGGL(act_fwd_k<T>, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, to_stream(stream),
                                        (const T*)x, (T*)y, n, act, alpha););
    return check_launch("sv_act_fwd");
}

size_t sv_colreduce_workspace_bytes(int64_t rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return col_ws_bytes(rows, C);
}

int sv_act_bwd(const void* dy, const void* y, void* dz, void* db, int db_accumulate, int64_t rows, int C, int act,
               float alpha, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_act_bwd");
    SV_REQUIRE(dy && y && dz && rows > 0 && C > 0 && C <= kMaxC, "sv_act_bwd: bad arguments (C=%d)", C);
    ColPlan p = col_plan(rows, C);
    float *part = nullptr, *folded = nullptr;
    if (db != nullptr) {
        int rc = col_ws(workspace, workspace_bytes, rows, C, &part, &folded, "sv_act_bwd");
        if (rc) return rc;
    }
    size_t lds;
    const int vec = row_vec(dtype, C, 1, 256 * sizeof(float), &lds, {dy, y, dz});
    SV_WITH_T(dtype, hipLaunchKernelGGL(act_bwd_k<T>, dim3(p.nblocks), dim3(256), lds, to_stream(stream), (const T*)dy,
                                        (const T*)y, (T*)dz, part, rows, C, act, alpha, p.rows_per_block, vec););
    int rc = check_launch("sv_act_bwd");
    if (rc || db == nullptr) return rc;
    if (defer_reduce(part, (float*)db, C, p.nblocks, db_accumulate)) return SV_OK;
    hipLaunchKernelGGL(fold_partials_k, dim3((unsigned)ceil_div(C, 64)), dim3(64 * kFoldWaves), 0, to_stream(stream), part, (float*)db,
                       p.nblocks, C, db_accumulate);
    return check_launch("sv_act_bwd/fold");
}

int sv_bias_grad(const void* dz, void* db, int accumulate, int64_t rows, int C, void* workspace,
                 size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_bias_grad");
    SV_REQUIRE(dz && db && rows > 0 && C > 0 && C <= kMaxC, "sv_bias_grad: bad arguments (C=%d)", C);
    ColPlan p = col_plan(rows, C);
    float *part = nullptr, *folded = nullptr;
    int rc = col_ws(workspace, workspace_bytes, rows, C, &part, &folded, "sv_bias_grad");
    if (rc) return rc;
    size_t lds;
    const int vec = row_vec(dtype, C, 1, 256 * sizeof(float), &lds, {dz});
    SV_WITH_T(dtype, hipLaunchKernelGGL(colsum_k<T>, dim3(p.nblocks), dim3(256), lds, to_stream(stream), (const T*)dz, part,
                                        rows, C, p.rows_per_block, vec););
    rc = check_launch("sv_bias_grad");
    if (rc) return rc;
    if (defer_reduce(part, (float*)db, C, p.nblocks, accumulate)) return SV_OK;
    hipLaunchKernelGGL(fold_partials_k, dim3((unsigned)ceil_div(C, 64)), dim3(64 * kFoldWaves), 0, to_stream(stream), part, (float*)db,
                       p.nblocks, C, accumulate);
    return check_launch("sv_bias_grad/fold");
}

int sv_maxpool2d_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int ph, int pw, int pad_t,
                     int pad_l, int Ho, int Wo, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_maxpool2d_fwd");
    SV_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && ph > 0 && pw > 0 && ph * pw <= 255 && Ho > 0 && Wo > 0,
               "sv_maxpool2d_fwd: bad arguments");
    SV_REQUIRE(pad_t >= 0 && pad_t < ph && pad_l >= 0 && pad_l < pw, "sv_maxpool2d_fwd: pad must be < pool");
    SV_REQUIRE((int64_t)(Ho - 1) * ph - pad_t < H && (int64_t)(Wo - 1) * pw - pad_l < W,
               "sv_maxpool2d_fwd: output extent does not fit the input");
    const int64_t total = (int64_t)N * Ho * Wo * C;
    hipLaunchKernelGGL(maxpool_fwd_k, dim3(ew_grid(total, 256, 8192)), dim3(256), 0, to_stream(stream),
                       (const float*)x, (float*)y, idx, N, H, W, C, ph, pw, pad_t, pad_l, Ho, Wo);
    return check_launch("sv_maxpool2d_fwd");
}

int sv_maxpool2d_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, int ph, int pw,
                     int pad_t, int pad_l, int Ho, int Wo, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_maxpool2d_bwd");
    SV_REQUIRE(dy && idx && dx && N > 0 && H > 0 && W > 0 && C > 0 && ph > 0 && pw > 0 && Ho > 0 && Wo > 0,
               "sv_maxpool2d_bwd: bad arguments");
    SV_REQUIRE(pad_t >= 0 && pad_t < ph && pad_l >= 0 && pad_l < pw, "sv_maxpool2d_bwd: pad must be < pool");
    const int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(maxpool_bwd_k, dim3(ew_grid(total, 256, 8192)), dim3(256), 0, to_stream(stream),
                       (const float*)dy, idx, (float*)dx, N, H, W, C, ph, pw, pad_t, pad_l, Ho, Wo);
    return check_launch("sv_maxpool2d_bwd");
}

int sv_upsample2d_fwd(const void* x, void* y, int N, int H, int W, int C, int uh, int uw, int dtype,
                      sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_upsample2d_fwd");
    SV_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && uh > 0 && uw > 0, "sv_upsample2d_fwd: bad arguments");
    const int64_t total = (int64_t)N * H * uh * W * uw * C;
    hipLaunchKernelGGL(upsample_fwd_k, dim3(ew_grid(total, 256, 8192)), dim3(256), 0, to_stream(stream),
                       (const float*)x, (float*)y, N, H, W, C, uh, uw);
    return check_launch("sv_upsample2d_fwd");
}

int sv_upsample2d_bwd(const void* dy, void* dx, int N, int H, int W, int C, int uh, int uw, int dtype,
                      sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_upsample2d_bwd");
    SV_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && uh > 0 && uw > 0, "sv_upsample2d_bwd: bad arguments");
    const int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(upsample_bwd_k, dim3(ew_grid(total, 256, 8192)), dim3(256), 0, to_stream(stream),
                       (const float*)dy, (float*)dx, N, H, W, C, uh, uw);
    return check_launch("sv_upsample2d_bwd");
}

// Row blocks of the backward launch that also carries neighbouring activations: it is a streaming pass, so it takes more
// (shorter) blocks than the statistics passes; each leaves one row of bias partial sums.
static ColPlan bn_apply_plan(int64_t rows_g, int C) {
    const int64_t sweep = C < 256 ? 256 / C : 1;
    int64_t rpb = 8 * sweep;
    const int64_t floor_rpb = ceil_div(rows_g, 2048);
    if (rpb < floor_rpb) rpb = floor_rpb;
    ColPlan p;
    p.rows_per_block = rpb;
    p.nblocks = (int)ceil_div(rows_g, rpb);
    return p;
}

static size_t bn_ws_bytes(int64_t rows_g, int C, int groups) {
    ColPlan p = col_plan(rows_g, C), pa = bn_apply_plan(rows_g, C);
    return ((size_t)groups * ((size_t)p.nblocks * 2 * C + 2 * (size_t)C + (size_t)pa.nblocks * C)) * sizeof(float);
}

size_t sv_batchnorm_workspace_bytes(int64_t rows, int C, int groups) {
    if (rows <= 0 || C <= 0 || groups <= 0 || rows % groups) return 0;
    return bn_ws_bytes(rows / groups, C, groups);
}

// workspace: [groups][nblocks][2][C] partial sums | [groups][2][C] folded sums | [groups * nblocks][C] bias partial sums
static int bn_ws(void* ws, size_t ws_bytes, int64_t rows_g, int C, int groups, float** part, float** folded, float** partb,
                 const char* who) {
    const size_t need = bn_ws_bytes(rows_g, C, groups);
    if (ws == nullptr || ws_bytes < need)
        return fail(SV_E_WORKSPACE, "%s: workspace %zu bytes < required %zu", who, ws_bytes, need);
    ColPlan p = col_plan(rows_g, C);
    *part = (float*)ws;
    *folded = *part + (size_t)groups * p.nblocks * 2 * C;
    *partb = *folded + (size_t)groups * 2 * C;
    return SV_OK;
}

int sv_batchnorm_act_fwd_train(const void* x, const float* gamma, const float* beta, void* y, float* save_mean,
                               float* save_invstd, float* moving_mean, float* moving_var, int64_t rows, int C, float eps,
                               float momentum, int unbiased_moving_var, int groups, int act, float alpha, void* workspace,
                               size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_batchnorm_fwd_train");
    SV_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && rows > 0 && C > 0 && C <= kMaxC,
               "sv_batchnorm_fwd_train: bad arguments (C=%d)", C);
    SV_REQUIRE(groups >= 1 && groups <= 64 && rows % groups == 0,
               "sv_batchnorm_fwd_train: %lld rows do not split into %d statistics groups", (long long)rows, groups);
    SV_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr),
               "sv_batchnorm_fwd_train: moving_mean and moving_var must both be given or both be NULL");
    const int64_t rows_g = rows / groups;
    ColPlan p = col_plan(rows_g, C);
    float *part = nullptr, *folded = nullptr, *partb = nullptr;
    int rc = bn_ws(workspace, workspace_bytes, rows_g, C, groups, &part, &folded, &partb, "sv_batchnorm_fwd_train");
    if (rc) return rc;
    hipStream_t st = to_stream(stream);
    size_t lds;
    const int vec = row_vec(dtype, C, 2, 512 * sizeof(float), &lds, {x});
    SV_WITH_T(dtype, hipLaunchKernelGGL(bn_stats_k<T>, dim3(p.nblocks, groups), dim3(256), lds, st, (const T*)x, part, rows_g,
                                        C, p.rows_per_block, vec););
    rc = check_launch("bn_stats");
    if (rc) return rc;
    SV_WITH_T(dtype, hipLaunchKernelGGL(bn_finalize_k<T>, dim3((unsigned)ceil_div(C, 64)), dim3(64 * kFoldWaves), 0, st, (const T*)x, part,
                                        p.nblocks, save_mean, save_invstd, moving_mean, moving_var, rows_g, C, eps, momentum,
                                        unbiased_moving_var, groups););
    rc = check_launch("bn_finalize");
    if (rc) return rc;
    const int64_t total = rows * C;
    size_t unused;
    const int vec_a = row_vec(dtype, C, 0, 0, &unused, {x, y});
    SV_WITH_T(dtype, hipLaunchKernelGGL(bn_apply_k<T>, dim3(ew_grid(total / 4 + 1, 256, 8192)), dim3(256), 0, st, (const T*)x,
                                        gamma, beta, save_mean, save_invstd, 0, eps, (T*)y, total, C, act, alpha, rows_g * C,
                                        vec_a););
    return check_launch("bn_apply");
}

int sv_batchnorm_fwd_train(const void* x, const float* gamma, const float* beta, void* y, float* save_mean,
                           float* save_invstd, float* moving_mean, float* moving_var, int64_t rows, int C, float eps,
                           float momentum, int unbiased_moving_var, void* workspace, size_t workspace_bytes, int dtype,
                           sv_stream_t stream) {
    // the plain layer; its workspace contract (sv_colreduce_workspace_bytes) is the smaller historical one
    if (workspace != nullptr && rows > 0 && C > 0 && workspace_bytes >= col_ws_bytes(rows, C) &&
        workspace_bytes < bn_ws_bytes(rows, C, 1)) {
        // the fused entry point's layout only adds the bias partial sums, which this call never writes
        workspace_bytes = bn_ws_bytes(rows, C, 1);
    }
    return sv_batchnorm_act_fwd_train(x, gamma, beta, y, save_mean, save_invstd, moving_mean, moving_var, rows, C, eps,
                                      momentum, unbiased_moving_var, 1, SV_ACT_NONE, 0.f, workspace, workspace_bytes, dtype,
                                      stream);
}

int sv_batchnorm_act_fwd_infer(const void* x, const float* gamma, const float* beta, const float* moving_mean,
                               const float* moving_var, void* y, int64_t rows, int C, float eps, int act, float alpha,
                               int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_batchnorm_fwd_infer");
    SV_REQUIRE(x && gamma && beta && moving_mean && moving_var && y && rows > 0 && C > 0,
               "sv_batchnorm_fwd_infer: bad arguments");
    const int64_t total = rows * C;
    size_t unused;
    const int vec = row_vec(dtype, C, 0, 0, &unused, {x, y});
    SV_WITH_T(dtype, hipLaunchKernelGGL(bn_apply_k<T>, dim3(ew_grid(total / 4 + 1, 256, 8192)), dim3(256), 0, to_stream(stream),
                                        (const T*)x, gamma, beta, moving_mean, moving_var, 1, eps, (T*)y, total, C, act, alpha,
                                        total, vec););
    return check_launch("sv_batchnorm_fwd_infer");
}

int sv_batchnorm_fwd_infer(const void* x, const float* gamma, const float* beta, const float* moving_mean,
                           const float* moving_var, void* y, int64_t rows, int C, float eps, int dtype,
                           sv_stream_t stream) {
    return sv_batchnorm_act_fwd_infer(x, gamma, beta, moving_mean, moving_var, y, rows, C, eps, SV_ACT_NONE, 0.f, dtype,
                                      stream);
}

// phase 0: the whole backward; 1: the column sums only (stats2 -> sums_io, dgamma / dbeta); 2: the input gradient from
// the sums in sums_io (summed over every replica, `all_rows` rows per statistics group among them)
static int bn_bwd_impl(int phase, float* sums_io, int64_t all_rows, const void* x, const void* dy, const void* y_act,
                       int act_out, float alpha_out, const float* gamma,
                         const float* save_mean, const float* save_invstd, void* dx, float* dgamma, float* dbeta,
                         int accumulate, int act_in, float alpha_in, float* dbias_in, int dbias_accumulate, int64_t rows,
                         int C, int groups, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_batchnorm_bwd");
    SV_REQUIRE(x && dy && save_mean && save_invstd && rows > 0 && C > 0 && C <= kMaxC &&
                   (phase == 2 || (dgamma && dbeta)) && (phase == 1 || (dx && gamma)),
               "sv_batchnorm_bwd: bad arguments (C=%d)", C);
    SV_REQUIRE(groups >= 1 && groups <= 64 && rows % groups == 0,
               "sv_batchnorm_bwd: %lld rows do not split into %d statistics groups", (long long)rows, groups);
    SV_REQUIRE((y_act == nullptr) == (act_out == SV_ACT_NONE), "sv_batchnorm_bwd: y_act and act_out go together");
    const int64_t rows_g = rows / groups;
    ColPlan p = col_plan(rows_g, C);
    float *part = nullptr, *stats2 = nullptr, *partb = nullptr;
    int rc = bn_ws(workspace, workspace_bytes, rows_g, C, groups, &part, &stats2, &partb, "sv_batchnorm_bwd");
    if (rc) return rc;
    hipStream_t st = to_stream(stream);
    size_t lds;
    const int vec = row_vec(dtype, C, 2, 512 * sizeof(float), &lds, {x, dy, y_act, dx ? dx : x});
    if (phase != 2) {
        if (phase == 1) stats2 = sums_io;
        SV_WITH_T(dtype, hipLaunchKernelGGL(bn_bwd_stats_k<T>, dim3(p.nblocks, groups), dim3(256), lds, st, (const T*)x,
                                            (const T*)dy, save_mean, save_invstd, part, rows_g, C, p.rows_per_block,
                                            (const T*)y_act, act_out, alpha_out, vec););
        rc = check_launch("bn_bwd_stats");
        if (rc) return rc;
        hipLaunchKernelGGL(bn_bwd_fold_k, dim3((unsigned)ceil_div(C, 64)), dim3(64 * kFoldWaves), 0, st, part, p.nblocks, stats2, dgamma, dbeta,
                           accumulate, C, groups);
        rc = check_launch("bn_bwd_fold");
        if (rc || phase == 1) return rc;
    } else {
        stats2 = sums_io;
    }
    const float invm_all = phase == 2 ? 1.f / (float)all_rows : 0.f;
    if (y_act == nullptr && act_in == SV_ACT_NONE && dbias_in == nullptr) {
        SV_WITH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply_k<T>, dim3(ew_grid(rows * C / 4 + 1, 256, 8192)), dim3(256), 0, st,
                                            (const T*)x, (const T*)dy, gamma, save_mean, save_invstd, stats2, (T*)dx, rows_g, C,
                                            groups, vec, invm_all););
        return check_launch("bn_bwd_apply");
    }
    const ColPlan pa = bn_apply_plan(rows_g, C);
    size_t lds_a;
    const int vec_a = row_vec(dtype, C, 1, 256 * sizeof(float), &lds_a, {x, dy, y_act, dx});
    SV_WITH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply_rows_k<T>, dim3(pa.nblocks, groups), dim3(256), lds_a, st, (const T*)x,
                                        (const T*)dy, (const T*)y_act, act_out, alpha_out, gamma, save_mean, save_invstd,
                                        stats2, (T*)dx, act_in, alpha_in, dbias_in ? partb : nullptr, rows_g, C,
                                        pa.rows_per_block, vec_a, invm_all););
    rc = check_launch("bn_bwd_apply_rows");
    if (rc || dbias_in == nullptr) return rc;
    const int slabs = groups * pa.nblocks;
    if (defer_reduce(partb, dbias_in, C, slabs, dbias_accumulate)) return SV_OK;
    hipLaunchKernelGGL(fold_partials_k, dim3((unsigned)ceil_div(C, 64)), dim3(64 * kFoldWaves), 0, st, partb, dbias_in, slabs, C,
                       dbias_accumulate);
    return check_launch("bn_bwd/bias fold");
}

int sv_batchnorm_act_bwd(const void* x, const void* dy, const void* y_act, int act_out, float alpha_out, const float* gamma,
                         const float* save_mean, const float* save_invstd, void* dx, float* dgamma, float* dbeta,
                         int accumulate, int act_in, float alpha_in, float* dbias_in, int dbias_accumulate, int64_t rows,
                         int C, int groups, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream) {
    return bn_bwd_impl(0, nullptr, 0, x, dy, y_act, act_out, alpha_out, gamma, save_mean, save_invstd, dx, dgamma, dbeta,
                       accumulate, act_in, alpha_in, dbias_in, dbias_accumulate, rows, C, groups, workspace, workspace_bytes,
                       dtype, stream);
}

// ---- BatchNorm with statistics over every data-parallel replica: the layer in three steps with the replicas' exchange
// (an all-gather of the moments, an all-reduce of the two backward sums: the caller's) between them
int sv_batchnorm_sync_moments(const void* x, double* moments, int64_t rows, int C, int groups, void* workspace,
                              size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_batchnorm_sync_moments");
    SV_REQUIRE(x && moments && rows > 0 && C > 0 && C <= kMaxC, "sv_batchnorm_sync_moments: bad arguments (C=%d)", C);
    SV_REQUIRE(groups >= 1 && groups <= 64 && rows % groups == 0,
               "sv_batchnorm_sync_moments: %lld rows do not split into %d statistics groups", (long long)rows, groups);
    const int64_t rows_g = rows / groups;
    ColPlan p = col_plan(rows_g, C);
    float *part = nullptr, *folded = nullptr, *partb = nullptr;
    int rc = bn_ws(workspace, workspace_bytes, rows_g, C, groups, &part, &folded, &partb, "sv_batchnorm_sync_moments");
    if (rc) return rc;
    hipStream_t st = to_stream(stream);
    size_t lds;
    const int vec = row_vec(dtype, C, 2, 512 * sizeof(float), &lds, {x});
    SV_WITH_T(dtype, hipLaunchKernelGGL(bn_stats_k<T>, dim3(p.nblocks, groups), dim3(256), lds, st, (const T*)x, part, rows_g,
                                        C, p.rows_per_block, vec););
    rc = check_launch("bn_stats");
    if (rc) return rc;
    SV_WITH_T(dtype, hipLaunchKernelGGL(bn_local_moments_k<T>, dim3((unsigned)ceil_div(C, 64)), dim3(64 * kFoldWaves), 0, st,
                                        (const T*)x, part, p.nblocks, moments, rows_g, C, groups););
    return check_launch("bn_local_moments");
}

int sv_batchnorm_sync_fwd_apply(const void* x, const float* gamma, const float* beta, const float* mean,
                                const float* invstd, void* y, int64_t rows, int C, int groups, int act, float alpha,
                                int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_batchnorm_sync_fwd_apply");
    SV_REQUIRE(x && gamma && beta && mean && invstd && y && rows > 0 && C > 0 && groups >= 1 && rows % groups == 0,
               "sv_batchnorm_sync_fwd_apply: bad arguments");
    const int64_t total = rows * C;
    size_t unused;
    const int vec = row_vec(dtype, C, 0, 0, &unused, {x, y});
    SV_WITH_T(dtype, hipLaunchKernelGGL(bn_apply_k<T>, dim3(ew_grid(total / 4 + 1, 256, 8192)), dim3(256), 0, to_stream(stream),
                                        (const T*)x, gamma, beta, mean, invstd, 0, 0.f, (T*)y, total, C, act, alpha,
                                        rows / groups * C, vec););
    return check_launch("sv_batchnorm_sync_fwd_apply");
}

int sv_batchnorm_sync_bwd_sums(const void* x, const void* dy, const void* y_act, int act_out, float alpha_out,
                               const float* mean, const float* invstd, float* sums, float* dgamma, float* dbeta,
                               int accumulate, int64_t rows, int C, int groups, void* workspace, size_t workspace_bytes,
                               int dtype, sv_stream_t stream) {
    SV_REQUIRE(sums != nullptr, "sv_batchnorm_sync_bwd_sums: null sums");
    return bn_bwd_impl(1, sums, 0, x, dy, y_act, act_out, alpha_out, nullptr, mean, invstd, nullptr, dgamma, dbeta, accumulate,
                       SV_ACT_NONE, 0.f, nullptr, 0, rows, C, groups, workspace, workspace_bytes, dtype, stream);
}

int sv_batchnorm_sync_bwd_apply(const void* x, const void* dy, const void* y_act, int act_out, float alpha_out,
                                const float* gamma, const float* mean, const float* invstd, const float* sums,
                                int64_t all_rows_per_group, void* dx, int act_in, float alpha_in, float* dbias_in,
                                int dbias_accumulate, int64_t rows, int C, int groups, void* workspace,
                                size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_REQUIRE(sums != nullptr && all_rows_per_group >= rows / (groups > 0 ? groups : 1),
               "sv_batchnorm_sync_bwd_apply: the sums must cover at least this replica's rows");
    return bn_bwd_impl(2, const_cast<float*>(sums), all_rows_per_group, x, dy, y_act, act_out, alpha_out, gamma, mean, invstd,
                       dx, nullptr, nullptr, 0, act_in, alpha_in, dbias_in, dbias_accumulate, rows, C, groups, workspace,
                       workspace_bytes, dtype, stream);
}

int sv_batchnorm_bwd(const void* x, const void* dy, const float* gamma, const float* save_mean,
                     const float* save_invstd, void* dx, float* dgamma, float* dbeta, int accumulate, int64_t rows,
                     int C, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream) {
    if (workspace != nullptr && rows > 0 && C > 0 && workspace_bytes >= col_ws_bytes(rows, C) &&
        workspace_bytes < bn_ws_bytes(rows, C, 1)) {
        workspace_bytes = bn_ws_bytes(rows, C, 1);  // see sv_batchnorm_fwd_train: the tail of the layout stays unwritten
    }
    return sv_batchnorm_act_bwd(x, dy, nullptr, SV_ACT_NONE, 0.f, gamma, save_mean, save_invstd, dx, dgamma, dbeta, accumulate,
                                SV_ACT_NONE, 0.f, nullptr, 0, rows, C, 1, workspace, workspace_bytes, dtype, stream);
}

int sv_concat_channels_fwd(const void* a, const void* b, void* y, int64_t rows, int Ca, int Cb, int dtype,
                           sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_concat_channels_fwd");
    SV_REQUIRE(a && b && y && rows > 0 && Ca > 0 && Cb > 0, "sv_concat_channels_fwd: bad arguments");
    SV_WITH_T(dtype, hipLaunchKernelGGL(concat_fwd_k<T>, dim3(ew_grid(rows * (Ca + Cb), 256, 8192)), dim3(256), 0,
                                        to_stream(stream), (const T*)a, (const T*)b, (T*)y, rows, Ca, Cb););
    return check_launch("sv_concat_channels_fwd");
}

int sv_concat_channels_bwd(const void* dy, void* da, void* db, int64_t rows, int Ca, int Cb, int dtype,
                           sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_concat_channels_bwd");
    SV_REQUIRE(dy && da && db && rows > 0 && Ca > 0 && Cb > 0, "sv_concat_channels_bwd: bad arguments");
    SV_WITH_T(dtype, hipLaunchKernelGGL(concat_bwd_k<T>, dim3(ew_grid(rows * (Ca + Cb), 256, 8192)), dim3(256), 0,
                                        to_stream(stream), (const T*)dy, (T*)da, (T*)db, rows, Ca, Cb););
    return check_launch("sv_concat_channels_bwd");
}

int sv_dropout_fwd(const void* x, void* y, int64_t n, float rate, uint64_t seed, uint64_t offset, int dtype,
                   sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_dropout_fwd");
    SV_REQUIRE(x && y && n > 0 && rate >= 0.f && rate < 1.f, "sv_dropout_fwd: bad arguments");
    hipLaunchKernelGGL(dropout_k, dim3(ew_grid(n, 256, 8192)), dim3(256), 0, to_stream(stream), (const float*)x,
                       (float*)y, n, rate, seed, offset);
    return check_launch("sv_dropout_fwd");
}

int sv_dropout_bwd(const void* dy, void* dx, int64_t n, float rate, uint64_t seed, uint64_t offset, int dtype,
                   sv_stream_t stream) {
    return sv_dropout_fwd(dy, dx, n, rate, seed, offset, dtype, stream);  // same mask, same scale
}

int sv_reduce_sum_h_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_reduce_sum_h_fwd");
    SV_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, "sv_reduce_sum_h_fwd: bad arguments");
    hipLaunchKernelGGL(reduce_sum_h_fwd_k, dim3(ew_grid((int64_t)N * W * C)), dim3(256), 0, to_stream(stream),
                       (const float*)x, (float*)y, N, H, W, C);
    return check_launch("sv_reduce_sum_h_fwd");
}

int sv_reduce_sum_h_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_reduce_sum_h_bwd");
    SV_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "sv_reduce_sum_h_bwd: bad arguments");
    hipLaunchKernelGGL(reduce_sum_h_bwd_k, dim3(ew_grid((int64_t)N * H * W * C, 256, 8192)), dim3(256), 0,
                       to_stream(stream), (const float*)dy, (float*)dx, N, H, W, C);
    return check_launch("sv_reduce_sum_h_bwd");
}

int sv_gather_rows(const void* src, int src_is_f64, const int64_t* idx, float* dst, int64_t n_rows,
                   int64_t row_elems, sv_stream_t stream) {
    SV_REQUIRE(src && dst && n_rows > 0 && row_elems > 0, "sv_gather_rows: bad arguments");
    if (!src_is_f64 && (row_elems & 3) == 0 && n_rows * (row_elems >> 2) < (1ll << 31) &&
        ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        const int row_f4 = (int)(row_elems >> 2);
        hipLaunchKernelGGL(gather_rows_f4_k, dim3(ew_grid(n_rows * row_f4, 256, 16384)), dim3(256), 0, to_stream(stream),
                           (const float4*)src, idx, (float4*)dst, (int)n_rows, row_f4);
        return check_launch("sv_gather_rows");
    }
    const int grid = ew_grid(n_rows * row_elems, 256, 8192);
    if (src_is_f64)
        hipLaunchKernelGGL(gather_rows_k<double>, dim3(grid), dim3(256), 0, to_stream(stream), (const double*)src,
                           idx, dst, n_rows, row_elems);
    else
        hipLaunchKernelGGL(gather_rows_k<float>, dim3(grid), dim3(256), 0, to_stream(stream), (const float*)src, idx,
                           dst, n_rows, row_elems);
    return check_launch("sv_gather_rows");
}

int sv_random_block_mask(float* mask, int N, int H, int W, int C, double pct_mask, uint64_t seed, uint64_t offset,
                         sv_stream_t stream) {
    SV_REQUIRE(mask && N > 0 && H > 0 && W >= 8 && C > 0, "sv_random_block_mask: bad arguments (W >= 8 expected)");
    SV_REQUIRE(pct_mask >= 0.0 && pct_mask <= 0.9, "sv_random_block_mask: pct_mask must be within [0, 0.9]");
    SV_REQUIRE((double)H / (double)W > pct_mask, "sv_random_block_mask: %dx%d images cannot reach a coverage of %g", H, W,
               pct_mask);
    SV_REQUIRE((size_t)H * W <= 64 * 1024, "sv_random_block_mask: image too large for the LDS bitmap");
    hipLaunchKernelGGL(random_block_mask_k, dim3((unsigned)N), dim3(64), (size_t)H * W, to_stream(stream), mask, H, W, C,
                       pct_mask, seed, offset);
    return check_launch("sv_random_block_mask");
}

int sv_mask_input(const float* x, const float* mask, float* y, int64_t n, sv_stream_t stream) {
    SV_REQUIRE(x && mask && y && n > 0, "sv_mask_input: bad arguments");
    hipLaunchKernelGGL(mask_input_k, dim3(ew_grid(n, 256, 8192)), dim3(256), 0, to_stream(stream), x, mask, y, n);
    return check_launch("sv_mask_input");
}

int sv_fill_f32(float* p, float value, int64_t n, sv_stream_t stream) {
    SV_REQUIRE(p && n > 0, "sv_fill_f32: bad arguments");
    hipLaunchKernelGGL(fill_k, dim3(ew_grid(n, 256, 8192)), dim3(256), 0, to_stream(stream), p, value, n);
    return check_launch("sv_fill_f32");
}

int sv_axpy_f32(float* y, const float* x, float a, int64_t n, sv_stream_t stream) {
    SV_REQUIRE(x && y && n > 0, "sv_axpy_f32: bad arguments");
    hipLaunchKernelGGL(axpy_k, dim3(ew_grid(n, 256, 8192)), dim3(256), 0, to_stream(stream), y, x, a, n);
    return check_launch("sv_axpy_f32");
}

int sv_axpy(void* y, const void* x, float a, int64_t n, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_axpy");
    SV_REQUIRE(x && y && n > 0, "sv_axpy: bad arguments");
    SV_WITH_T(dtype, hipLaunchKernelGGL(axpy_t_k<T>, dim3(ew_grid(n, 256, 8192)), dim3(256), 0, to_stream(stream), (T*)y,
                                        (const T*)x, a, n););
    return check_launch("sv_axpy");
}

int sv_mask_input_dt(const float* x, const float* mask, void* y, int64_t n, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_mask_input_dt");
    SV_REQUIRE(x && mask && y && n > 0, "sv_mask_input_dt: bad arguments");
    SV_WITH_T(dtype, hipLaunchKernelGGL(mask_input_t_k<T>, dim3(ew_grid(n, 256, 8192)), dim3(256), 0, to_stream(stream), x,
                                        mask, (T*)y, n););
    return check_launch("sv_mask_input_dt");
}

int sv_cast_f32_bf16(const float* x, void* y, int64_t n, sv_stream_t stream) {
    SV_REQUIRE(x && y && n > 0, "sv_cast_f32_bf16: bad arguments");
    hipLaunchKernelGGL(cast_f32_bf16_k, dim3(ew_grid(n, 256, 8192)), dim3(256), 0, to_stream(stream), x,
                       (__hip_bfloat16*)y, n);
    return check_launch("sv_cast_f32_bf16");
}

int sv_cast_bf16_f32(const void* x, float* y, int64_t n, sv_stream_t stream) {
    SV_REQUIRE(x && y && n > 0, "sv_cast_bf16_f32: bad arguments");
    hipLaunchKernelGGL(cast_bf16_f32_k, dim3(ew_grid(n, 256, 8192)), dim3(256), 0, to_stream(stream),
                       (const __hip_bfloat16*)x, y, n);
    return check_launch("sv_cast_bf16_f32");
}

}  // extern "C"
