// Losses with fused dL/dprediction, the Keras-form Adam step, Dense (GEMM) and the Sampling/KL pair.  f32.
#include <stdarg.h>

#include "common.h"

namespace sv {

constexpr int kLossBlocks = 1024;

// ------------------------------------------------------------------------------------------------ losses
// mode 0: squared error, weight per pixel optional;  mode 1: absolute error;
// mode 2: BCE on probabilities against constant label; mode 3: BCE on logits against constant label.
// value_i is summed and scaled by inv_count; dpred_i = loss_scale * d(value_i)/dpred * inv_count.
// A launch of ONE workgroup (a discriminator's logits) writes the loss itself, rounded as loss_fold_k rounds a single
// partial sum; larger launches leave partials[blockIdx.x] for loss_fold_k.
template <int MODE, typename T>
__global__ __launch_bounds__(256) void loss_k(const float* __restrict__ t, const T* __restrict__ p,
                                               const float* __restrict__ weight, int wstride, float wscale, int C,
                                               float label, float inv_count, T* __restrict__ dpred,
                                               float loss_scale, int64_t n, float* __restrict__ partials,
                                               float* __restrict__ out, int accumulate, float* __restrict__ running,
                                               float running_weight) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v, g;
        if (MODE == 0) {
            const float d = (float)p[i] - t[i];
            const float w = weight != nullptr ? weight[(i / C) * wstride] * wscale : 1.f;
            v = w * d * d;
            g = 2.f * w * d;
        } else if (MODE == 1) {
            const float d = (float)p[i] - t[i];
            v = fabsf(d);
            g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        } else if (MODE == 2) {
            const float eps = 1e-7f;
            const float q0 = (float)p[i];
            const float q = fminf(fmaxf(q0, eps), 1.f - eps);
            v = -(label * logf(q + eps) + (1.f - label) * logf(1.f - q + eps));
            g = (q0 >= eps && q0 <= 1.f - eps) ? -(label / (q + eps) - (1.f - label) / (1.f - q + eps)) : 0.f;
        } else {
            const float z = (float)p[i];
            v = fmaxf(z, 0.f) - z * label + log1pf(expf(-fabsf(z)));
            g = 1.f / (1.f + expf(-z)) - label;
        }
        acc += v;
        if (dpred != nullptr) dpred[i] = (T)(loss_scale * g * inv_count);
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x != 0) return;
    if (gridDim.x > 1) {
        partials[blockIdx.x] = s;
        return;
    }
    const float v = (float)((double)s * (double)inv_count);
    const float r = accumulate ? out[0] + v : v;
    out[0] = r;
    if (running != nullptr) running[0] += running_weight * r;
}

// A scalar fold some later launch of the same stream carries out on behalf of a kernel that left per-workgroup partial sums
// (the fused forward + loss): out = [out +] scale * sum(partials[0..n)), running += running_weight * out.
struct ScalarFold {
    const float* partials;
    int n;
    float scale;
    float* out;
    int accumulate;
    float* running;
    float running_weight;
};

// 256 threads of one block: fixed-order tree over the partial sums
__device__ __forceinline__ void fold_scalar_256(const float* __restrict__ partials, int nblocks, float inv_count,
                                                float* __restrict__ out, int accumulate, float* __restrict__ running,
                                                float running_weight) {
    __shared__ double red[256];
    if (threadIdx.x < 256) {  // a larger block lends its first 256 threads: same order of additions
        double s = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += 256) s += (double)partials[b];
        red[threadIdx.x] = s;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float v = (float)(red[0] * (double)inv_count);
        const float r = accumulate ? out[0] + v : v;
        out[0] = r;
        if (running != nullptr) running[0] += running_weight * r;  // an epoch's running sum, in the same launch
    }
}

__global__ __launch_bounds__(256) void loss_fold_k(const float* __restrict__ partials, int nblocks, float inv_count,
                                                    float* __restrict__ out, int accumulate,
                                                    float* __restrict__ running = nullptr, float running_weight = 0.f) {
    fold_scalar_256(partials, nblocks, inv_count, out, accumulate, running, running_weight);
}

// A few scalar statements out_s = [out_s +] sum_k w_sk * in_sk, run in order by one lane: the loss bookkeeping of a GAN step
// (disc_loss = 0.5 lt + 0.5 lf; total = w l2 + (1 - w) disc_loss) in one launch instead of a fill and an axpy per term.
// Each term enters through one fused multiply-add, as in the axpy launches this replaces (bit-equal results).
constexpr int kScalarStatements = 4, kScalarTerms = 3;
struct ScalarProgram {
    int n;
    float* out[kScalarStatements];
    int accumulate[kScalarStatements];
    int n_in[kScalarStatements];
    const float* in[kScalarStatements][kScalarTerms];
    float w[kScalarStatements][kScalarTerms];
};
__global__ void scalar_program_k(ScalarProgram pr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int s = 0; s < pr.n; ++s) {
        float acc = pr.accumulate[s] ? pr.out[s][0] : 0.f;
        for (int k = 0; k < pr.n_in[s]; ++k) acc = fmaf(pr.w[s][k], pr.in[s][k][0], acc);
        pr.out[s][0] = acc;
    }
}

template <typename T>
__global__ void mse_map_k(const float* __restrict__ t, const T* __restrict__ p, const float* __restrict__ weight,
                          int wstride, float wscale, float* __restrict__ out, int64_t n_pix, int C) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
            const float d = (float)p[i * C + c] - t[i * C + c];
            s = fmaf(d, d, s);
        }
        s /= (float)C;
        if (weight != nullptr) s *= weight[i * wstride] * wscale;
        out[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------ Adam (Keras legacy form)
// Keras schedule and bias correction of update number it + 1 (run_model.py:77-89: ExponentialDecay, legacy Adam):
// lr = lr0 * decay_rate^(it / decay_steps), alpha = lr * sqrt(1 - beta2^t) / (1 - beta1^t).  b^t is evaluated as
// exp(t log b) in f64 (a shorter dependent chain than pow); one function, so that every kernel derives the same bits.
__device__ __forceinline__ void adam_schedule(int64_t it, float lr0, float beta1, float beta2, int decay_steps,
                                              float decay_rate, int staircase, float& alpha_out, float& lr_out) {
    double lr = (double)lr0;
    if (decay_steps > 0) {
        double e = (double)it / (double)decay_steps;
        if (staircase) e = floor(e);
        lr *= exp(e * log((double)decay_rate));
    }
    const double t = (double)(it + 1);
    alpha_out = (float)(lr * sqrt(-expm1(t * log((double)beta2))) / -expm1(t * log((double)beta1)));
    lr_out = (float)lr;
}

// state[0] = number of updates applied so far (int64); state[1] holds {alpha, lr} as two floats.
__global__ void adam_prepare_k(int64_t* __restrict__ state, float lr0, float beta1, float beta2, int decay_steps,
                               float decay_rate, int staircase) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t it = state[0];
        float* f = reinterpret_cast<float*>(state + 1);
        adam_schedule(it, lr0, beta1, beta2, decay_steps, decay_rate, staircase, f[0], f[1]);
        state[0] = it + 1;
    }
}

// One element's update with the roundings spelled out, so that every kernel using it (vector body or scalar tail,
// one-launch or two-launch form) produces the same bits.
__device__ __forceinline__ void adam_update1(float& th, float& m, float& v, float g, float alpha, float beta1, float beta2,
                                             float eps, float grad_scale) {
#pragma clang fp contract(off)
    const float gg = g * grad_scale;
    m = __fmaf_rn(beta1, m, (1.f - beta1) * gg);
    v = __fmaf_rn(beta2, v, ((1.f - beta2) * gg) * gg);
    th = th - (alpha * m) / (sqrtf(v) + eps);
}

__global__ void adam_apply_k(float* __restrict__ theta, float* __restrict__ grad, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, const int64_t* __restrict__ state, float beta1,
                             float beta2, float eps, float grad_scale, int zero_grad) {
    const float alpha = reinterpret_cast<const float*>(state + 1)[0];
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 g = reinterpret_cast<const float4*>(grad)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float4 th = reinterpret_cast<float4*>(theta)[i];
#define SV_ADAM1(f) adam_update1(th.f, mm.f, vv.f, g.f, alpha, beta1, beta2, eps, grad_scale);
        SV_ADAM1(x) SV_ADAM1(y) SV_ADAM1(z) SV_ADAM1(w)
#undef SV_ADAM1
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        reinterpret_cast<float4*>(theta)[i] = th;
        if (zero_grad) reinterpret_cast<float4*>(grad)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float th = theta[i], mi = m[i], vi = v[i];
        adam_update1(th, mi, vi, grad[i], alpha, beta1, beta2, eps, grad_scale);
        m[i] = mi;
        v[i] = vi;
        theta[i] = th;
        if (zero_grad) grad[i] = 0.f;
    }
}

// Schedule + update in ONE launch.  Every thread derives the step size from the update count it reads at its start; the
// count is advanced by the workgroup that finishes last (an arrival counter in the state's second word), i.e. after every
// workgroup has read it.  state: int64 count, {alpha, lr} of the last update as two floats (the layout the two-launch form
// keeps), then a third 8-byte word whose low half is the arrival counter (zero between launches).
// theta_lp (optional): a bf16 copy of theta refreshed in the same pass (the filters a bf16 model's convs read).
__global__ __launch_bounds__(256) void adam_step_k(float* __restrict__ theta, float* __restrict__ grad, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, int64_t* __restrict__ state, float lr0,
                                                    float beta1, float beta2, float eps, int decay_steps, float decay_rate,
                                                    int staircase, float grad_scale, int zero_grad,
                                                    __bf16* __restrict__ theta_lp, ScalarFold fold) {
    __shared__ float alpha_s, lr_s;
    // the last workgroup carries a pending scalar fold (it has the least parameter work)
    if (fold.partials != nullptr && blockIdx.x == gridDim.x - 1)
        fold_scalar_256(fold.partials, fold.n, fold.scale, fold.out, fold.accumulate, fold.running, fold.running_weight);
    const int64_t it = state[0];
    if (threadIdx.x == 0)  // a chain of dependent f64 operations, paid once per workgroup
        adam_schedule(it, lr0, beta1, beta2, decay_steps, decay_rate, staircase, alpha_s, lr_s);
    __syncthreads();
    const float alpha = alpha_s;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 g = reinterpret_cast<const float4*>(grad)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float4 th = reinterpret_cast<float4*>(theta)[i];
#define SV_ADAM1(f) adam_update1(th.f, mm.f, vv.f, g.f, alpha, beta1, beta2, eps, grad_scale);
        SV_ADAM1(x) SV_ADAM1(y) SV_ADAM1(z) SV_ADAM1(w)
#undef SV_ADAM1
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        reinterpret_cast<float4*>(theta)[i] = th;
        if (zero_grad) reinterpret_cast<float4*>(grad)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (theta_lp != nullptr) {
            theta_lp[4 * i] = (__bf16)th.x; theta_lp[4 * i + 1] = (__bf16)th.y;
            theta_lp[4 * i + 2] = (__bf16)th.z; theta_lp[4 * i + 3] = (__bf16)th.w;
        }
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float th = theta[i], mi = m[i], vi = v[i];
        adam_update1(th, mi, vi, grad[i], alpha, beta1, beta2, eps, grad_scale);
        m[i] = mi;
        v[i] = vi;
        theta[i] = th;
        if (zero_grad) grad[i] = 0.f;
        if (theta_lp != nullptr) theta_lp[i] = (__bf16)th;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int* ticket = reinterpret_cast<unsigned int*>(state + 2);
        const unsigned int prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {  // every workgroup has read the count: advance it for the next launch
            state[0] = it + 1;
            reinterpret_cast<float*>(state + 1)[0] = alpha_s;
            reinterpret_cast<float*>(state + 1)[1] = lr_s;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// adam_step_k whose gradients arrive as slabs of partial sums (sv_conv2d_wgrad_batch_slabs): a workgroup of 16 x 64
// threads owns 64 consecutive elements of one description; group g adds rows g, g + 16, ... (8 loads in flight), the
// groups meet in LDS and group 0 adds them in index order on top of grad[i] - the order wn_reduce_multi uses, so the
// result has the bits of the two-launch route - and applies the update.  The elements no description covers (alignment
// gaps of the flat buffer, variables of other layer kinds) are the `rest` workgroups' share.
constexpr int kAdamSlabJobs = 16;
struct AdamSlabJob {
    const float* slab;
    int64_t w_at, b_at;  // element index of dw / db in the flat buffers (b_at < 0: no bias)
    int slab_floats, n_dw, n_db, n_split;
};
struct AdamSlabArgs {
    float *theta, *grad, *m, *v;
    int64_t n;
    int64_t* state;
    float lr0, beta1, beta2, eps, decay_rate, grad_scale;
    int decay_steps, staircase, zero_grad;
    __bf16* theta_lp;
    ScalarFold fold;
    int n_jobs, rest_first, n_gaps;
    int first[kAdamSlabJobs + 1];
    int64_t gap_at[2 * kAdamSlabJobs + 1];
    int64_t gap_first[2 * kAdamSlabJobs + 2];  // prefix sums of the gap lengths
    AdamSlabJob job[kAdamSlabJobs];
};

__global__ __launch_bounds__(1024) void adam_slabs_k(AdamSlabArgs a) {
    __shared__ float red[16][64];
    __shared__ float alpha_s, lr_s;
    if (a.fold.partials != nullptr && blockIdx.x == gridDim.x - 1)
        fold_scalar_256(a.fold.partials, a.fold.n, a.fold.scale, a.fold.out, a.fold.accumulate, a.fold.running,
                        a.fold.running_weight);
    // The schedule (one thread) and the element's parameter / moment loads do not depend on the slab sums: they are
    // issued first, so the launch is two global round trips deep (slabs beside state, then the stores), not four.
    const int64_t it = a.state[0];
    auto apply = [&](int64_t i, float th, float mi, float vi, float g, float alpha) {
        adam_update1(th, mi, vi, g, alpha, a.beta1, a.beta2, a.eps, a.grad_scale);
        a.m[i] = mi;
        a.v[i] = vi;
        a.theta[i] = th;
        if (a.zero_grad) a.grad[i] = 0.f;
        if (a.theta_lp != nullptr) a.theta_lp[i] = (__bf16)th;
    };
    if ((int)blockIdx.x < a.rest_first) {
        int j = 0;
        while (j + 1 < a.n_jobs && (int)blockIdx.x >= a.first[j + 1]) ++j;
        // the job's fields come from the kernel-argument segment at a wave-uniform offset (scalar loads); indexing the
        // by-value table with j would make the compiler copy all of it to per-lane scratch memory first
        typedef const __attribute__((address_space(4))) AdamSlabArgs* KArgs;
        const KArgs ap = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
        AdamSlabJob q;
        __builtin_memcpy(&q, &ap->job[j], sizeof(AdamSlabJob));
        const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
        const int e = ((int)blockIdx.x - a.first[j]) * 64 + lane;
        const int ne = q.n_dw + q.n_db;
        const int64_t i = e >= ne ? -1 : (e < q.n_dw ? q.w_at + e : (q.b_at >= 0 ? q.b_at + (e - q.n_dw) : -1));
        float th = 0.f, mi = 0.f, vi = 0.f, g0 = 0.f;
        if (grp == 0 && i >= 0) { th = a.theta[i]; mi = a.m[i]; vi = a.v[i]; g0 = a.grad[i]; }
        float s = 0.f;
        if (e < ne) {
            int sp = grp;
            for (; sp + 7 * 16 < q.n_split; sp += 8 * 16) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = q.slab[(size_t)(sp + u * 16) * q.slab_floats + e];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += t[u];
            }
            for (; sp < q.n_split; sp += 16) s += q.slab[(size_t)sp * q.slab_floats + e];
        }
        red[grp][lane] = s;
        if (threadIdx.x == 0) adam_schedule(it, a.lr0, a.beta1, a.beta2, a.decay_steps, a.decay_rate, a.staircase, alpha_s, lr_s);
        __syncthreads();
        if (grp == 0 && i >= 0) {
            float t = g0;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += red[g][lane];
            apply(i, th, mi, vi, t, alpha_s);
        }
    } else {
        if (threadIdx.x == 0) adam_schedule(it, a.lr0, a.beta1, a.beta2, a.decay_steps, a.decay_rate, a.staircase, alpha_s, lr_s);
        __syncthreads();
        const float alpha = alpha_s;
        const int64_t total = a.gap_first[a.n_gaps];
        const int64_t stride = (int64_t)(gridDim.x - a.rest_first) * 1024;
        for (int64_t p = (int64_t)((int)blockIdx.x - a.rest_first) * 1024 + threadIdx.x; p < total; p += stride) {
            int k = 0;
            while (k + 1 < a.n_gaps && p >= a.gap_first[k + 1]) ++k;
            const int64_t i = a.gap_at[k] + (p - a.gap_first[k]);
            apply(i, a.theta[i], a.m[i], a.v[i], a.grad[i], alpha);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int* ticket = reinterpret_cast<unsigned int*>(a.state + 2);
        const unsigned int prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            a.state[0] = it + 1;
            reinterpret_cast<float*>(a.state + 1)[0] = alpha_s;
            reinterpret_cast<float*>(a.state + 1)[1] = lr_s;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------ GEMM (Dense)
// C[m,n] (+)= epi( sum_k A(m,k) * B(k,n) ), element (m,k) of A at a[m*am + k*ak], (k,n) of B at b[k*bk + n*bn].
// 64x64 tile, BK = 16, 256 threads; the products run on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gemm_k(const float* __restrict__ a, int64_t am, int64_t ak,
                                               const float* __restrict__ b, int64_t bk, int64_t bn,
                                               const float* __restrict__ bias, float* __restrict__ c, int M, int N,
                                               int K, int act, float alpha, int accumulate, int k_per_split) {
    __shared__ float As[16][64 + 4];
    __shared__ float Bs[16][64 + 4];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    // blockIdx.z > 0 only with a cut reduction: this workgroup sums k in [kbeg, kend) and stores raw sums into slab z
    const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
    const bool slab = gridDim.z > 1;
    // 4 waves as 2 x 2, each 32 x 32 of the tile = 2 x 2 matrix-core tiles of 16 x 16 (k step 4)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    v4f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    // 1024 elements per tile and operand, 4 per thread; the unit-stride dimension is the fastest-varying one.  The
    // loads of step k0 + 16 are in flight while step k0 is multiplied (register prefetch).
    float ra[4], rb[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int id = threadIdx.x + e * 256;
            const int mm = ak == 1 ? id >> 4 : id & 63, kk = ak == 1 ? id & 15 : id >> 6;
            const int gm = m0 + mm, gk = k0 + kk;
            ra[e] = (gm < M && gk < kend) ? a[gm * am + gk * ak] : 0.f;
            const int nn = bk == 1 ? id >> 4 : id & 63, k2 = bk == 1 ? id & 15 : id >> 6;
            const int gn = n0 + nn, gk2 = k0 + k2;
            rb[e] = (gn < N && gk2 < kend) ? b[gk2 * bk + gn * bn] : 0.f;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int id = threadIdx.x + e * 256;
            As[ak == 1 ? id & 15 : id >> 6][ak == 1 ? id >> 4 : id & 63] = ra[e];
            Bs[bk == 1 ? id & 15 : id >> 6][bk == 1 ? id >> 4 : id & 63] = rb[e];
        }
    };
    if (kbeg < kend) load(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        __syncthreads();  // the previous step's operand reads are done
        store();
        __syncthreads();
        if (k0 + 16 < kend) load(k0 + 16);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = As[kk + lk][wm + 16 * i + li];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = Bs[kk + lk][wn + 16 * j + li];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    // lane (li, lk) holds rows 4 lk .. 4 lk + 3, column li of each 16 x 16 tile
    float* out = slab ? c + (size_t)blockIdx.z * M * N : c;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wn + 16 * j + li;
            if (gn >= N) continue;
            const float bb = (!slab && bias != nullptr) ? bias[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm + 16 * i + 4 * lk + r;
                if (gm >= M) continue;
                float* cp = out + (int64_t)gm * N + gn;
                if (slab) {
                    *cp = acc[i][j][r];
                } else {
                    const float v = act_apply(acc[i][j][r] + bb, act, alpha);
                    *cp = accumulate ? *cp + v : v;
                }
            }
        }
}

int launch_gemm(const float* a, int64_t am, int64_t ak, const float* b, int64_t bk, int64_t bn, const float* bias,
                float* c, int M, int N, int K, int act, float alpha, int accumulate, hipStream_t st) {
    dim3 grid((unsigned)ceil_div(N, 64), (unsigned)ceil_div(M, 64));
    hipLaunchKernelGGL(gemm_k, grid, dim3(256), 0, st, a, am, ak, b, bk, bn, bias, c, M, N, K, act, alpha,
                       accumulate, K);
    return check_launch("gemm");
}

// How many ways the reduction of a GEMM with a small result is cut: about 256 workgroups, at least 32 terms each.
static int gemm_splits(int M, int N, int K) {
    const int64_t tiles = ceil_div(M, 64) * ceil_div(N, 64);
    int64_t s = ceil_div(256, tiles);
    const int64_t maxs = K / 32;
    if (s > maxs) s = maxs;
    return s < 2 ? 1 : (int)s;
}

int launch_slab_sum(const float* part, float* dst, int64_t n, int slabs, int accumulate, hipStream_t st);  // conv_direct.hip

// ------------------------------------------------------------------------------------------------ Sampling + KL
__global__ __launch_bounds__(256) void reparam_fwd_k(const float* __restrict__ mean, const float* __restrict__ logvar,
                                                      const float* __restrict__ eps_in, float* __restrict__ z,
                                                      float* __restrict__ eps_out, float* __restrict__ partials,
                                                      int64_t n, uint64_t seed, uint64_t offset,
                                                      const uint64_t* __restrict__ counter) {
    __shared__ float red[4];
    float acc = 0.f;
    if (counter != nullptr) offset += counter[0];  // advanced by the fold launch that follows, after every read here
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float e;
        if (eps_in != nullptr) {
            e = eps_in[i];
        } else {  // Box-Muller on two counter-hash uniforms
            const float u1 = fmaxf(u01(seed, 2 * (offset + (uint64_t)i)), 5.9604645e-8f);
            const float u2 = u01(seed, 2 * (offset + (uint64_t)i) + 1);
            e = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
        }
        const float mu = mean[i], lv = logvar[i];
        z[i] = mu + expf(0.5f * lv) * e;
        if (eps_out != nullptr) eps_out[i] = e;
        acc += -0.5f * (1.f + lv - mu * mu - expf(lv));
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// the KL fold; also moves the device-side random-stream position past the n values the sampling launch just drew
__global__ __launch_bounds__(256) void reparam_fold_k(const float* __restrict__ partials, int nblocks, float inv_b,
                                                       float* __restrict__ kl_out, uint64_t* __restrict__ counter,
                                                       uint64_t n) {
    fold_scalar_256(partials, nblocks, inv_b, kl_out, 0, nullptr, 0.f);
    if (counter != nullptr && threadIdx.x == 0) counter[0] += n;
}

__global__ void reparam_bwd_k(const float* __restrict__ mean, const float* __restrict__ logvar,
                              const float* __restrict__ eps, const float* __restrict__ dz, float* __restrict__ dmean,
                              float* __restrict__ dlogvar, float klw_over_b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float mu = mean[i], lv = logvar[i], g = dz[i];
        dmean[i] = g + klw_over_b * mu;
        dlogvar[i] = g * 0.5f * expf(0.5f * lv) * eps[i] + klw_over_b * 0.5f * (expf(lv) - 1.f);
    }
}

}  // namespace sv

using namespace sv;

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

size_t sv_loss_workspace_bytes(void) { return (size_t)kLossBlocks * sizeof(float); }

int sv_loss_mse(const void* y_true, const void* y_pred, const float* weight, int weight_stride, float weight_scale,
                float* loss_out, int accumulate, void* dpred, float loss_scale, int64_t n_pix, int C, float* partials,
                float* running, float running_weight, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_loss_mse");
    SV_REQUIRE(y_true && y_pred && loss_out && partials && n_pix > 0 && C > 0, "sv_loss_mse: bad arguments");
    SV_REQUIRE(weight == nullptr || weight_stride > 0, "sv_loss_mse: weight_stride must be positive");
    const int64_t n = n_pix * C;
    const int nb = ew_grid(n, 256, kLossBlocks);
    const float inv = 1.f / (float)n;
    SV_WITH_T(dtype, hipLaunchKernelGGL((loss_k<0, T>), dim3(nb), dim3(256), 0, to_stream(stream), (const float*)y_true,
                                        (const T*)y_pred, weight, weight_stride, weight_scale, C, 0.f, inv, (T*)dpred,
                                        loss_scale, n, partials, loss_out, accumulate, running, running_weight););
    int rc = check_launch("sv_loss_mse");
    if (rc || nb == 1) return rc;
    hipLaunchKernelGGL(loss_fold_k, dim3(1), dim3(256), 0, to_stream(stream), partials, nb, inv, loss_out, accumulate,
                       running, running_weight);
    return check_launch("sv_loss_mse/fold");
}

int sv_scalar_program(int n_statements, float* const* outs, const int* accumulate, const int* n_terms,
                      const float* const* terms, const float* weights, sv_stream_t stream) {
    SV_REQUIRE(n_statements >= 1 && n_statements <= kScalarStatements && outs && accumulate && n_terms && terms && weights,
               "sv_scalar_program: 1..4 statements, non-null tables");
    ScalarProgram pr{};
    pr.n = n_statements;
    int at = 0;
    for (int s = 0; s < n_statements; ++s) {
        SV_REQUIRE(outs[s] != nullptr && n_terms[s] >= 0 && n_terms[s] <= kScalarTerms,
                   "sv_scalar_program: a statement has a null result or more than 3 terms");
        pr.out[s] = outs[s];
        pr.accumulate[s] = accumulate[s];
        pr.n_in[s] = n_terms[s];
        for (int k = 0; k < n_terms[s]; ++k, ++at) {
            SV_REQUIRE(terms[at] != nullptr, "sv_scalar_program: null term");
            pr.in[s][k] = terms[at];
            pr.w[s][k] = weights[at];
        }
    }
    hipLaunchKernelGGL(scalar_program_k, dim3(1), dim3(64), 0, to_stream(stream), pr);
    return check_launch("sv_scalar_program");
}

int sv_loss_mse_map(const void* y_true, const void* y_pred, const float* weight, int weight_stride,
                    float weight_scale, float* out, int64_t n_pix, int C, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_loss_mse_map");
    SV_REQUIRE(y_true && y_pred && out && n_pix > 0 && C > 0, "sv_loss_mse_map: bad arguments");
    SV_WITH_T(dtype, hipLaunchKernelGGL(mse_map_k<T>, dim3(ew_grid(n_pix, 256, 8192)), dim3(256), 0, to_stream(stream),
                                        (const float*)y_true, (const T*)y_pred, weight, weight_stride, weight_scale, out,
                                        n_pix, C););
    return check_launch("sv_loss_mse_map");
}

int sv_loss_mae(const void* y_true, const void* y_pred, float* loss_out, int accumulate, void* dpred,
                float loss_scale, int64_t n, float* partials, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_loss_mae");
    SV_REQUIRE(y_true && y_pred && loss_out && partials && n > 0, "sv_loss_mae: bad arguments");
    const int nb = ew_grid(n, 256, kLossBlocks);
    const float inv = 1.f / (float)n;
    SV_WITH_T(dtype, hipLaunchKernelGGL((loss_k<1, T>), dim3(nb), dim3(256), 0, to_stream(stream), (const float*)y_true,
                                        (const T*)y_pred, (const float*)nullptr, 1, 1.f, 1, 0.f, inv, (T*)dpred, loss_scale,
                                        n, partials, loss_out, accumulate, (float*)nullptr, 0.f););
    int rc = check_launch("sv_loss_mae");
    if (rc || nb == 1) return rc;
    hipLaunchKernelGGL(loss_fold_k, dim3(1), dim3(256), 0, to_stream(stream), partials, nb, inv, loss_out, accumulate);
    return check_launch("sv_loss_mae/fold");
}

int sv_loss_bce(const void* pred, float label, int from_logits, float* loss_out, int accumulate, void* dpred,
                float loss_scale, int64_t n, float* partials, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_loss_bce");
    SV_REQUIRE(pred && loss_out && partials && n > 0, "sv_loss_bce: bad arguments");
    const int nb = ew_grid(n, 256, kLossBlocks);
    const float inv = 1.f / (float)n;
    if (from_logits)
        SV_WITH_T(dtype, hipLaunchKernelGGL((loss_k<3, T>), dim3(nb), dim3(256), 0, to_stream(stream), (const float*)nullptr,
                                            (const T*)pred, (const float*)nullptr, 1, 1.f, 1, label, inv, (T*)dpred,
                                            loss_scale, n, partials, loss_out, accumulate, (float*)nullptr, 0.f););
    else
        SV_WITH_T(dtype, hipLaunchKernelGGL((loss_k<2, T>), dim3(nb), dim3(256), 0, to_stream(stream), (const float*)nullptr,
                                            (const T*)pred, (const float*)nullptr, 1, 1.f, 1, label, inv, (T*)dpred,
                                            loss_scale, n, partials, loss_out, accumulate, (float*)nullptr, 0.f););
    int rc = check_launch("sv_loss_bce");
    if (rc || nb == 1) return rc;
    hipLaunchKernelGGL(loss_fold_k, dim3(1), dim3(256), 0, to_stream(stream), partials, nb, inv, loss_out, accumulate);
    return check_launch("sv_loss_bce/fold");
}

int sv_adam_keras_prepare(int64_t* step_state, float lr0, float beta1, float beta2, int decay_steps,
                          float decay_rate, int staircase, sv_stream_t stream) {
    SV_REQUIRE(step_state != nullptr, "sv_adam_keras_prepare: null state");
    SV_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "sv_adam_keras_prepare: betas out of range");
    hipLaunchKernelGGL(adam_prepare_k, dim3(1), dim3(64), 0, to_stream(stream), step_state, lr0, beta1, beta2,
                       decay_steps, decay_rate, staircase);
    return check_launch("sv_adam_keras_prepare");
}

int sv_adam_keras_apply(float* theta, float* grad, float* m, float* v, int64_t n, const int64_t* step_state,
                        float beta1, float beta2, float eps, float grad_scale, int zero_grad, sv_stream_t stream) {
    SV_REQUIRE(theta && grad && m && v && step_state && n > 0, "sv_adam_keras_apply: bad arguments");
    SV_REQUIRE(((uintptr_t)theta % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                   ((uintptr_t)v % 16 == 0),
               "sv_adam_keras_apply: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adam_apply_k, dim3(ew_grid(n / 4 + 1, 256, 4096)), dim3(256), 0, to_stream(stream), theta,
                       grad, m, v, n, step_state, beta1, beta2, eps, grad_scale, zero_grad);
    return check_launch("sv_adam_keras_apply");
}

int sv_fold_scalar(const float* partials, int n, float scale, float* out, int accumulate, float* running,
                   float running_weight, sv_stream_t stream) {
    SV_REQUIRE(partials && out && n > 0, "sv_fold_scalar: bad arguments");
    hipLaunchKernelGGL(loss_fold_k, dim3(1), dim3(256), 0, to_stream(stream), partials, n, scale, out, accumulate, running,
                       running_weight);
    return check_launch("sv_fold_scalar");
}

int sv_adam_keras_step(float* theta, float* grad, float* m, float* v, int64_t n, int64_t* step_state, float lr0,
                       float beta1, float beta2, float eps, int decay_steps, float decay_rate, int staircase,
                       float grad_scale, int zero_grad, void* theta_bf16, const float* fold_partials, int fold_n,
                       float fold_scale, float* fold_out, float* fold_running, float fold_running_weight,
                       sv_stream_t stream) {
    SV_REQUIRE(theta && grad && m && v && step_state && n > 0, "sv_adam_keras_step: bad arguments");
    SV_REQUIRE(fold_partials == nullptr || (fold_n > 0 && fold_out != nullptr), "sv_adam_keras_step: bad fold arguments");
    SV_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "sv_adam_keras_step: betas out of range");
    SV_REQUIRE(((uintptr_t)theta % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                   ((uintptr_t)v % 16 == 0),
               "sv_adam_keras_step: buffers must be 16-byte aligned");
    const ScalarFold fold{fold_partials, fold_n, fold_scale, fold_out, 0, fold_running, fold_running_weight};
    // few workgroups (about one per CU): each pays the schedule's latency once and then streams its share
    hipLaunchKernelGGL(adam_step_k, dim3(ew_grid(n / 4 + 1, 256, 256)), dim3(256), 0, to_stream(stream), theta, grad, m, v,
                       n, step_state, lr0, beta1, beta2, eps, decay_steps, decay_rate, staircase, grad_scale, zero_grad,
                       (__bf16*)theta_bf16, fold);
    return check_launch("sv_adam_keras_step");
}

int sv_adam_keras_step_slabs(float* theta, float* grad, float* m, float* v, int64_t n, int64_t* step_state, float lr0,
                             float beta1, float beta2, float eps, int decay_steps, float decay_rate, int staircase,
                             float grad_scale, int zero_grad, void* theta_bf16, const float* fold_partials, int fold_n,
                             float fold_scale, float* fold_out, float* fold_running, float fold_running_weight,
                             const sv_grad_slabs* slabs, int n_slabs, sv_stream_t stream) {
    SV_REQUIRE(theta && grad && m && v && step_state && n > 0, "sv_adam_keras_step_slabs: bad arguments");
    SV_REQUIRE(fold_partials == nullptr || (fold_n > 0 && fold_out != nullptr), "sv_adam_keras_step_slabs: bad fold arguments");
    SV_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "sv_adam_keras_step_slabs: betas out of range");
    SV_REQUIRE(slabs != nullptr && n_slabs >= 1 && n_slabs <= kAdamSlabJobs, "sv_adam_keras_step_slabs: 1..%d descriptions", kAdamSlabJobs);
    AdamSlabArgs a{};
    a.theta = theta; a.grad = grad; a.m = m; a.v = v; a.n = n; a.state = step_state;
    a.lr0 = lr0; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.decay_rate = decay_rate; a.grad_scale = grad_scale;
    a.decay_steps = decay_steps; a.staircase = staircase; a.zero_grad = zero_grad; a.theta_lp = (__bf16*)theta_bf16;
    a.fold = ScalarFold{fold_partials, fold_n, fold_scale, fold_out, 0, fold_running, fold_running_weight};
    a.n_jobs = n_slabs;
    // covered ranges of the flat buffer, in ascending order, and the gaps between them
    struct Range { int64_t at, len; };
    Range rg[2 * kAdamSlabJobs];
    int nr = 0, blocks = 0;
    for (int j = 0; j < n_slabs; ++j) {
        const sv_grad_slabs& d = slabs[j];
        SV_REQUIRE(d.slabs && d.dw && d.n_split >= 1 && d.n_dw >= 1 && d.n_db >= 0 && d.slab_floats >= d.n_dw + d.n_db,
                   "sv_adam_keras_step_slabs: description %d is malformed", j);
        AdamSlabJob& q = a.job[j];
        q.slab = (const float*)d.slabs; q.slab_floats = d.slab_floats; q.n_dw = d.n_dw; q.n_db = d.n_db; q.n_split = d.n_split;
        q.w_at = (const float*)d.dw - grad;
        q.b_at = d.db != nullptr ? (const float*)d.db - grad : -1;
        SV_REQUIRE(q.w_at >= 0 && q.w_at + d.n_dw <= n, "sv_adam_keras_step_slabs: description %d: dw is not inside grad", j);
        SV_REQUIRE(d.db == nullptr || (q.b_at >= 0 && q.b_at + d.n_db <= n),
                   "sv_adam_keras_step_slabs: description %d: db is not inside grad", j);
        rg[nr++] = Range{q.w_at, d.n_dw};
        if (d.db != nullptr && d.n_db > 0) rg[nr++] = Range{q.b_at, d.n_db};
        a.first[j] = blocks;
        blocks += (d.n_dw + d.n_db + 63) / 64;
    }
    a.first[n_slabs] = blocks;
    a.rest_first = blocks;
    for (int i = 1; i < nr; ++i)
        for (int k = i; k > 0 && rg[k].at < rg[k - 1].at; --k) { const Range t = rg[k]; rg[k] = rg[k - 1]; rg[k - 1] = t; }
    int64_t at = 0, total = 0;
    a.n_gaps = 0;
    for (int i = 0; i <= nr; ++i) {
        const int64_t end = i < nr ? rg[i].at : n;
        SV_REQUIRE(end >= at, "sv_adam_keras_step_slabs: descriptions overlap");
        if (end > at) {
            a.gap_at[a.n_gaps] = at;
            a.gap_first[a.n_gaps] = total;
            total += end - at;
            ++a.n_gaps;
        }
        if (i < nr) at = rg[i].at + rg[i].len;
    }
    a.gap_first[a.n_gaps] = total;
    // the last workgroup carries the scalar fold: keep at least one `rest` workgroup (it has the least to do)
    int rest = (int)ceil_div(total, 4096);
    if (rest < 1) rest = 1;
    if (rest > 256) rest = 256;
    hipLaunchKernelGGL(adam_slabs_k, dim3((unsigned)(blocks + rest)), dim3(1024), 0, to_stream(stream), a);
    return check_launch("sv_adam_keras_step_slabs");
}

int sv_dense_fwd(const void* x, const void* w, const void* bias, void* y, int M, int K, int N, int act, float alpha,
                 int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_dense_fwd");
    SV_REQUIRE(x && w && y && M > 0 && K > 0 && N > 0, "sv_dense_fwd: bad arguments");
    return launch_gemm((const float*)x, K, 1, (const float*)w, N, 1, (const float*)bias, (float*)y, M, N, K, act,
                       alpha, 0, to_stream(stream));
}

// dx[M,K] = dz[M,N] w[K,N]^T
int sv_dense_dgrad(const void* dz, const void* w, void* dx, int M, int K, int N, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_dense_dgrad");
    SV_REQUIRE(dz && w && dx && M > 0 && K > 0 && N > 0, "sv_dense_dgrad: bad arguments");
    return launch_gemm((const float*)dz, N, 1, (const float*)w, 1, N, nullptr, (float*)dx, M, K, N, SV_ACT_NONE, 0.f,
                       0, to_stream(stream));
}

// dw[K,N] (+)= x[M,K]^T dz[M,N]
size_t sv_dense_wgrad_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    const int s = gemm_splits(K, N, M);
    return s > 1 ? (size_t)s * K * N * sizeof(float) : 0;
}

int sv_dense_wgrad_ws(const void* x, const void* dz, void* dw, int accumulate, int M, int K, int N, void* workspace,
                      size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_dense_wgrad_ws");
    SV_REQUIRE(x && dz && dw && M > 0 && K > 0 && N > 0, "sv_dense_wgrad_ws: bad arguments");
    const int splits = gemm_splits(K, N, M);
    const size_t need = sv_dense_wgrad_workspace_bytes(M, K, N);
    if (splits < 2 || workspace == nullptr || workspace_bytes < need)  // not an error: the reduction then stays whole
        return sv_dense_wgrad(x, dz, dw, accumulate, M, K, N, dtype, stream);
    // dw[K, N] = x[M, K]^T dz[M, N]: the batch is the reduction, cut into `splits` ranges of whole 16-steps
    const int per = (int)(ceil_div(ceil_div(M, splits), 16) * 16);
    dim3 grid((unsigned)ceil_div(N, 64), (unsigned)ceil_div(K, 64), (unsigned)ceil_div(M, per));
    hipLaunchKernelGGL(gemm_k, grid, dim3(256), 0, to_stream(stream), (const float*)x, (int64_t)1, (int64_t)K,
                       (const float*)dz, (int64_t)N, (int64_t)1, (const float*)nullptr, (float*)workspace, K, N, M,
                       SV_ACT_NONE, 0.f, 0, per);
    int rc = check_launch("gemm (cut reduction)");
    if (rc) return rc;
    if (grid.z == 1) {  // a single range wrote plain sums into slab 0
        return launch_slab_sum((const float*)workspace, (float*)dw, (int64_t)K * N, 1, accumulate, to_stream(stream));
    }
    return launch_slab_sum((const float*)workspace, (float*)dw, (int64_t)K * N, (int)grid.z, accumulate, to_stream(stream));
}

int sv_dense_wgrad(const void* x, const void* dz, void* dw, int accumulate, int M, int K, int N, int dtype,
                   sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_dense_wgrad");
    SV_REQUIRE(x && dz && dw && M > 0 && K > 0 && N > 0, "sv_dense_wgrad: bad arguments");
    return launch_gemm((const float*)x, 1, K, (const float*)dz, N, 1, nullptr, (float*)dw, K, N, M, SV_ACT_NONE, 0.f,
                       accumulate, to_stream(stream));
}

int sv_reparam_kl_fwd(const float* mean, const float* logvar, const float* eps_in, float* z, float* eps_out,
                      float* kl_out, float* partials, int B, int D, uint64_t seed, uint64_t offset,
                      uint64_t* rng_counter, sv_stream_t stream) {
    SV_REQUIRE(mean && logvar && z && kl_out && partials && B > 0 && D > 0, "sv_reparam_kl_fwd: bad arguments");
    const int64_t n = (int64_t)B * D;
    const int nb = ew_grid(n, 256, kLossBlocks);
    hipLaunchKernelGGL(reparam_fwd_k, dim3(nb), dim3(256), 0, to_stream(stream), mean, logvar, eps_in, z, eps_out,
                       partials, n, seed, offset, (const uint64_t*)rng_counter);
    int rc = check_launch("sv_reparam_kl_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(reparam_fold_k, dim3(1), dim3(256), 0, to_stream(stream), partials, nb, 1.f / (float)B, kl_out,
                       eps_in == nullptr ? rng_counter : nullptr, (uint64_t)n);
    return check_launch("sv_reparam_kl_fwd/fold");
}

int sv_reparam_kl_bwd(const float* mean, const float* logvar, const float* eps, const float* dz, float* dmean,
                      float* dlogvar, float kl_weight, int B, int D, sv_stream_t stream) {
    SV_REQUIRE(mean && logvar && eps && dz && dmean && dlogvar && B > 0 && D > 0, "sv_reparam_kl_bwd: bad arguments");
    const int64_t n = (int64_t)B * D;
    hipLaunchKernelGGL(reparam_bwd_k, dim3(ew_grid(n)), dim3(256), 0, to_stream(stream), mean, logvar, eps, dz,
                       dmean, dlogvar, kl_weight / (float)B, n);
    return check_launch("sv_reparam_kl_bwd");
}

}  // extern "C"
