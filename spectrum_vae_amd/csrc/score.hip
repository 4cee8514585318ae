// Evaluation after training: per-sample reconstruction error and the ROC AUC of two score sets, both on the device
// (run_model.py:356-398).  HBM-bound streaming / integer compare work.
#include "common.h"

namespace sv {

// out[n] = mean over the sample's elements of (pred - true)^2: tf.reduce_mean(losses.mse(pred, true), axis=(1, 2))
// (run_model.py:361-362; the mean over channels and then over pixels is the mean over all elements of the sample).
// One workgroup per sample, 16-byte loads when the sample is a whole number of 4-element pieces.
template <typename T>
__global__ __launch_bounds__(256) void sample_mse_k(const float* __restrict__ t, const T* __restrict__ p,
                                                     float* __restrict__ out, int64_t elems) {
    __shared__ float red[4];
    const int64_t base = (int64_t)blockIdx.x * elems;
    float acc = 0.f;
    if (sizeof(T) == 4 && (elems & 3) == 0) {
        const float4* t4 = reinterpret_cast<const float4*>(t + base);
        const float4* p4 = reinterpret_cast<const float4*>(p + base);
        for (int64_t i = threadIdx.x; i < (elems >> 2); i += 256) {
            const float4 a = t4[i], b = p4[i];
            const float d0 = b.x - a.x, d1 = b.y - a.y, d2 = b.z - a.z, d3 = b.w - a.w;
            acc += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    } else {
        for (int64_t i = threadIdx.x; i < elems; i += 256) {
            const float d = (float)p[base + i] - t[base + i];
            acc += d * d;
        }
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)((double)s / (double)elems);
}

// Pair counts behind the AUC of sklearn.metrics.roc_curve + auc (run_model.py:390-398) with y_true = 1 for set a and 0 for
// set b and the score as y_pred: AUC = (#{(i, j): a_i > b_j} + 0.5 #{(i, j): a_i == b_j}) / (na nb).  Integer counts, so the
// result does not depend on the order of the sums.  A workgroup owns 256 entries of a and sweeps b through LDS.
__global__ __launch_bounds__(256) void auc_pair_counts_k(const float* __restrict__ a, int64_t na,
                                                          const float* __restrict__ b, int64_t nb,
                                                          unsigned long long* __restrict__ counts) {
    __shared__ float tile[1024];
    __shared__ unsigned long long red[8];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float ai = i < na ? a[i] : 0.f;
    unsigned long long gt = 0, eq = 0;
    for (int64_t j0 = 0; j0 < nb; j0 += 1024) {
        const int m = (int)(nb - j0 < 1024 ? nb - j0 : 1024);
        __syncthreads();
        for (int j = threadIdx.x; j < m; j += 256) tile[j] = b[j0 + j];
        __syncthreads();
        unsigned g = 0, e = 0;
        for (int j = 0; j < m; ++j) {
            const float bj = tile[j];
            g += ai > bj ? 1u : 0u;
            e += ai == bj ? 1u : 0u;
        }
        gt += g;
        eq += e;
    }
    if (i >= na) gt = eq = 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        gt += __shfl_xor(gt, o, 64);
        eq += __shfl_xor(eq, o, 64);
    }
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wid] = gt;
        red[4 + wid] = eq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&counts[0], red[0] + red[1] + red[2] + red[3]);
        atomicAdd(&counts[1], red[4] + red[5] + red[6] + red[7]);
    }
}

}  // namespace sv

using namespace sv;

extern "C" {

int sv_sample_mse(const float* y_true, const void* y_pred, float* out, int64_t n_samples, int64_t sample_elems,
                  int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_sample_mse");
    SV_REQUIRE(y_true && y_pred && out && n_samples > 0 && sample_elems > 0, "sv_sample_mse: bad arguments");
    SV_REQUIRE(n_samples <= 0x7fffffff, "sv_sample_mse: more than 2^31 - 1 samples");
    hipStream_t st = to_stream(stream);
    if (dtype == SV_F32)
        sample_mse_k<float><<<(int)n_samples, 256, 0, st>>>(y_true, (const float*)y_pred, out, sample_elems);
    else
        sample_mse_k<__hip_bfloat16><<<(int)n_samples, 256, 0, st>>>(y_true, (const __hip_bfloat16*)y_pred, out,
                                                                      sample_elems);
    return check_launch("sv_sample_mse");
}

int sv_roc_auc_counts(const float* scores_label1, int64_t n1, const float* scores_label0, int64_t n0,
                      uint64_t* counts, sv_stream_t stream) {
    SV_REQUIRE(scores_label1 && scores_label0 && counts && n1 > 0 && n0 > 0, "sv_roc_auc_counts: bad arguments");
    SV_REQUIRE(ceil_div(n1, 256) <= 0x7fffffff, "sv_roc_auc_counts: too many scores");
    hipStream_t st = to_stream(stream);
    hipError_t e = hipMemsetAsync(counts, 0, 2 * sizeof(uint64_t), st);
    if (e != hipSuccess) return fail(SV_E_LAUNCH, "sv_roc_auc_counts: memset failed: %s", hipGetErrorString(e));
    auc_pair_counts_k<<<(int)ceil_div(n1, 256), 256, 0, st>>>(scores_label1, n1, scores_label0, n0,
                                                              (unsigned long long*)counts);
    return check_launch("sv_roc_auc_counts");
}

}  // extern "C"
