// Data preparation on the device: the step immediately upstream of the training path.
//
//   process_regions_k   util.py:152-225 (process_region) with util.py:34-45 (inter_snp_distances) and util.py:228-256
//                       (major_minor) folded in: one workgroup turns one ragged genotype array (n_sites, n_haps, n_mut)
//                       + its int64 positions into one (n_haps, S, C) float32 training image.
//   sort_min_diff_k     util.py:48-58: per image, the order of the haplotype rows by Manhattan distance from the row whose
//                       summed distance to all rows is smallest.
//   reorder_rows_k      run_model.py:22-29 / run_context_encoder.py:101-112: image[order] for a whole batch.
//
// All of it is byte / integer / compare work bound by HBM: every input element is read once, every output element
// written once, with consecutive lanes on consecutive addresses.  Floating point appears only in the distance channel,
// which is formed in float64 with the reference's operation order ((2 * (d - min)) / (max - min) - 1) and rounded to
// float32 on the store, so results are bit-identical to the NumPy path.
#include "tile_common.h"

namespace sv {

constexpr int kPrepThreads = 256;
constexpr int kPrepMaxTile = 8192;   // n_haps * S floats of LDS for the window's haplotype sums
constexpr int kPrepMaxS = 1024;
constexpr int kSortMaxRows = 64;

struct PrepArgs {
    const float* x;            // [total_sites][n_haps][n_mut]
    const int64_t* pos;        // [total_sites]
    const int64_t* off;        // [R + 1] first site of each region
    float* out;                // [R][n_haps][S][C]
    double* dist;              // optional [total_sites]: the normalised distance vector of every region (util.py:34-45)
    int32_t* status;           // [R]: 0 ok, 1 = the summed genotype matrix's maximum is not 1 (util.py:198)
    int n_haps, n_mut, S, C;
    double norm_len;
};

__device__ __forceinline__ double block_reduce_minmax(double v, bool is_max, double* red) {
    // red: 4 doubles of LDS (one per wave)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double w = __shfl_xor(v, o, 64);
        v = is_max ? (w > v ? w : v) : (w < v ? w : v);
    }
    __syncthreads();  // red may still be read from a previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int k = 1; k < kPrepThreads / 64; ++k) r = is_max ? (red[k] > r ? red[k] : r) : (red[k] < r ? red[k] : r);
    return r;
}

__global__ __launch_bounds__(kPrepThreads) void process_regions_k(PrepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // hs[n_haps][S] | cnt[S] (int) | dwin[S] | red[4] (double)
    const int H = a.n_haps, S = a.S, C = a.C, M = a.n_mut;
    float* hs = sm;
    int* cnt = reinterpret_cast<int*>(sm + (size_t)H * S);
    float* dwin = sm + (size_t)H * S + S;
    double* red = reinterpret_cast<double*>(sm + (((size_t)H * S + 2 * S + 1) & ~(size_t)1));
    __shared__ float wave_max[kPrepThreads / 64];

    const int r = blockIdx.x;
    const int64_t base = a.off[r];
    const int n = (int)(a.off[r + 1] - base);
    const int64_t* pos = a.pos + base;
    const float* x = a.x + (size_t)base * H * M;

    // ---- inter-SNP distances: d[0] = 0, d[k] = (pos[k] - pos[k-1]) / norm_len; min and max over the whole region
    double dmin = INFINITY, dmax = -INFINITY;
    for (int k = threadIdx.x; k < n; k += kPrepThreads) {
        const double d = k == 0 ? 0.0 : (double)(pos[k] - pos[k - 1]) / a.norm_len;
        dmin = d < dmin ? d : dmin;
        dmax = d > dmax ? d : dmax;
    }
    dmin = block_reduce_minmax(dmin, false, red);
    dmax = block_reduce_minmax(dmax, true, red);
    const double span = dmax - dmin;
    if (a.dist != nullptr) {
        for (int k = threadIdx.x; k < n; k += kPrepThreads) {
            const double d = k == 0 ? 0.0 : (double)(pos[k] - pos[k - 1]) / a.norm_len;
            a.dist[base + k] = (2.0 * (d - dmin)) / span - 1.0;
        }
    }

    // ---- the window of S sites: centre crop when the region has at least S sites, centred zero padding otherwise
    const int mid = n / 2, half = S / 2;
    int src0, dst0, nwin;  // sites src0 .. src0 + nwin - 1 land in columns dst0 .. dst0 + nwin - 1
    if (mid >= half) {
        src0 = mid - half; dst0 = 0; nwin = 2 * half;
    } else {
        src0 = 0; dst0 = half - mid; nwin = n;
    }

    for (int i = threadIdx.x; i < H * S; i += kPrepThreads) hs[i] = 0.f;
    for (int col = threadIdx.x; col < S; col += kPrepThreads) {  // the window's distance channel, rounded once
        const int j = col - dst0;
        float v = 0.f;
        if (j >= 0 && j < nwin) {
            const int k = src0 + j;
            const double d = k == 0 ? 0.0 : (double)(pos[k] - pos[k - 1]) / a.norm_len;
            v = (float)((2.0 * (d - dmin)) / span - 1.0);
        }
        dwin[col] = v;
    }
    __syncthreads();

    // ---- haplotype sums over the mutation-type channels for EVERY site (the reference checks the maximum of the whole
    // matrix); the window's sums and per-site derived-allele counts stay in LDS
    float vmax = -INFINITY;
    const int64_t pairs = (int64_t)n * H;
    const FDiv fH = fd_make(H);
    const bool muldiv = (uint64_t)pairs * (uint64_t)H < (1ull << 32);  // range in which the multiply-based division is exact
    auto keep = [&](int64_t p, float v) {  // p = s * H + h
        vmax = v > vmax ? v : vmax;
        const int s = muldiv ? fd((int)p, fH) : (int)(p / H);
        const int h = (int)(p - (int64_t)s * H);
        const int j = s - src0;
        if (j >= 0 && j < nwin) hs[h * S + dst0 + j] = v;
    };
    if (M == 6 && (H & 1) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        // two (site, haplotype) pairs = 12 floats = three 16-byte loads per lane, consecutive lanes on consecutive
        // 48-byte pieces; two couples per iteration keep six loads in flight
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const int64_t couples = pairs >> 1;
        int64_t q = threadIdx.x;
        for (; q + kPrepThreads < couples; q += 2 * kPrepThreads) {
            const float4 a0 = x4[3 * q], a1 = x4[3 * q + 1], a2 = x4[3 * q + 2];
            const int64_t q2 = q + kPrepThreads;
            const float4 b0 = x4[3 * q2], b1 = x4[3 * q2 + 1], b2 = x4[3 * q2 + 2];
            keep(2 * q, ((((a0.x + a0.y) + a0.z) + a0.w) + a1.x) + a1.y);
            keep(2 * q + 1, ((((a1.z + a1.w) + a2.x) + a2.y) + a2.z) + a2.w);
            keep(2 * q2, ((((b0.x + b0.y) + b0.z) + b0.w) + b1.x) + b1.y);
            keep(2 * q2 + 1, ((((b1.z + b1.w) + b2.x) + b2.y) + b2.z) + b2.w);
        }
        for (; q < couples; q += kPrepThreads) {
            const float4 a0 = x4[3 * q], a1 = x4[3 * q + 1], a2 = x4[3 * q + 2];
            keep(2 * q, ((((a0.x + a0.y) + a0.z) + a0.w) + a1.x) + a1.y);
            keep(2 * q + 1, ((((a1.z + a1.w) + a2.x) + a2.y) + a2.z) + a2.w);
        }
    } else {
        for (int64_t p = threadIdx.x; p < pairs; p += kPrepThreads) {
            const float* q = x + (size_t)p * M;
            float v = q[0];
            for (int c = 1; c < M; ++c) v += q[c];
            keep(p, v);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(vmax, o, 64);
        vmax = w > vmax ? w : vmax;
    }
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = wave_max[0];
#pragma unroll
        for (int k = 1; k < kPrepThreads / 64; ++k) mx = wave_max[k] > mx ? wave_max[k] : mx;
        a.status[r] = (mx == 1.0f) ? 0 : 1;
    }
    // derived-allele count of each window column
    for (int col = threadIdx.x; col < S; col += kPrepThreads) {
        int k = 0;
        for (int h = 0; h < H; ++h) k += hs[h * S + col] != 0.f;
        cnt[col] = k;
    }
    __syncthreads();

    // ---- write the image: channel 0 = polarised haplotypes, channels 1 .. C-2 = 0, channel C-1 = distances
    float* out = a.out + (size_t)r * H * S * C;
    const int total = H * S * C;
    const FDiv fC = fd_make(C), fS = fd_make(S);  // total * max(C, S) stays far below 2^32 (n_haps * S <= 8192)
    for (int e = threadIdx.x; e < total; e += kPrepThreads) {
        const int px = fd(e, fC), c = e - px * C;
        const int h = fd(px, fS), col = px - h * S;
        float v = 0.f;
        if (c == C - 1) {
            v = dwin[col];
        } else if (c == 0) {
            // util.py:228-256: untouched when no haplotype carries the derived allele; flipped (1 - x) when more than
            // half do; zeros become -1 in both remaining cases.  Columns outside the window hold zeros and count 0.
            v = hs[h * S + col];
            const int k = cnt[col];
            if (k != 0) {
                if (2 * k > H) v = 1.f - v;
                if (v == 0.f) v = -1.f;
            }
        }
        out[e] = v;
    }
}

struct SortArgs {
    const float* img;   // [R][H][S][C]; channel `ch` is the haplotype matrix
    int64_t* order;     // [R][H]
    int H, S, C, ch;
};

__global__ __launch_bounds__(kPrepThreads) void sort_min_diff_k(SortArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // m[H][S+1] floats | D[H][H] doubles | tot[H] doubles
    const int H = a.H, S = a.S;
    const int pitch = S + 1;
    float* m = sm;
    double* D = reinterpret_cast<double*>(sm + (((size_t)H * pitch + 1) & ~(size_t)1));
    double* tot = D + (size_t)H * H;
    __shared__ int s_start;
    const float* img = a.img + (size_t)blockIdx.x * H * S * a.C;
    for (int i = threadIdx.x; i < H * S; i += kPrepThreads) {
        const int h = i / S, j = i - h * S;
        const float v = img[(size_t)i * a.C + a.ch];
        m[h * pitch + j] = v < 0.f ? 0.f : v;  // util.py:54
    }
    __syncthreads();
    for (int p = threadIdx.x; p < H * H; p += kPrepThreads) {
        const int u = p / H, v = p - u * H;
        double d = 0.0;
        for (int j = 0; j < S; ++j) d += fabs((double)m[u * pitch + j] - (double)m[v * pitch + j]);
        D[p] = d;
    }
    __syncthreads();
    if (threadIdx.x < H) {
        double t = 0.0;
        for (int v = 0; v < H; ++v) t += D[threadIdx.x * H + v];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        for (int u = 1; u < H; ++u)
            if (tot[u] < tot[best]) best = u;  // first minimum (np.argmin)
        s_start = best;
    }
    __syncthreads();
    if (threadIdx.x < H) {
        const double* row = D + (size_t)s_start * H;
        const double mine = row[threadIdx.x];
        int rank = 0;
        for (int v = 0; v < H; ++v) rank += (row[v] < mine) || (row[v] == mine && v < (int)threadIdx.x);
        a.order[(size_t)blockIdx.x * H + rank] = threadIdx.x;
    }
}

// out[r][k][:] = img[r][order[r][k]][:], rows of `row` elements of type T (float4 when the row allows it)
template <typename T>
__global__ void reorder_rows_k(const T* __restrict__ img, const int64_t* __restrict__ order, T* __restrict__ out,
                               int64_t R, int H, int row) {
    const int64_t total = R * H * row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t rk = i / row;
        const int e = (int)(i - rk * row);
        const int64_t r = rk / H;
        const int64_t src = order[rk];
        out[i] = img[(r * H + src) * row + e];
    }
}

}  // namespace sv

using namespace sv;

extern "C" {

int sv_process_regions(const float* x, const int64_t* positions, const int64_t* site_offsets, int64_t n_regions,
                       int n_haps, int n_mut, int n_snps, int n_channels, int64_t norm_len, float* out, double* distances,
                       int32_t* status, sv_stream_t stream) {
    SV_REQUIRE(x && positions && site_offsets && out && status, "sv_process_regions: null tensor");
    SV_REQUIRE(n_regions >= 0 && n_regions <= INT32_MAX, "sv_process_regions: bad region count");
    SV_REQUIRE(n_haps > 0 && n_mut > 0 && n_snps > 0 && n_snps <= kPrepMaxS && (n_snps & 1) == 0 && n_channels >= 2,
               "sv_process_regions: bad shape (n_haps=%d n_mut=%d n_snps=%d n_channels=%d; n_snps must be even)", n_haps,
               n_mut, n_snps, n_channels);
    SV_REQUIRE((int64_t)n_haps * n_snps <= kPrepMaxTile, "sv_process_regions: n_haps * n_snps = %lld exceeds %d",
               (long long)n_haps * n_snps, kPrepMaxTile);
    SV_REQUIRE(norm_len != 0, "sv_process_regions: norm_len is 0");
    if (n_regions == 0) return SV_OK;
    PrepArgs a{x, positions, site_offsets, out, distances, status, n_haps, n_mut, n_snps, n_channels, (double)norm_len};
    const size_t lds = ((((size_t)n_haps * n_snps + 2 * n_snps + 1) & ~(size_t)1)) * sizeof(float) + 4 * sizeof(double);
    hipLaunchKernelGGL(process_regions_k, dim3((unsigned)n_regions), dim3(kPrepThreads), lds, to_stream(stream), a);
    return check_launch("sv_process_regions");
}

int sv_sort_min_diff(const float* images, int64_t n_images, int n_haps, int n_snps, int n_channels, int channel,
                     int64_t* order, sv_stream_t stream) {
    SV_REQUIRE(images && order, "sv_sort_min_diff: null tensor");
    SV_REQUIRE(n_images >= 0 && n_images <= INT32_MAX, "sv_sort_min_diff: bad image count");
    SV_REQUIRE(n_haps > 0 && n_haps <= kSortMaxRows && n_snps > 0 && n_snps <= kPrepMaxS && n_channels > 0 && channel >= 0 &&
                   channel < n_channels,
               "sv_sort_min_diff: bad shape (n_haps=%d (max %d) n_snps=%d n_channels=%d channel=%d)", n_haps, kSortMaxRows,
               n_snps, n_channels, channel);
    if (n_images == 0) return SV_OK;
    const size_t mfl = (((size_t)n_haps * (n_snps + 1) + 1) & ~(size_t)1);
    const size_t lds = mfl * sizeof(float) + ((size_t)n_haps * n_haps + n_haps) * sizeof(double);
    SV_REQUIRE(lds <= 64 * 1024, "sv_sort_min_diff: %zu bytes of LDS needed, 65536 available", lds);
    SortArgs a{images, order, n_haps, n_snps, n_channels, channel};
    hipLaunchKernelGGL(sort_min_diff_k, dim3((unsigned)n_images), dim3(kPrepThreads), lds, to_stream(stream), a);
    return check_launch("sv_sort_min_diff");
}

int sv_reorder_rows(const float* images, const int64_t* order, float* out, int64_t n_images, int n_rows, int row_elems,
                    sv_stream_t stream) {
    SV_REQUIRE(images && order && out && images != out, "sv_reorder_rows: null tensor or in-place call");
    SV_REQUIRE(n_images >= 0 && n_rows > 0 && row_elems > 0, "sv_reorder_rows: bad shape");
    if (n_images == 0) return SV_OK;
    if ((row_elems & 3) == 0 && ((reinterpret_cast<uintptr_t>(images) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
        hipLaunchKernelGGL(reorder_rows_k<float4>, dim3(ew_grid(n_images * n_rows * (row_elems / 4), 256, 16384)), dim3(256),
                           0, to_stream(stream), reinterpret_cast<const float4*>(images), order,
                           reinterpret_cast<float4*>(out), n_images, n_rows, row_elems / 4);
    else
        hipLaunchKernelGGL(reorder_rows_k<float>, dim3(ew_grid(n_images * n_rows * row_elems, 256, 16384)), dim3(256), 0,
                           to_stream(stream), images, order, out, n_images, n_rows, row_elems);
    return check_launch("sv_reorder_rows");
}

}  // extern "C"
