// Implicit-GEMM convolution on the matrix cores (f32 MFMA 16x16x4) for the GEMM-shaped geometries that the LDS-tiled
// family does not cover: strided 4x4 / 2x2 convs and transposed convs with 16..1000 channels (U-Net, context encoder,
// discriminators), the 4x4 VALID bottleneck pair (a 2048 x 1000 GEMM), any kernel size / stride / padding.
//
//   forward        Y[(n,oh,ow)][co]   = sum_k  X(n, oh*SH-pt+kh, ow*SW-pl+kw, ci) * W[(kh,kw,ci)][co]
//   input gradient dX[(n,ih,iw)][ci]  = sum_k  dZ(n, (ih+pt-kh)/SH, (iw+pl-kw)/SW, co) * W[kh][kw][ci][co],  k = (kh,kw,co)
//   kernel gradient dW[(kh,kw,ci)][co] = sum_r X(...) * dZ[r][co],  r = (n,oh,ow), split over workgroups (slabs + fixed-order sum)
//
// Nothing is materialised: a workgroup (4 waves) owns a 64 x 64 tile of the result and walks the reduction in steps of
// 16; the A / B tiles are gathered straight from the NHWC tensors into LDS (k-major, so the MFMA operand reads are
// conflict-free), each wave holds a 32 x 32 sub-tile as 2 x 2 MFMA accumulators.
// Lane l of an MFMA feeds A[row l % 16][k l / 16], B[k l / 16][col l % 16] and receives D[4 * (l / 16) + r][l % 16].
#include <type_traits>

#include "tile_common.h"

namespace sv {

typedef float v4f_t __attribute__((ext_vector_type(4)));

enum { GEMM_FWD = 0, GEMM_DGRAD = 1, GEMM_WGRAD = 2 };

struct ConvGemmArgs {
    sv_conv_geom g;
    const float* a;     // FWD/WGRAD: x;  DGRAD: dz
    const float* b;     // FWD/DGRAD: w;  WGRAD: dz
    const float* bias;  // FWD / DGRAD epilogue (may be null)
    float* c;           // FWD: y;  DGRAD: dx;  WGRAD: slabs [split][M][N]
    int M, N, K;        // GEMM extents (K: reduction)
    int k_per_split;    // WGRAD: reduction elements per blockIdx.z
    int act;
    float alpha;
    FDiv f_wo, f_howo;  // FWD/WGRAD rows r -> (n, oh, ow);  DGRAD rows -> (n, ih, iw) (then these divide by W, H*W)
    FDiv f_ck, f_kw;    // reduction / row index (tap, channel): divide by the channel count, then by KW
    int one_out;        // DGRAD: the conv has a single output position (Ho == Wo == 1): one row class per input pixel
    int splits;         // FWD / DGRAD: > 1 cuts the reduction into `splits` ranges over blockIdx.z; raw partial sums go to
    float* slabs;       //              slabs[split][M][N] (indexed like c) and splitk_finish_k applies bias + activation
    int accumulate;     // WGRAD with a single split: c is dw itself and the tile is added to it when set
    unsigned a_bytes, b_bytes;  // sizes of the A / B tensors: the gathers are range-checked buffer loads
    const float* amask;  // FWD / WGRAD (MK kernels): x is read as x * (1 - amask), amask laid out like x
};

typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
constexpr int BM = 64, BN = 64, BK = 16, LDT = 68;  // LDS row pitch (floats): 64 + 4

// BNT: columns of the result tile.  64: the 4 waves sit 2 x 2, each on 32 x 32 (2 x 2 accumulators);  16 (results with at
// most 16 columns: the 3-channel image convs, single-logit heads): the waves sit 4 x 1, each on 16 x 16.
// VA / VB: the A / B operand is gathered with 16-byte loads along its memory-contiguous index (channels); needs that
// channel count to be a multiple of 4 and a 16-byte aligned tensor, else the operand falls back to 4-byte gathers.
// MK (FWD / WGRAD): the input is masked on the way in, x * (1 - mask) (run_context_encoder.py:242): the mask element rides
// the register ring beside the x element it belongs to and the product is formed when the pair is written to LDS, so
// the masked tensor is never materialised.
template <int MODE, int BNT, bool VA, bool VB, bool MK = false>
__global__ __launch_bounds__(256) void conv_gemm_mfma(ConvGemmArgs p) {
    constexpr int WAVES_N = BNT == 64 ? 2 : 1, WAVES_M = 4 / WAVES_N;
    constexpr int TI = BM / WAVES_M / 16, TJ = BNT / WAVES_N / 16;  // accumulators per wave
    constexpr int NB = BK * BNT / 256;                               // B elements per thread and step (4-byte gathers)
    constexpr int BROWS = 256 / BNT;                                 // reduction rows one sweep of the threads covers (k-major B)
    constexpr int NR = VA ? 1 : 4;                                   // result rows a thread gathers A for (FWD / DGRAD)
    constexpr int NBR = VB ? 4 : NB;                                 // B registers per thread
    __shared__ __attribute__((aligned(16))) float As[BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK][LDT];
    const sv_conv_geom& g = p.g;
    // Gathers go through buffer descriptors: a dead element (padding, tile edge, end of the reduction) gets an offset
    // beyond the tensor and comes back as zero, so no load sits behind a branch.
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, p.b_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MK ? p.amask : p.a), 0, p.a_bytes, 0x00020000);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BNT;
    const int wm = (wave % WAVES_M) * (16 * TI), wn = (wave / WAVES_M) * (16 * TJ);

    int kbeg = 0, kend = p.K;
    if (MODE == GEMM_WGRAD) {
        kbeg = blockIdx.z * p.k_per_split;
        kend = min(p.K, kbeg + p.k_per_split);
    }

    // ---- DGRAD: rows are grouped into classes (blockIdx.z) that share their set of contributing filter taps, so the
    // reduction only walks taps that can hit: with stride s only every s-th tap reaches a given input row / column
    // (class = (ih % SH, iw % SW)); a conv with a single output position has exactly one tap per input pixel
    // (class = the pixel, rows = images).
    __shared__ int row_img[BM], row_h[BM], row_w[BM];
    int kh0 = 0, kw0 = 0, nkw = 1;
    FDiv f_nkw = fd_make(1);
    const int split = p.splits > 1 ? blockIdx.z % p.splits : 0;
    if (MODE == GEMM_DGRAD) {
        int Mc;
        const int z = p.splits > 1 ? blockIdx.z / p.splits : blockIdx.z;
        if (p.one_out) {
            const int ih = z / g.W, iw = z - ih * g.W;
            kh0 = ih + g.pad_t; kw0 = iw + g.pad_l;
            Mc = g.N;
            kend = (kh0 < g.KH && kw0 < g.KW) ? g.Cout : 0;
            if (t < BM) {
                const int m = m0 + t;
                row_img[t] = m < Mc ? m : -1; row_h[t] = ih; row_w[t] = iw;
            }
        } else {
            const int ph = z / g.SW, pw = z - ph * g.SW;
            const int Ha = (g.H - ph + g.SH - 1) / g.SH, Wa = (g.W - pw + g.SW - 1) / g.SW;
            Mc = g.N * Ha * Wa;
            kh0 = (ph + g.pad_t) % g.SH; kw0 = (pw + g.pad_l) % g.SW;
            const int nkh = kh0 < g.KH ? (g.KH - kh0 + g.SH - 1) / g.SH : 0;
            nkw = kw0 < g.KW ? (g.KW - kw0 + g.SW - 1) / g.SW : 0;
            kend = nkh * nkw * g.Cout;
            if (nkw > 0) f_nkw = fd_make(nkw);
            if (t < BM) {
                const int m = m0 + t;
                int img = -1, hh = 0, ww = 0;
                if (m < Mc) {
                    img = m / (Ha * Wa);
                    const int rem = m - img * (Ha * Wa);
                    const int a = rem / Wa;
                    hh = ph + a * g.SH; ww = pw + (rem - a * Wa) * g.SW;
                }
                row_img[t] = img; row_h[t] = hh; row_w[t] = ww;
            }
        }
        if (m0 >= Mc) return;  // whole workgroup: this class has fewer rows
        __syncthreads();
    }
    if (MODE != GEMM_WGRAD && p.splits > 1) {
        const int kper = ((kend + p.splits - 1) / p.splits + BK - 1) / BK * BK;
        kbeg = split * kper;
        kend = min(kend, kbeg + kper);  // may leave an empty range: the workgroup then stores zeros
    }

    // ---- per-thread bookkeeping of the A gather (fixed over the reduction loop)
    // 4-byte gathers: FWD / DGRAD thread loads k = t & 15 of rows (t >> 4) + 16 i;  WGRAD: row m = t & 63, reduction (t >> 6) + 4 i
    // 16-byte gathers: FWD / DGRAD thread loads k = 4 (t & 3) .. + 3 of row t >> 2;  WGRAD: rows m = 4 (t & 15) .. + 3, reduction t >> 4
    // r_base: element offset of the row's tap-(0, 0), channel-0 operand element (possibly outside the tensor: only
    // offsets whose bounds test passes are used); a reduction step adds one offset shared by all rows of the thread
    int r_h[NR], r_w[NR], r_base[NR];
    bool r_ok[NR];
    int a_kh = 0, a_kw = 0, a_off = 0;
    bool a_ok = true;
    if (MODE == GEMM_WGRAD) {
        const int m = m0 + (VA ? (t & 15) * 4 : (t & 63));
        a_ok = m < p.M;
        const int tap = a_ok ? fd(m, p.f_ck) : 0;
        const int a_ci = a_ok ? m - tap * g.Cin : 0;
        a_kh = fd(tap, p.f_kw);
        a_kw = tap - a_kh * g.KW;
        a_off = (a_kh * g.W + a_kw) * g.Cin + a_ci;
    } else if (MODE == GEMM_DGRAD) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            // a row of a class reaches output pixel (r_h - jh, r_w - jw) through the class's tap (jh, jw): the divisions by the
            // strides are exact by construction of the class and done once here, not once per reduction step
            const int rl = VA ? (t >> 2) : (t >> 4) + 16 * i;
            const int img = row_img[rl];
            r_ok[i] = img >= 0;
            r_h[i] = p.one_out ? 0 : (row_h[rl] + g.pad_t - kh0) / g.SH;
            r_w[i] = p.one_out ? 0 : (row_w[rl] + g.pad_l - kw0) / g.SW;
            r_base[i] = ((img * g.Ho + r_h[i]) * g.Wo + r_w[i]) * g.Cout;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int m = m0 + (VA ? (t >> 2) : (t >> 4) + 16 * i);
            r_ok[i] = m < p.M;
            const int mm = r_ok[i] ? m : 0;
            const int img = fd(mm, p.f_howo);
            const int rem = mm - img * p.f_howo.d;
            const int hh = fd(rem, p.f_wo);
            r_h[i] = hh * g.SH - g.pad_t; r_w[i] = (rem - hh * p.f_wo.d) * g.SW - g.pad_l;
            r_base[i] = ((img * g.H + r_h[i]) * g.W + r_w[i]) * g.Cin;
        }
    }

    v4f_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = v4f_t{0.f, 0.f, 0.f, 0.f};

    // 16 / 4 bytes at element offset `e` of a tensor, zeros when !ok
    auto ld4 = [](__amdgpu_buffer_rsrc_t rs, int e, bool ok, float* dst) {
        const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? e * 4 : (int)0x80000000, 0, 0);
        // the whole vector at once: element-wise bit casts of v.x .. v.w make hipcc (ROCm 7.2) load one dword only
        const v4f_t f = __builtin_bit_cast(v4f_t, v);
        dst[0] = f[0]; dst[1] = f[1]; dst[2] = f[2]; dst[3] = f[3];
    };
    auto ld1 = [](__amdgpu_buffer_rsrc_t rs, int e, bool ok) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? e * 4 : (int)0x80000000, 0, 0));
    };
    const bool b_lane = BNT == 64 || t < 64;  // 16-byte B gathers of a 16-column tile need 64 threads only

    // global -> registers for one reduction step (issued ahead of its use, so the round trip overlaps the MFMAs)
    auto gather = [&](int k0, float (&av)[4], float (&bv)[NBR], float (&mv)[4]) {
        // ---- the reduction index this thread decodes (FWD / DGRAD)
        const int kt = k0 + (VA ? (t & 3) * 4 : (t & 15));
        // ---- gather A
        if (MODE == GEMM_FWD) {
            const bool kok = kt < kend;
            const int kk = kok ? kt : 0;
            const int tap = fd(kk, p.f_ck), ci = kk - tap * g.Cin;
            const int kh = fd(tap, p.f_kw), kw = tap - kh * g.KW;
            const int koff = (kh * g.W + kw) * g.Cin + ci;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const bool ok = kok && r_ok[i] && (unsigned)(r_h[i] + kh) < (unsigned)g.H && (unsigned)(r_w[i] + kw) < (unsigned)g.W;
                if (VA) ld4(rs_a, r_base[i] + koff, ok, av);
                else av[i] = ld1(rs_a, r_base[i] + koff, ok);
                if (MK) {
                    if (VA) ld4(rs_m, r_base[i] + koff, ok, mv);
                    else mv[i] = ld1(rs_m, r_base[i] + koff, ok);
                }
            }
        } else if (MODE == GEMM_DGRAD) {
            const bool kok = kt < kend;
            const int kk = kok ? kt : 0;
            const int j = fd(kk, p.f_ck), co = kk - j * g.Cout;  // j: index among the class's taps
            const int jh = p.one_out ? 0 : fd(j, f_nkw), jw = p.one_out ? 0 : j - jh * nkw;
            const int koff = co - (jh * g.Wo + jw) * g.Cout;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const bool ok = kok && r_ok[i] && (unsigned)(r_h[i] - jh) < (unsigned)g.Ho && (unsigned)(r_w[i] - jw) < (unsigned)g.Wo;
                if (VA) ld4(rs_a, r_base[i] + koff, ok, av);
                else av[i] = ld1(rs_a, r_base[i] + koff, ok);
            }
        } else {
#pragma unroll
            for (int i = 0; i < (VA ? 1 : 4); ++i) {
                const int r = k0 + (VA ? (t >> 4) : (t >> 6) + 4 * i);
                const bool rok = a_ok && r < kend;
                const int rr = rok ? r : 0;
                const int img = fd(rr, p.f_howo);
                const int rem = rr - img * p.f_howo.d;
                const int oh = fd(rem, p.f_wo), ow = rem - oh * p.f_wo.d;
                const int h0 = oh * g.SH - g.pad_t, w0 = ow * g.SW - g.pad_l;
                const bool ok = rok && (unsigned)(h0 + a_kh) < (unsigned)g.H && (unsigned)(w0 + a_kw) < (unsigned)g.W;
                const int e = ((img * g.H + h0) * g.W + w0) * g.Cin + a_off;
                if (VA) ld4(rs_a, e, ok, av);
                else av[i] = ld1(rs_a, e, ok);
                if (MK) {
                    if (VA) ld4(rs_m, e, ok, mv);
                    else mv[i] = ld1(rs_m, e, ok);
                }
            }
        }
        // ---- gather B
        if (MODE == GEMM_DGRAD) {
            const int kb = k0 + (VB ? (t & 3) * 4 : (t & 15));
            const bool kok = kb < kend;
            const int kk = kok ? kb : 0;
            const int j = fd(kk, p.f_ck), co = kk - j * g.Cout;
            const int jh = fd(j, f_nkw), jw = j - jh * nkw;
            const int kh = p.one_out ? kh0 : kh0 + jh * g.SH, kw = p.one_out ? kw0 : kw0 + jw * g.SW;
            const int boff = (kh * g.KW + kw) * g.Cin * g.Cout + co;
            if (VB) {
                const int n = n0 + (t >> 2);
                ld4(rs_b, boff + n * g.Cout, b_lane && kok && n < p.N, bv);
            } else {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int n = n0 + (t >> 4) + 16 * i;
                    bv[i] = ld1(rs_b, boff + n * g.Cout, kok && n < p.N);
                }
            }
        } else if (VB) {
            const int kr = BNT == 64 ? (t >> 4) : (t >> 2);
            const int n = n0 + (BNT == 64 ? (t & 15) : (t & 3)) * 4;
            ld4(rs_b, k0 * p.N + (kr * p.N + n), b_lane && k0 + kr < kend && n < p.N, bv);
        } else {
            const int n = n0 + (t & (BNT - 1));
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int kr = (t / BNT) + BROWS * i;
                bv[i] = ld1(rs_b, k0 * p.N + (kr * p.N + n), k0 + kr < kend && n < p.N);
            }
        }
    };

    // Register ring of PF reduction steps: the gather of step s + PF - 1 is issued while step s is multiplied, so a load
    // has PF - 1 multiply phases to arrive (one phase is 16 MFMAs per wave, shorter than a trip to L2 / HBM; the small
    // launches of the GAN drivers' batch sizes leave a SIMD one or two waves and nothing else to cover it).
    constexpr int PF = 3;
    float av[PF][4], bv[PF][NBR], mv[PF][4];
#pragma unroll
    for (int d = 0; d < PF - 1; ++d) gather(kbeg + d * BK, av[d], bv[d], mv[d]);  // beyond kend: all pieces dead, zeros
    // one reduction step out of ring slot D; GATHER: also issue the gather PF - 1 steps ahead (into the slot this step's
    // predecessor just freed).  Gathers are issued whether or not that step exists (beyond kend every piece is dead): a
    // fixed number of loads in flight lets the compiler wait for the oldest slot only instead of draining the ring.
    auto step = [&](auto slot, int k0, bool do_gather) {
        constexpr int d = decltype(slot)::value;
        __syncthreads();  // the previous step's MFMA reads are done
        if (MK) {
#pragma unroll
            for (int i = 0; i < 4; ++i) av[d][i] *= 1.f - mv[d][i];
        }
        if (MODE == GEMM_WGRAD) {
            if (VA) {
                *reinterpret_cast<float4*>(&As[t >> 4][(t & 15) * 4]) = make_float4(av[d][0], av[d][1], av[d][2], av[d][3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) As[(t >> 6) + 4 * i][t & 63] = av[d][i];
            }
        } else if (VA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) As[(t & 3) * 4 + i][t >> 2] = av[d][i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) As[t & 15][(t >> 4) + 16 * i] = av[d][i];
        }
        if constexpr (MODE == GEMM_DGRAD) {
            if constexpr (VB) {
                if (b_lane) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) Bs[(t & 3) * 4 + i][t >> 2] = bv[d][i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < NB; ++i) Bs[t & 15][(t >> 4) + 16 * i] = bv[d][i];
            }
        } else if constexpr (VB) {
            if (b_lane)
                *reinterpret_cast<float4*>(&Bs[BNT == 64 ? (t >> 4) : (t >> 2)][(BNT == 64 ? (t & 15) : (t & 3)) * 4]) =
                    make_float4(bv[d][0], bv[d][1], bv[d][2], bv[d][3]);
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) Bs[(t / BNT) + BROWS * i][t & (BNT - 1)] = bv[d][i];
        }
        __syncthreads();
        if (do_gather) gather(k0 + (PF - 1) * BK, av[(d + PF - 1) % PF], bv[(d + PF - 1) % PF], mv[(d + PF - 1) % PF]);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            float a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = As[kk + lk][wm + 16 * i + li];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = Bs[kk + lk][wn + 16 * j + li];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    const int nsteps = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    int k0 = kbeg;
    for (int it = nsteps / PF; it > 0; --it) {  // whole turns of the ring, no exits inside
        step(std::integral_constant<int, 0>{}, k0, true);
        step(std::integral_constant<int, 1>{}, k0 + BK, true);
        step(std::integral_constant<int, 2>{}, k0 + 2 * BK, true);
        k0 += PF * BK;
    }
    static_assert(PF == 3, "the turn above and the tail below are written out for a ring of three");
    const int rem = nsteps % PF;  // their gathers are in flight already
    if (rem >= 1) step(std::integral_constant<int, 0>{}, k0, false);
    if (rem >= 2) step(std::integral_constant<int, 1>{}, k0 + BK, false);
    // ---- epilogue
    float* out = p.c;
    if (MODE == GEMM_WGRAD) out += (size_t)blockIdx.z * p.M * p.N;
    const bool raw = MODE == GEMM_WGRAD || p.splits > 1;
    if (MODE != GEMM_WGRAD && p.splits > 1) out = p.slabs + (size_t)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int n = n0 + wn + 16 * j + li;
            if (n >= p.N) continue;
            const float bb = (!raw && p.bias != nullptr) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = wm + 16 * i + 4 * lk + r;
                const int m = m0 + rl;
                float v = acc[i][j][r];
                if (!raw) v = act_apply(v + bb, p.act, p.alpha);
                if (MODE == GEMM_DGRAD) {
                    if (row_img[rl] < 0) continue;
                    out[(((size_t)row_img[rl] * g.H + row_h[rl]) * g.W + row_w[rl]) * g.Cin + n] = v;
                } else {
                    if (m >= p.M) continue;
                    if (MODE == GEMM_WGRAD && p.accumulate) v += out[(size_t)m * p.N + n];
                    out[(size_t)m * p.N + n] = v;
                }
            }
        }
}

// out[i] = act(sum_s slabs[s][i] + bias[i % N]), splits summed in index order (bit-reproducible)
__global__ __launch_bounds__(256) void splitk_finish_k(const float* __restrict__ slabs, int splits, int64_t MN, int N,
                                                       const float* __restrict__ bias, int act, float alpha,
                                                       float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    float s = 0.f;
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
        const float v0 = slabs[(int64_t)k * MN + i], v1 = slabs[(int64_t)(k + 1) * MN + i];
        const float v2 = slabs[(int64_t)(k + 2) * MN + i], v3 = slabs[(int64_t)(k + 3) * MN + i];
        s = (((s + v0) + v1) + v2) + v3;
    }
    for (; k < splits; ++k) s += slabs[(int64_t)k * MN + i];
    if (bias != nullptr) s += bias[(int)(i % N)];
    out[i] = act_apply(s, act, alpha);
}

// ------------------------------------------------------------------------------------------------------ host side
// Index ranges the implicit-GEMM kernels can decode, and a reduction of at least min_k terms.
bool conv_gemm_shape_ok(const sv_conv_geom& g, int mode, int min_k) {
    if (g.up_h != 1 || g.up_w != 1) return false;
    const int ncols = mode == GEMM_DGRAD ? g.Cin : g.Cout;
    const int64_t kdim = mode == GEMM_DGRAD ? (int64_t)g.KH * g.KW * g.Cout
                                            : (mode == GEMM_FWD ? (int64_t)g.KH * g.KW * g.Cin : (int64_t)g.N * g.Ho * g.Wo);
    const int64_t rows = mode == GEMM_FWD ? (int64_t)g.N * g.Ho * g.Wo
                                          : (mode == GEMM_DGRAD ? (int64_t)g.N * g.H * g.W : (int64_t)g.KH * g.KW * g.Cin);
    // index decoding uses 32-bit multiply-shift division: keep index * divisor below 2^32
    const int64_t sp = mode == GEMM_DGRAD ? (int64_t)g.H * g.W : (int64_t)g.Ho * g.Wo;
    const int64_t rdec = mode == GEMM_WGRAD ? kdim : rows, cdec = mode == GEMM_WGRAD ? rows : kdim;
    if (rdec * sp >= (1ll << 32) || cdec * (g.Cin > g.Cout ? g.Cin : g.Cout) >= (1ll << 32)) return false;
    if ((int64_t)g.SH * g.SW > 65535 || (int64_t)g.H * g.W > 65535) return false;  // row classes ride on grid.z
    // the gathers address the tensors with 32-bit byte offsets, 2^31 being the "no such element" offset
    const int64_t lim = 1ll << 29;
    if ((int64_t)g.N * g.H * g.W * g.Cin >= lim || (int64_t)g.N * g.Ho * g.Wo * g.Cout >= lim ||
        (int64_t)g.KH * g.KW * g.Cin * g.Cout >= lim) return false;
    return ncols >= 1 && kdim >= min_k && rows < (1ll << 30) && kdim < (1ll << 30);
}

// The f32 GEMM path pays off from a reduction of 16 terms (one step of the kernel) on.
bool conv_gemm_worthwhile(const sv_conv_geom& g, int mode) { return conv_gemm_shape_ok(g, mode, 16); }

static int tile_cols(int ncols) { return ncols <= 16 ? 16 : BN; }

static bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int MODE>
static void launch_variant(const ConvGemmArgs& p, dim3 grid, int bnt, bool va, bool vb, hipStream_t st) {
#define SV_GEMM_CASE(B, A_, B_)                                                                                          \
    if (bnt == B && va == A_ && vb == B_) {                                                                              \
        if constexpr (MODE != GEMM_DGRAD) {                                                                              \
            if (p.amask != nullptr) {                                                                                    \
                hipLaunchKernelGGL((conv_gemm_mfma<MODE, B, A_, B_, true>), grid, dim3(256), 0, st, p);                  \
                return;                                                                                                  \
            }                                                                                                            \
        }                                                                                                                \
        hipLaunchKernelGGL((conv_gemm_mfma<MODE, B, A_, B_>), grid, dim3(256), 0, st, p);                                \
        return;                                                                                                          \
    }
    SV_GEMM_CASE(64, true, true) SV_GEMM_CASE(64, true, false) SV_GEMM_CASE(64, false, true) SV_GEMM_CASE(64, false, false)
    SV_GEMM_CASE(16, true, true) SV_GEMM_CASE(16, true, false) SV_GEMM_CASE(16, false, true) SV_GEMM_CASE(16, false, false)
#undef SV_GEMM_CASE
}

// FWD / DGRAD: a result with few 64 x 64 tiles leaves most of the 256 CUs idle while each workgroup walks a long
// reduction; cut the reduction so that about 512 workgroups exist, each keeping at least 4 steps.
static void dgrad_classes(const sv_conv_geom& g, int& classes, int& rows, int64_t& kclass) {
    if (g.Ho == 1 && g.Wo == 1) {
        classes = g.H * g.W; rows = g.N; kclass = g.Cout;
    } else {
        classes = g.SH * g.SW;
        rows = g.N * (int)ceil_div(g.H, g.SH) * (int)ceil_div(g.W, g.SW);
        kclass = (int64_t)ceil_div(g.KH, g.SH) * ceil_div(g.KW, g.SW) * g.Cout;
    }
}

int conv_gemm_splits(const sv_conv_geom& g, int mode) {
    int64_t blocks, kdim;
    int zbase = 1;
    if (mode == GEMM_FWD) {
        blocks = ceil_div((int64_t)g.N * g.Ho * g.Wo, BM) * ceil_div(g.Cout, tile_cols(g.Cout));
        kdim = (int64_t)g.KH * g.KW * g.Cin;
    } else {
        int classes, rows;
        dgrad_classes(g, classes, rows, kdim);
        blocks = ceil_div(rows, BM) * ceil_div(g.Cin, tile_cols(g.Cin)) * classes;
        zbase = classes;
    }
    // 192 .. 511 workgroups (about one per CU, one wave per SIMD): each walks its reduction at the latency of a lone
    // global -> LDS -> MFMA chain, and from 64 steps on a cut into about 1024 workgroups wins over the extra finish pass
    if (blocks >= 512 || (blocks >= 192 && kdim < 64 * BK)) return 1;
    int64_t s = ceil_div(blocks >= 192 ? 1024 : 512, blocks);
    const int64_t maxs = kdim / (4 * BK);
    if (s > maxs) s = maxs;
    if (s > 256) s = 256;
    if (s * zbase > 65535) s = 65535 / zbase;
    return s < 2 ? 1 : (int)s;
}

size_t conv_gemm_splitk_workspace(const sv_conv_geom& g, int mode) {
    if (mode == GEMM_WGRAD || !conv_gemm_worthwhile(g, mode)) return 0;
    const int s = conv_gemm_splits(g, mode);
    if (s < 2) return 0;
    const size_t mn = mode == GEMM_FWD ? (size_t)g.N * g.Ho * g.Wo * g.Cout : (size_t)g.N * g.H * g.W * g.Cin;
    return (size_t)s * mn * sizeof(float);
}

static int launch_splitk_finish(const ConvGemmArgs& p, hipStream_t st) {
    const int64_t mn = (int64_t)p.M * p.N;
    hipLaunchKernelGGL(splitk_finish_k, dim3((unsigned)ceil_div(mn, 256)), dim3(256), 0, st, p.slabs, p.splits, mn, p.N,
                       p.bias, p.act, p.alpha, p.c);
    return check_launch("splitk_finish_k");
}

int conv_gemm_wgrad_splits(const sv_conv_geom& g) {
    const int64_t M = (int64_t)g.KH * g.KW * g.Cin, R = (int64_t)g.N * g.Ho * g.Wo;
    const int64_t tiles = ceil_div(M, BM) * ceil_div(g.Cout, tile_cols(g.Cout));
    int64_t s = ceil_div(512, tiles);
    const int64_t maxs = ceil_div(R, 4 * BK);
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

size_t conv_gemm_bf16_wgrad_workspace(const sv_conv_geom& g);  // conv_gemm_bf16.hip

// One query serves both storage types (the entry point does not know the dtype): the larger of the two kernels' needs.
size_t conv_gemm_wgrad_workspace(const sv_conv_geom& g) {
    const size_t f32 = conv_gemm_worthwhile(g, GEMM_WGRAD)
                           ? (size_t)conv_gemm_wgrad_splits(g) * g.KH * g.KW * g.Cin * g.Cout * sizeof(float) : 0;
    const size_t b16 = conv_gemm_bf16_wgrad_workspace(g);
    return f32 > b16 ? f32 : b16;
}

static void fill_common(ConvGemmArgs& p, const sv_conv_geom& g, int mode) {
    p.g = g;
    if (mode == GEMM_DGRAD) {
        p.f_wo = fd_make(g.W); p.f_howo = fd_make(g.H * g.W); p.f_ck = fd_make(g.Cout);
    } else {
        p.f_wo = fd_make(g.Wo); p.f_howo = fd_make(g.Ho * g.Wo); p.f_ck = fd_make(g.Cin);
    }
    p.f_kw = fd_make(g.KW);
    const unsigned bx = (unsigned)((int64_t)g.N * g.H * g.W * g.Cin * 4), bz = (unsigned)((int64_t)g.N * g.Ho * g.Wo * g.Cout * 4);
    const unsigned bw = (unsigned)((int64_t)g.KH * g.KW * g.Cin * g.Cout * 4);
    p.a_bytes = mode == GEMM_DGRAD ? bz : bx;
    p.b_bytes = mode == GEMM_WGRAD ? bz : bw;
}

// ws / ws_bytes: optional scratch of conv_gemm_splitk_workspace(g, mode) bytes; without it the reduction is not cut
int launch_conv_gemm_fwd(const sv_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act,
                         float alpha, void* ws, size_t ws_bytes, hipStream_t st, const float* xmask) {
    ConvGemmArgs p{};
    fill_common(p, g, GEMM_FWD);
    p.a = x; p.b = w; p.bias = bias; p.c = y; p.act = act; p.alpha = alpha; p.amask = xmask;
    p.M = g.N * g.Ho * g.Wo; p.N = g.Cout; p.K = g.KH * g.KW * g.Cin;
    p.splits = 1;
    const size_t need = conv_gemm_splitk_workspace(g, GEMM_FWD);
    if (need > 0 && ws != nullptr && ws_bytes >= need) {
        p.splits = conv_gemm_splits(g, GEMM_FWD);
        p.slabs = (float*)ws;
    }
    const int bnt = tile_cols(p.N);
    dim3 grid((unsigned)ceil_div(p.M, BM), (unsigned)ceil_div(p.N, bnt), (unsigned)p.splits);
    launch_variant<GEMM_FWD>(p, grid, bnt, (g.Cin & 3) == 0 && aligned16(x) && aligned16(xmask), (g.Cout & 3) == 0 && aligned16(w), st);
    int rc = check_launch("conv_gemm_mfma<fwd>");
    if (rc || p.splits == 1) return rc;
    return launch_splitk_finish(p, st);
}

int launch_conv_gemm_dgrad(const sv_conv_geom& g, const float* dz, const float* w, const float* bias, float* dx, int act,
                           float alpha, void* ws, size_t ws_bytes, hipStream_t st) {
    ConvGemmArgs p{};
    fill_common(p, g, GEMM_DGRAD);
    p.a = dz; p.b = w; p.bias = bias; p.c = dx; p.act = act; p.alpha = alpha;
    p.M = g.N * g.H * g.W; p.N = g.Cin; p.K = g.KH * g.KW * g.Cout;
    p.one_out = g.Ho == 1 && g.Wo == 1;
    int classes, rows;  // row classes and the largest class's row count
    int64_t kclass;
    dgrad_classes(g, classes, rows, kclass);
    p.splits = 1;
    const size_t need = conv_gemm_splitk_workspace(g, GEMM_DGRAD);
    if (need > 0 && ws != nullptr && ws_bytes >= need) {
        p.splits = conv_gemm_splits(g, GEMM_DGRAD);
        p.slabs = (float*)ws;
    }
    const int bnt = tile_cols(p.N);
    dim3 grid((unsigned)ceil_div(rows, BM), (unsigned)ceil_div(p.N, bnt), (unsigned)(classes * p.splits));
    launch_variant<GEMM_DGRAD>(p, grid, bnt, (g.Cout & 3) == 0 && aligned16(dz), (g.Cout & 3) == 0 && aligned16(w), st);
    int rc = check_launch("conv_gemm_mfma<dgrad>");
    if (rc || p.splits == 1) return rc;
    return launch_splitk_finish(p, st);
}

// splits > 1: slabs only, the caller sums them (wgrad_reduce); splits == 1: `slabs` is dw itself (accumulate applies)
int launch_conv_gemm_wgrad(const sv_conv_geom& g, const float* x, const float* dz, float* slabs, int splits, hipStream_t st,
                           int accumulate, const float* xmask) {
    ConvGemmArgs p{};
    fill_common(p, g, GEMM_WGRAD);
    p.a = x; p.b = dz; p.bias = nullptr; p.c = slabs; p.act = SV_ACT_NONE; p.alpha = 0.f; p.amask = xmask;
    p.accumulate = splits == 1 && accumulate;
    p.M = g.KH * g.KW * g.Cin; p.N = g.Cout; p.K = g.N * g.Ho * g.Wo;
    p.k_per_split = (int)(ceil_div(ceil_div(p.K, splits), BK) * BK);
    const int bnt = tile_cols(p.N);
    dim3 grid((unsigned)ceil_div(p.M, BM), (unsigned)ceil_div(p.N, bnt), (unsigned)splits);
    launch_variant<GEMM_WGRAD>(p, grid, bnt, (g.Cin & 3) == 0 && aligned16(x) && aligned16(xmask), (g.Cout & 3) == 0 && aligned16(dz), st);
    return check_launch("conv_gemm_mfma<wgrad>");
}

}  // namespace sv
