// Implicit-GEMM convolution on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16): activations and filters stored as
// bf16, sums in f32.  The same three products as conv_gemm.hip (forward, input gradient, kernel gradient) for the
// strided 4x4 / 2x2 convs and transposed convs of the U-Net, the context encoder and their discriminators.
//
//   forward         Y[(n,oh,ow)][co]   = sum_k X(n, oh*SH-pt+kh, ow*SW-pl+kw, ci) * W[(kh,kw,ci)][co],   k = (kh,kw,ci)
//   input gradient  dX[(n,ih,iw)][ci]  = sum_k dZ(n, (ih+pt-kh)/SH, (iw+pl-kw)/SW, co) * W[kh][kw][ci][co], k = (kh,kw,co)
//   kernel gradient dW[(kh,kw,ci)][co] = sum_r X(...) * dZ[r][co],   r = (n,oh,ow), split over workgroups (f32 slabs)
//
// A workgroup (4 waves) owns a BM x BNT tile of the result (128 x 128, 128 x 64, 64 x 64 or 64 x 16: at these channel
// counts the kernel is bound by the operand traffic between L2 and the CUs, which falls with the tile edge) and walks the
// reduction in steps of 64.  Both
// operands are gathered from the NHWC / HWIO tensors in 16-byte pieces (8 channels) into registers one step ahead, then
// dropped into one of two kinds of LDS image:
//   ROW image [row][64 k]   - the operand's reduction index is contiguous in memory (x in the forward, dz and w in the
//                             input gradient): an MFMA fragment (8 consecutive k of one row) is one ds_read_b128;
//   TR  image [64 k][cols]  - the operand's ROW/COLUMN index is contiguous in memory (w in the forward, x and dz in the
//                             kernel gradient): fragments come out of ds_read_b64_tr_b16, the transposing LDS read.
// Both images are XOR-swizzled so that the stores and both kinds of read are free of bank conflicts.
// Lane l of an MFMA feeds A[row l % 16][k 8 (l / 16) .. + 7], B[k 8 (l / 16) .. + 7][col l % 16] and receives
// D[4 (l / 16) + r][l % 16].
#include "tile_common.h"

namespace sv {

typedef __bf16 bf16_t;
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef float v4f_b __attribute__((ext_vector_type(4)));

enum { BGEMM_FWD = 0, BGEMM_DGRAD = 1, BGEMM_WGRAD = 2 };

struct ConvGemmBf16Args {
    sv_conv_geom g;
    const bf16_t* a;    // FWD/WGRAD: x;  DGRAD: dz
    const bf16_t* b;    // FWD/DGRAD: w (HWIO);  WGRAD: dz
    const float* bias;  // FWD / DGRAD epilogue (may be null)
    bf16_t* c;          // FWD: y;  DGRAD: dx
    float* cf;          // WGRAD: slabs [split][M][N] (or dw itself);  FWD / DGRAD with splits > 1: slabs [split][M][N]
    int M, N, K;
    int k_per_split;    // WGRAD: reduction elements per blockIdx.z
    int act;
    float alpha;
    FDiv f_wo, f_howo;  // FWD/WGRAD rows r -> (n, oh, ow);  DGRAD: divide by W, H*W
    FDiv f_ck, f_kw;    // reduction / row index (tap, channel): divide by the channel count, then by KW
    int one_out;        // DGRAD: the conv has a single output position: one row class per input pixel
    int splits;         // FWD / DGRAD: > 1 cuts the reduction over blockIdx.z, raw partial sums go to cf
    int accumulate;     // WGRAD with a single split: cf is dw itself and the tile is added to it when set
    int va, vb;         // the A / B operand can be gathered in 16-byte pieces (channel count % 8 == 0, 16-byte aligned base)
    unsigned a_bytes, b_bytes;  // sizes of the A / B tensors: the 16-byte gathers are range-checked buffer loads
};

constexpr int BBK = 64;

// ---- LDS images (byte offsets) -----------------------------------------------------------------------------------------
// ROW image: 128-byte rows of 64 k; 16-byte chunk c of row r sits at chunk c ^ (r & 7).
__device__ __forceinline__ int row_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }
// TR image with 64 columns: 128-byte rows of 64 columns; 32-byte segment s (16 columns) of row k sits at segment s ^ f(k).
__device__ __forceinline__ int tr64_off(int k, int col) {
    const int f = ((k >> 1) & 1) | (((k >> 3) & 1) << 1);
    return k * 128 + ((((col >> 4) ^ f)) << 5) + ((col & 15) << 1);
}
// TR image with 16 columns: 32-byte rows, rows 8..15 of every 16 swapped with the 4 rows 4 apart.
__device__ __forceinline__ int tr16_off(int k, int col) {
    return 512 * (k >> 4) + 32 * ((k & 15) ^ (((k >> 3) & 1) << 2)) + (col << 1);
}

typedef v4bf __attribute__((address_space(3))) * lds_v4bf_ptr;

// 8 consecutive k (k = 32 kk + 8 lk .. + 7) of column c0 + li out of a TR image: two transposing reads of 4 k each.
template <int COLS>
__device__ __forceinline__ v8bf tr_frag(const unsigned char* img, int kk, int lk, int li, int c0) {
    const int k = 32 * kk + 8 * lk + (li >> 2);
    const int col = c0 + 4 * (li & 3);
    const int o0 = COLS == 64 ? tr64_off(k, col) : tr16_off(k, col);
    const int o1 = COLS == 64 ? tr64_off(k + 4, col) : tr16_off(k + 4, col);
    const v4bf lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4bf_ptr)(img + o0));
    const v4bf hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4bf_ptr)(img + o1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// 16 bytes at element offset `elem` of a tensor, zeros when !ok: a buffer load whose offset is pushed out of the
// descriptor's range for the dead pieces (the hardware answers those with zeros), so the gather has no branches.
typedef unsigned int v4u_b __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16(__amdgpu_buffer_rsrc_t rs, int elem, bool ok) {
    const v4u_b v = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? elem * 2 : (int)0x80000000, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ unsigned short bf_bits(bf16_t v) { return __builtin_bit_cast(unsigned short, v); }
__device__ __forceinline__ void put_elem(uint4& v, int e, unsigned short bits) {
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
    w[e >> 1] |= (uint32_t)bits << ((e & 1) * 16);
}

// BBM x BNT: the result tile.  BNT >= 64: the waves sit 2 x 2, each on (BBM / 2) x (BNT / 2);  BNT == 16: 4 x 1, each on
// (BBM / 4) x 16.  Images wider than 64 columns are kept as side-by-side 64-column images.
template <int MODE, int BBM, int BNT>
__global__ __launch_bounds__(256, (MODE == 2 || BBM * BNT >= 128 * 128) ? 2 : 3) void conv_gemm_bf16(ConvGemmBf16Args p) {
    constexpr int WAVES_N = BNT >= 64 ? 2 : 1, WAVES_M = 4 / WAVES_N;
    constexpr int TI = BBM / WAVES_M / 16, TJ = BNT / WAVES_N / 16;
    constexpr int NAQ = BBM / 32;                   // 16-byte A pieces per thread and step
    constexpr int NBQ = BNT >= 64 ? BNT / 32 : 1;   // 16-byte B pieces per thread and step (BNT == 16: threads 0..127 only)
    constexpr int A_BYTES = BBM * 128, B_BYTES = BNT >= 64 ? BNT * 128 : 2048;
    constexpr int CP = BNT + 8;                     // pitch (columns) of the epilogue's bf16 output tile, which reuses the images
    constexpr int C_BYTES = BBM * CP * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + B_BYTES > C_BYTES ? A_BYTES + B_BYTES : C_BYTES];
    unsigned char* As = smem;
    unsigned char* Bs = smem + A_BYTES;
    const sv_conv_geom& g = p.g;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.b), 0, p.b_bytes, 0x00020000);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.x * BBM, n0 = blockIdx.y * BNT;
    const int wm = (wave % WAVES_M) * (16 * TI), wn = (wave / WAVES_M) * (16 * TJ);

    int kbeg = 0, kend = p.K;
    if (MODE == BGEMM_WGRAD) {
        kbeg = blockIdx.z * p.k_per_split;
        kend = min(p.K, kbeg + p.k_per_split);
    }

    // ---- DGRAD row classes (see conv_gemm.hip): rows of one class share their set of contributing taps
    __shared__ int row_img[BBM], row_h[BBM], row_w[BBM];
    int kh0 = 0, kw0 = 0, nkw = 1;
    FDiv f_nkw = fd_make(1);
    const int split = p.splits > 1 ? blockIdx.z % p.splits : 0;
    if (MODE == BGEMM_DGRAD) {
        int Mc;
        const int z = p.splits > 1 ? blockIdx.z / p.splits : blockIdx.z;
        if (p.one_out) {
            const int ih = z / g.W, iw = z - ih * g.W;
            kh0 = ih + g.pad_t; kw0 = iw + g.pad_l;
            Mc = g.N;
            kend = (kh0 < g.KH && kw0 < g.KW) ? g.Cout : 0;
            if (t < BBM) {
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
            if (t < BBM) {
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
    if (MODE != BGEMM_WGRAD && p.splits > 1) {
        const int kper = ((kend + p.splits - 1) / p.splits + BBK - 1) / BBK * BBK;
        kbeg = split * kper;
        kend = min(kend, kbeg + kper);  // may leave an empty range: the workgroup then stores zeros
    }

    // ---- per-thread bookkeeping of the gathers.  Piece q = t + 256 i (i = 0, 1): image row q >> 3, 16-byte chunk q & 7.
    // ROW images (FWD / DGRAD A): image row = result row;  TR images (WGRAD A, B; FWD B): image row = reduction index.
    const int qc = t & 7;  // chunk of every piece of this thread
    constexpr int NRW = MODE == BGEMM_WGRAD ? 1 : NAQ;  // result rows this thread gathers A for
    // r_base: element offset of the row's tap-(0, 0), channel-0 operand element (it may lie outside the tensor: only
    // addresses whose bounds test passes are dereferenced); a reduction step then adds one offset shared by all rows
    int r_img[NRW], r_h[NRW], r_w[NRW];
    int r_base[NRW];
    bool r_ok[NRW];
    if (MODE == BGEMM_FWD) {
#pragma unroll
        for (int i = 0; i < NAQ; ++i) {
            const int m = m0 + (t >> 3) + 32 * i;
            r_ok[i] = m < p.M;
            const int mm = r_ok[i] ? m : 0;
            const int img = fd(mm, p.f_howo);
            const int rem = mm - img * p.f_howo.d;
            const int hh = fd(rem, p.f_wo);
            r_img[i] = img; r_h[i] = hh * g.SH - g.pad_t; r_w[i] = (rem - hh * p.f_wo.d) * g.SW - g.pad_l;
            r_base[i] = ((img * g.H + r_h[i]) * g.W + r_w[i]) * g.Cin;
        }
    } else if (MODE == BGEMM_DGRAD) {
        // A row of a class reaches output pixel (ohb - jh, owb - jw) through the class's tap (jh, jw): the divisions by the
        // strides are exact by construction of the class and done once here, not once per reduction step
#pragma unroll
        for (int i = 0; i < NAQ; ++i) {
            const int rl = (t >> 3) + 32 * i;
            r_img[i] = row_img[rl];
            r_ok[i] = r_img[i] >= 0;
            r_h[i] = p.one_out ? 0 : (row_h[rl] + g.pad_t - kh0) / g.SH;
            r_w[i] = p.one_out ? 0 : (row_w[rl] + g.pad_l - kw0) / g.SW;
            r_base[i] = ((r_img[i] * g.Ho + r_h[i]) * g.Wo + r_w[i]) * g.Cout;
        }
    }
    // WGRAD: piece i of this thread covers reduction row (t >> 3) + 32 (i & 1) and the 8 result rows
    // m = m0 + 64 (i >> 1) + 8 qc .. + 7, which share a tap when the A operand is vectorised
    constexpr int NAM = MODE == BGEMM_WGRAD ? BBM / 64 : 1;
    int a_kh[NAM], a_kw[NAM], a_ci[NAM], a_off[NAM];
    bool a_ok[NAM];
    if (MODE == BGEMM_WGRAD) {
#pragma unroll
        for (int s = 0; s < NAM; ++s) {
            const int m = m0 + 64 * s + 8 * qc;
            a_ok[s] = m < p.M;
            const int tap = a_ok[s] ? fd(m, p.f_ck) : 0;
            a_ci[s] = a_ok[s] ? m - tap * g.Cin : 0;
            a_kh[s] = fd(tap, p.f_kw);
            a_kw[s] = tap - a_kh[s] * g.KW;
            a_off[s] = (a_kh[s] * g.W + a_kw[s]) * g.Cin + a_ci[s];
        }
    }

    v4f_b acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = v4f_b{0.f, 0.f, 0.f, 0.f};

    // tap decode of a reduction index of the FWD / DGRAD products
    auto fwd_tap = [&](int k, int& kh, int& kw, int& ci) {
        const int tap = fd(k, p.f_ck);
        ci = k - tap * g.Cin;
        kh = fd(tap, p.f_kw);
        kw = tap - kh * g.KW;
    };
    auto dgrad_tap = [&](int k, int& kh, int& kw, int& co, int& jh, int& jw) {
        const int j = fd(k, p.f_ck);
        co = k - j * g.Cout;
        jh = p.one_out ? 0 : fd(j, f_nkw);
        jw = p.one_out ? 0 : j - jh * nkw;
        kh = kh0 + jh * g.SH;
        kw = kw0 + jw * g.SW;
    };

    // global -> registers for one reduction step (issued one step ahead of its use)
    auto gather = [&](int k0, uint4 (&av)[NAQ], uint4 (&bv)[NBQ]) {
        // ------------------------------------------------------------------ A
        if (MODE == BGEMM_FWD) {
            const int kt = k0 + 8 * qc;
            if (p.va) {
                const bool kok = kt < kend;
                int kh, kw, ci;
                fwd_tap(kok ? kt : 0, kh, kw, ci);
                const int koff = (kh * g.W + kw) * g.Cin + ci;
#pragma unroll
                for (int i = 0; i < NAQ; ++i) {
                    const bool ok = kok && r_ok[i] && (unsigned)(r_h[i] + kh) < (unsigned)g.H && (unsigned)(r_w[i] + kw) < (unsigned)g.W;
                    av[i] = ld16(rs_a, r_base[i] + koff, ok);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NAQ; ++i) av[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = kt + e;
                    const bool kok = k < kend;
                    int kh, kw, ci;
                    fwd_tap(kok ? k : 0, kh, kw, ci);
#pragma unroll
                    for (int i = 0; i < NAQ; ++i) {
                        const int ih = r_h[i] + kh, iw = r_w[i] + kw;
                        const bool ok = kok && r_ok[i] && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
                        if (ok) put_elem(av[i], e, bf_bits(p.a[(((int64_t)r_img[i] * g.H + ih) * g.W + iw) * g.Cin + ci]));
                    }
                }
            }
        } else if (MODE == BGEMM_DGRAD) {
            const int kt = k0 + 8 * qc;
            if (p.va) {
                const bool kok = kt < kend;
                int kh, kw, co, jh, jw;
                dgrad_tap(kok ? kt : 0, kh, kw, co, jh, jw);
                const int koff = co - (jh * g.Wo + jw) * g.Cout;
#pragma unroll
                for (int i = 0; i < NAQ; ++i) {
                    const bool ok = kok && r_ok[i] && (unsigned)(r_h[i] - jh) < (unsigned)g.Ho && (unsigned)(r_w[i] - jw) < (unsigned)g.Wo;
                    av[i] = ld16(rs_a, r_base[i] + koff, ok);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NAQ; ++i) av[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool kok = kt + e < kend;
                    int kh, kw, co, jh, jw;
                    dgrad_tap(kok ? kt + e : 0, kh, kw, co, jh, jw);
#pragma unroll
                    for (int i = 0; i < NAQ; ++i) {
                        const int oh = r_h[i] - jh, ow = r_w[i] - jw;
                        if (kok && r_ok[i] && oh >= 0 && ow >= 0 && oh < g.Ho && ow < g.Wo)
                            put_elem(av[i], e, bf_bits(p.a[(((int64_t)r_img[i] * g.Ho + oh) * g.Wo + ow) * g.Cout + co]));
                    }
                }
            }
        } else if (p.va) {
            // the two reduction rows of this thread: pixel decode and the offset of the pixel's window once per row
            int h0[2], w0[2], rb[2];
            bool rok[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = k0 + (t >> 3) + 32 * j;
                rok[j] = r < kend;
                const int rr = rok[j] ? r : 0;
                const int img = fd(rr, p.f_howo);
                const int rem = rr - img * p.f_howo.d;
                const int oh = fd(rem, p.f_wo), ow = rem - oh * p.f_wo.d;
                h0[j] = oh * g.SH - g.pad_t; w0[j] = ow * g.SW - g.pad_l;
                rb[j] = ((img * g.H + h0[j]) * g.W + w0[j]) * g.Cin;
            }
#pragma unroll
            for (int i = 0; i < NAQ; ++i) {
                const int j = i & 1, sm = i >> 1;  // sm: which 64-row group of the tile
                const bool ok = rok[j] && a_ok[sm] && (unsigned)(h0[j] + a_kh[sm]) < (unsigned)g.H &&
                                (unsigned)(w0[j] + a_kw[sm]) < (unsigned)g.W;
                av[i] = ld16(rs_a, rb[j] + a_off[sm], ok);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NAQ; ++i) {
                const int r = k0 + (t >> 3) + 32 * (i & 1);
                const int sm = i >> 1;
                const bool rok = r < kend;
                const int rr = rok ? r : 0;
                const int img = fd(rr, p.f_howo);
                const int rem = rr - img * p.f_howo.d;
                const int oh = fd(rem, p.f_wo), ow = rem - oh * p.f_wo.d;
                {
                    av[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int m = m0 + 64 * sm + 8 * qc + e;
                        const bool mok = m < p.M;
                        const int tap = fd(mok ? m : 0, p.f_ck);
                        const int ci = (mok ? m : 0) - tap * g.Cin;
                        const int kh = fd(tap, p.f_kw), kw = tap - kh * g.KW;
                        const int ih = oh * g.SH - g.pad_t + kh, iw = ow * g.SW - g.pad_l + kw;
                        if (rok && mok && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)
                            put_elem(av[i], e, bf_bits(p.a[(((int64_t)img * g.H + ih) * g.W + iw) * g.Cin + ci]));
                    }
                }
            }
        }
        // ------------------------------------------------------------------ B
        if (MODE == BGEMM_DGRAD) {  // ROW image [n][k]: w[tap][ci = n][co = k]
            const int kt = k0 + 8 * qc;
            if (p.vb) {
                const bool kok = kt < kend;
                int kh, kw, co, jh, jw;
                dgrad_tap(kok ? kt : 0, kh, kw, co, jh, jw);
                const int boff = (kh * g.KW + kw) * g.Cin * g.Cout + co;  // the filter tensor is far below 2^31 elements
#pragma unroll
                for (int i = 0; i < NBQ; ++i) {
                    const int nl = (t >> 3) + 32 * i;
                    const int n = n0 + nl;
                    bv[i] = ld16(rs_b, boff + n * g.Cout, kok && nl < BNT && n < p.N);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NBQ; ++i) bv[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool kok = kt + e < kend;
                    int kh, kw, co, jh, jw;
                    dgrad_tap(kok ? kt + e : 0, kh, kw, co, jh, jw);
                    const int tap = kh * g.KW + kw;
#pragma unroll
                    for (int i = 0; i < NBQ; ++i) {
                        const int nl = (t >> 3) + 32 * i;
                        const int n = n0 + nl;
                        if (kok && nl < BNT && n < p.N) put_elem(bv[i], e, bf_bits(p.b[((int64_t)tap * g.Cin + n) * g.Cout + co]));
                    }
                }
            }
        } else if (p.vb) {  // TR image [k][n]: FWD w[k][n], WGRAD dz[r][n]; n contiguous in memory
#pragma unroll
            for (int i = 0; i < NBQ; ++i) {
                const int kr = BNT >= 64 ? (t >> 3) + 32 * (i & 1) : (t >> 1);
                const int c8 = BNT >= 64 ? 8 * (i >> 1) + qc : (t & 1);
                const int n = n0 + 8 * c8;
                const bool live = (BNT >= 64 || t < 128) && k0 + kr < kend;
                bv[i] = ld16(rs_b, k0 * p.N + (kr * p.N + n), live && n + 8 <= p.N);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NBQ; ++i) {
                const int kr = BNT >= 64 ? (t >> 3) + 32 * (i & 1) : (t >> 1);
                const int c8 = BNT >= 64 ? 8 * (i >> 1) + qc : (t & 1);
                const int k = k0 + kr;
                const int n = n0 + 8 * c8;
                const bool live = (BNT >= 64 || t < 128) && k < kend;
                bv[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (live && n + e < p.N) put_elem(bv[i], e, bf_bits(p.b[(int64_t)k * p.N + n + e]));
            }
        }
    };

    uint4 av[NAQ], bv[NBQ];
    if (kbeg < kend) gather(kbeg, av, bv);
    for (int k0 = kbeg; k0 < kend; k0 += BBK) {
        __syncthreads();  // the previous step's fragment reads are done
#pragma unroll
        for (int i = 0; i < NAQ; ++i) {
            const int off = MODE == BGEMM_WGRAD ? 8192 * (i >> 1) + tr64_off((t >> 3) + 32 * (i & 1), 8 * qc)
                                                : row_off((t >> 3) + 32 * i, qc);
            *reinterpret_cast<uint4*>(As + off) = av[i];
        }
#pragma unroll
        for (int i = 0; i < NBQ; ++i) {
            if (MODE == BGEMM_DGRAD) {
                const int nl = (t >> 3) + 32 * i;
                if (nl < BNT) *reinterpret_cast<uint4*>(Bs + row_off(nl, qc)) = bv[i];
            } else if (BNT >= 64) {
                *reinterpret_cast<uint4*>(Bs + 8192 * (i >> 1) + tr64_off((t >> 3) + 32 * (i & 1), 8 * qc)) = bv[i];
            } else if (t < 128) {
                *reinterpret_cast<uint4*>(Bs + tr16_off(t >> 1, 8 * (t & 1))) = bv[i];
            }
        }
        __syncthreads();
        if (k0 + BBK < kend) gather(k0 + BBK, av, bv);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            v8bf a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int c0 = wm + 16 * i;
                if (MODE == BGEMM_WGRAD) a[i] = tr_frag<64>(As + 8192 * (c0 >> 6), kk, lk, li, c0 & 63);
                else a[i] = *reinterpret_cast<const v8bf*>(As + row_off(c0 + li, 4 * kk + lk));
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int c0 = wn + 16 * j;
                if (MODE == BGEMM_DGRAD) b[j] = *reinterpret_cast<const v8bf*>(Bs + row_off(c0 + li, 4 * kk + lk));
                else if (BNT >= 64) b[j] = tr_frag<64>(Bs + 8192 * (c0 >> 6), kk, lk, li, c0 & 63);
                else b[j] = tr_frag<16>(Bs, kk, lk, li, c0);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue
    const bool raw = MODE == BGEMM_WGRAD || p.splits > 1;
    if (raw) {  // f32 partial sums (kernel gradient slabs / cut reductions), straight from the accumulators
        float* out = p.cf + (size_t)(MODE == BGEMM_WGRAD ? blockIdx.z : split) * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int n = n0 + wn + 16 * j + li;
                if (n >= p.N) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = wm + 16 * i + 4 * lk + r;
                    float v = acc[i][j][r];
                    if (MODE == BGEMM_DGRAD) {
                        if (row_img[rl] < 0) continue;
                        out[(((size_t)row_img[rl] * g.H + row_h[rl]) * g.W + row_w[rl]) * g.Cin + n] = v;
                    } else {
                        const int m = m0 + rl;
                        if (m >= p.M) continue;
                        if (MODE == BGEMM_WGRAD && p.accumulate) v += out[(size_t)m * p.N + n];
                        out[(size_t)m * p.N + n] = v;
                    }
                }
            }
        return;
    }
    // bias + activation, rounded to bf16 into an LDS tile [row][BNT] (pitch BNT + 8 columns), then 16-byte stores of
    // whole channel runs
    __syncthreads();  // all fragment reads of the last step are done: the tile may overwrite the images
    bf16_t* Cs = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int nl = wn + 16 * j + li;
            const int n = n0 + nl;
            const float bb = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = wm + 16 * i + 4 * lk + r;
                Cs[rl * CP + nl] = (bf16_t)act_apply(acc[i][j][r] + bb, p.act, p.alpha);
            }
        }
    __syncthreads();
    const bool vec_out = (p.N & 7) == 0 && (reinterpret_cast<uintptr_t>(p.c) & 15) == 0;
    constexpr int CH = BNT / 8;  // 16-byte chunks per row of the tile
    for (int q = t; q < BBM * CH; q += 256) {
        const int rl = q / CH, c8 = q - rl * CH;
        const int n = n0 + 8 * c8;
        if (n >= p.N) continue;
        size_t row_base;
        if (MODE == BGEMM_DGRAD) {
            if (row_img[rl] < 0) continue;
            row_base = (((size_t)row_img[rl] * g.H + row_h[rl]) * g.W + row_w[rl]) * g.Cin;
        } else {
            const int m = m0 + rl;
            if (m >= p.M) continue;
            row_base = (size_t)m * p.N;
        }
        if (vec_out && n + 8 <= p.N) {
            *reinterpret_cast<uint4*>(p.c + row_base + n) = *reinterpret_cast<const uint4*>(Cs + rl * CP + 8 * c8);
        } else {
            for (int e = 0; e < 8 && n + e < p.N; ++e) p.c[row_base + n + e] = Cs[rl * CP + 8 * c8 + e];
        }
    }
}

// out[i] = bf16(act(sum_s slabs[s][i] + bias[i % N])), splits summed in index order (bit-reproducible)
__global__ __launch_bounds__(256) void splitk_finish_bf16_k(const float* __restrict__ slabs, int splits, int64_t MN, int N,
                                                            const float* __restrict__ bias, int act, float alpha,
                                                            bf16_t* __restrict__ out) {
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
    out[i] = (bf16_t)act_apply(s, act, alpha);
}

// ------------------------------------------------------------------------------------------------------ host side
// conv_gemm.hip: shape test and split counts of the f32 kernel.  Cut reductions (few result tiles) always run on 64-row
// tiles here, as there, and never finer than the f32 kernel cuts them, so sv_conv2d_scratch_bytes covers both kernels;
// the kernel-gradient workspace query takes the larger of the two kernels' needs (conv_gemm_bf16_wgrad_workspace).
bool conv_gemm_worthwhile(const sv_conv_geom& g, int mode);
bool conv_gemm_shape_ok(const sv_conv_geom& g, int mode, int min_k);
int conv_gemm_splits(const sv_conv_geom& g, int mode);
size_t conv_gemm_splitk_workspace(const sv_conv_geom& g, int mode);

static bool aligned16b(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// Every geometry the index decoding covers runs on this kernel, short reductions and odd channel counts included (their
// operands are gathered element by element): the direct kernels' 2-byte loads through the vector path are slower still.
bool conv_gemm_bf16_ok(const sv_conv_geom& g, int mode) {
    // the gathers address all three tensors with 32-bit byte offsets, 2^31 being the "no such piece" offset
    const int64_t lim = 1ll << 30;
    return (int64_t)g.KH * g.KW * g.Cin * g.Cout < lim && (int64_t)g.N * g.H * g.W * g.Cin < lim &&
           (int64_t)g.N * g.Ho * g.Wo * g.Cout < lim && conv_gemm_shape_ok(g, mode, 1);
}
static unsigned bytes_x(const sv_conv_geom& g) { return (unsigned)((int64_t)g.N * g.H * g.W * g.Cin * 2); }
static unsigned bytes_z(const sv_conv_geom& g) { return (unsigned)((int64_t)g.N * g.Ho * g.Wo * g.Cout * 2); }
static unsigned bytes_w(const sv_conv_geom& g) { return (unsigned)((int64_t)g.KH * g.KW * g.Cin * g.Cout * 2); }

// Result tile for `rows` x `ncols` results in `zmult` independent groups (row classes, reduction splits): the largest
// tile that still leaves every CU more than one workgroup.
struct TileSel {
    int bm, bnt;
};
static TileSel pick_tile(int64_t rows, int ncols, int64_t zmult) {
    if (ncols <= 16) return {64, 16};
    if (ncols > 64 && ceil_div(rows, 128) * ceil_div(ncols, 128) * zmult >= 384) return {128, 128};
    if (ceil_div(rows, 128) * ceil_div(ncols, 64) * zmult >= 384) return {128, 64};
    return {64, 64};
}

static int bf16_splits(const sv_conv_geom& g, int mode, int64_t kdim) {
    int s = conv_gemm_splits(g, mode);
    const int64_t maxs = kdim / (2 * BBK);  // keep at least two steps per workgroup
    if (s > maxs) s = (int)maxs;
    return s < 2 ? 1 : s;
}

static void fill_common_b(ConvGemmBf16Args& p, const sv_conv_geom& g, int mode) {
    p.g = g;
    if (mode == BGEMM_DGRAD) {
        p.f_wo = fd_make(g.W); p.f_howo = fd_make(g.H * g.W); p.f_ck = fd_make(g.Cout);
    } else {
        p.f_wo = fd_make(g.Wo); p.f_howo = fd_make(g.Ho * g.Wo); p.f_ck = fd_make(g.Cin);
    }
    p.f_kw = fd_make(g.KW);
}

template <int MODE>
static void launch_b(const ConvGemmBf16Args& p, int64_t rows, int64_t zdim, TileSel ts, hipStream_t st) {
    dim3 grid((unsigned)ceil_div(rows, ts.bm), (unsigned)ceil_div(p.N, ts.bnt), (unsigned)zdim);
    if (ts.bm == 128 && ts.bnt == 128) hipLaunchKernelGGL((conv_gemm_bf16<MODE, 128, 128>), grid, dim3(256), 0, st, p);
    else if (ts.bm == 128) hipLaunchKernelGGL((conv_gemm_bf16<MODE, 128, 64>), grid, dim3(256), 0, st, p);
    else if (ts.bnt == 64) hipLaunchKernelGGL((conv_gemm_bf16<MODE, 64, 64>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_gemm_bf16<MODE, 64, 16>), grid, dim3(256), 0, st, p);
}

static int finish_b(const ConvGemmBf16Args& p, hipStream_t st) {
    const int64_t mn = (int64_t)p.M * p.N;
    hipLaunchKernelGGL(splitk_finish_bf16_k, dim3((unsigned)ceil_div(mn, 256)), dim3(256), 0, st, p.cf, p.splits, mn, p.N,
                       p.bias, p.act, p.alpha, p.c);
    return check_launch("splitk_finish_bf16_k");
}

int launch_conv_gemm_bf16_fwd(const sv_conv_geom& g, const void* x, const void* w, const float* bias, void* y, int act,
                              float alpha, void* ws, size_t ws_bytes, hipStream_t st) {
    ConvGemmBf16Args p{};
    fill_common_b(p, g, BGEMM_FWD);
    p.a = (const bf16_t*)x; p.b = (const bf16_t*)w; p.bias = bias; p.c = (bf16_t*)y; p.act = act; p.alpha = alpha;
    p.M = g.N * g.Ho * g.Wo; p.N = g.Cout; p.K = g.KH * g.KW * g.Cin;
    p.a_bytes = bytes_x(g); p.b_bytes = bytes_w(g);
    p.va = (g.Cin & 7) == 0 && aligned16b(x);
    p.vb = (g.Cout & 7) == 0 && aligned16b(w);
    p.splits = 1;
    const size_t need = conv_gemm_splitk_workspace(g, BGEMM_FWD);
    if (need > 0 && ws != nullptr && ws_bytes >= need) {
        p.splits = bf16_splits(g, BGEMM_FWD, p.K);
        p.cf = (float*)ws;
    }
    const TileSel ts = p.splits > 1 ? TileSel{64, p.N <= 16 ? 16 : 64} : pick_tile(p.M, p.N, 1);
    launch_b<BGEMM_FWD>(p, p.M, p.splits, ts, st);
    int rc = check_launch("conv_gemm_bf16<fwd>");
    if (rc || p.splits == 1) return rc;
    return finish_b(p, st);
}

int launch_conv_gemm_bf16_dgrad(const sv_conv_geom& g, const void* dz, const void* w, const float* bias, void* dx, int act,
                                float alpha, void* ws, size_t ws_bytes, hipStream_t st) {
    ConvGemmBf16Args p{};
    fill_common_b(p, g, BGEMM_DGRAD);
    p.a = (const bf16_t*)dz; p.b = (const bf16_t*)w; p.bias = bias; p.c = (bf16_t*)dx; p.act = act; p.alpha = alpha;
    p.M = g.N * g.H * g.W; p.N = g.Cin; p.K = g.KH * g.KW * g.Cout;
    p.one_out = g.Ho == 1 && g.Wo == 1;
    p.a_bytes = bytes_z(g); p.b_bytes = bytes_w(g);
    p.va = (g.Cout & 7) == 0 && aligned16b(dz);
    p.vb = (g.Cout & 7) == 0 && aligned16b(w);
    int classes, rows;
    int64_t kclass;
    if (p.one_out) {
        classes = g.H * g.W; rows = g.N; kclass = g.Cout;
    } else {
        classes = g.SH * g.SW;
        rows = g.N * (int)ceil_div(g.H, g.SH) * (int)ceil_div(g.W, g.SW);
        kclass = (int64_t)ceil_div(g.KH, g.SH) * ceil_div(g.KW, g.SW) * g.Cout;
    }
    p.splits = 1;
    const size_t need = conv_gemm_splitk_workspace(g, BGEMM_DGRAD);
    if (need > 0 && ws != nullptr && ws_bytes >= need) {
        p.splits = bf16_splits(g, BGEMM_DGRAD, kclass);
        p.cf = (float*)ws;
    }
    const TileSel ts = p.splits > 1 ? TileSel{64, p.N <= 16 ? 16 : 64} : pick_tile(rows, p.N, classes);
    launch_b<BGEMM_DGRAD>(p, rows, (int64_t)classes * p.splits, ts, st);
    int rc = check_launch("conv_gemm_bf16<dgrad>");
    if (rc || p.splits == 1) return rc;
    return finish_b(p, st);
}

// Kernel gradient: result (kh, kw, ci) x co, reduction over the pixels, cut into `splits` ranges so that the launch has
// about 512 workgroups; each range leaves a float32 slab.
struct WgradSel {
    TileSel ts;
    int splits;
};
static WgradSel bf16_wgrad_plan(const sv_conv_geom& g) {
    const int64_t M = (int64_t)g.KH * g.KW * g.Cin, R = (int64_t)g.N * g.Ho * g.Wo;
    const int N = g.Cout;
    WgradSel w;
    // A result with many tiles is not cut at all (each tile goes straight into dw): then the largest tile that still gives
    // every CU a workgroup.  Otherwise the largest tile whose cut reduction keeps at least four steps per workgroup at
    // about 512 workgroups.
    const TileSel cands[3] = {{128, 128}, {128, 64}, {64, 64}};
    w.ts = TileSel{64, N <= 16 ? 16 : 64};
    if (N > 16) {
        const bool uncut = ceil_div(M, 64) * ceil_div(N, 64) >= 384;
        for (const TileSel& c : cands) {
            if (c.bnt == 128 && N <= 64) continue;
            if (c.bm == 128 && M <= 64) continue;
            const int64_t tiles = ceil_div(M, c.bm) * ceil_div(N, c.bnt);
            if (uncut ? tiles >= 256 : ceil_div(512, tiles) * 4 * BBK <= R) { w.ts = c; break; }
        }
    }
    const int64_t tiles = ceil_div(M, w.ts.bm) * ceil_div(N, w.ts.bnt);
    int64_t s = ceil_div(512, tiles);
    const int64_t maxs = ceil_div(R, 2 * BBK);
    if (s > maxs) s = maxs;
    if (s < 1 || (N > 16 && ceil_div(M, 64) * ceil_div(N, 64) >= 384)) s = 1;
    if (s > 1024) s = 1024;
    w.splits = (int)s;
    return w;
}

int conv_gemm_bf16_wgrad_splits(const sv_conv_geom& g) { return bf16_wgrad_plan(g).splits; }

size_t conv_gemm_bf16_wgrad_workspace(const sv_conv_geom& g) {
    if (!conv_gemm_bf16_ok(g, BGEMM_WGRAD)) return 0;
    return (size_t)bf16_wgrad_plan(g).splits * g.KH * g.KW * g.Cin * g.Cout * sizeof(float);
}

// splits > 1: slabs only, the caller sums them (wgrad_reduce); splits == 1: `slabs` is dw itself (accumulate applies)
int launch_conv_gemm_bf16_wgrad(const sv_conv_geom& g, const void* x, const void* dz, float* slabs, int splits,
                                hipStream_t st, int accumulate) {
    ConvGemmBf16Args p{};
    fill_common_b(p, g, BGEMM_WGRAD);
    p.a = (const bf16_t*)x; p.b = (const bf16_t*)dz; p.cf = slabs; p.act = SV_ACT_NONE;
    p.accumulate = splits == 1 && accumulate;
    p.M = g.KH * g.KW * g.Cin; p.N = g.Cout; p.K = g.N * g.Ho * g.Wo;
    p.a_bytes = bytes_x(g); p.b_bytes = bytes_z(g);
    p.va = (g.Cin & 7) == 0 && aligned16b(x);
    p.vb = (g.Cout & 7) == 0 && aligned16b(dz);
    p.k_per_split = (int)(ceil_div(ceil_div(p.K, splits), BBK) * BBK);
    p.splits = 1;
    launch_b<BGEMM_WGRAD>(p, p.M, splits, bf16_wgrad_plan(g).ts, st);
    return check_launch("conv_gemm_bf16<wgrad>");
}

}  // namespace sv
