// Kernel (+ bias) gradients of 3x3 stride-1 SAME convs with few channels, all layers of a backward pass in one launch,
// entirely on the f32 matrix cores with channel-last LDS images (the gradient tape of run_model.py:281-289 for the conv
// stack of autoencoder.py:437-526).
//
//   dw[kh][kw][ci][co] = sum over samples and pixels p of x[p + (kh - 1, kw - 1)][ci] * dz[p][co]
//
// is a GEMM whose reduction runs over the pixels.  The tensor with FEWER channels plays S ("shifted": staged with a zero
// halo, its 9 shifted views are the GEMM rows (tap, cs)), the other one P ("plain": its channels are the GEMM columns):
//   SWAP = 0:  S = x,  P = dz          (Cin < Cout: 3 -> 8, 8 -> 16, 16 -> 32)
//   SWAP = 1:  S = dz, P = x           (Cin > Cout: 32 -> 16, 16 -> 8, 8 -> 3; the tap index comes out mirrored)
// so the 16-wide tiles are full except for the 3 / 8-channel pair.  Both images stay in the tensors' own NHWC order:
// staging is a copy (16-byte pieces; dz = dy * act'(y) and the scatter through the pooling argmax are formed on the way
// in), an upsampled x is read through (row >> 1, col >> 1) instead of being replicated, and one v_mfma_f32_16x16x4_f32
// takes its A operand (16 rows (tap, cs) x 4 consecutive pixels) and B operand (4 pixels x 16 channels) as single 4-byte
// LDS reads at a wave-uniform base plus a per-lane constant.  A workgroup walks passes of `ns` samples; the global
// loads of pass i + 1 are in flight while the matrix cores work on pass i.  Each workgroup leaves one slab of partial
// sums in dw order; wn_reduce_multi adds the slabs in a fixed order (bit-reproducible, no float atomics).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "tile_common.h"

namespace sv {

typedef float wn_v4f __attribute__((ext_vector_type(4)));

struct WnArgs {
    const float* x;       // [N][Hs][Ws][Cin]
    const float* dy;      // zmode 0/1: [N][Ho][Wo][Cout];  zmode 2: [N][Ho/2][Wo/2][Cout]
    const float* zy;      // activation output, extents of dy (zmode 1/2)
    const uint8_t* zidx;  // zmode 2: argmax slot of every pooling window
    float* slab;          // [gx][slab_floats]
    int N, Hs, Ws, Ho, Wo, up, zmode, with_bias;
    int ns;               // samples per pass
    int slab_floats;      // 9 * Cin * Cout + Cout rounded up to 4
    int hq_log;           // log2(Ho / 4)
    int dbg;              // diagnostic build only (see WN_DBG): 1 skip the matrix-core loop, 2 the LDS stores, 4 the global loads
    ActGradCoef zk;
};

// Ablation switches exist in the diagnostic build (make TRACE=1) only: they make the results invalid, so the release
// library has no way to turn them on.
#ifdef SV_TRACE
#define WN_DBG(a, bit) ((a).dbg & (bit))
#else
#define WN_DBG(a, bit) 0
#endif

#ifdef SV_TRACE
// diagnostic builds: thread 0 of every workgroup stores the 100 MHz wall clock at phase marks, 32 slots per workgroup:
// 0 start, 1 LDS cleared, per pass p (the first seven) 2 + 4 p pass entered, + 1 its loads landed, + 2 staged, + 3 matrix
// instructions done, 31 exit
__device__ unsigned long long* g_wn_trace = nullptr;
__device__ __forceinline__ void wn_mark(int slot, bool drain) {
    if (g_wn_trace == nullptr) return;
    if (drain) __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0 && slot < 32) g_wn_trace[(size_t)blockIdx.x * 32 + slot] = wall_clock64();
}
#define WN_MARK(slot, drain) wn_mark(slot, drain)
#else
#define WN_MARK(slot, drain)
#endif

// Order of one turn of a matrix-core loop: NM matrix instructions with the NR LDS reads of the NEXT turn's operands
// placed two behind each of the first ones, so the reads issue in the shadow of the matrix pipe and the last of them
// have the remaining instructions' time to land.
template <int NM, int NR, int... I>
__device__ __forceinline__ void wn_interleave_seq(std::integer_sequence<int, I...>) {
    ((__builtin_amdgcn_sched_group_barrier(0x008, 1, 0),
      __builtin_amdgcn_sched_group_barrier(0x100, (NR - 2 * I) >= 2 ? 2 : ((NR - 2 * I) > 0 ? (NR - 2 * I) : 0), 0)), ...);
}
template <int NM, int NR>
__device__ __forceinline__ void wn_interleave() {
    static_assert(NR <= 2 * NM, "two reads per matrix instruction at most");
    wn_interleave_seq<NM, NR>(std::make_integer_sequence<int, NM>{});
}

constexpr int kWnUnits = 4;  // 16-byte pieces per thread, tensor and pass held in registers between load and store

struct WnRegs {
    float4 sv[kWnUnits], sy[kWnUnits];                 // S side: values (+ activation output when S = dz)
    float4 pv[kWnUnits], py[kWnUnits];                 // P side
    uint32_t pk[kWnUnits];
};

// Floats from one row of the S image to the next.  With 3 channels a 16-row tile reads two kernel rows of the image in
// one LDS cycle; a pitch of 16 (mod 32) floats puts them on disjoint banks (102 floats, the packed pitch of a 34-pixel
// row, overlaps them on three).
constexpr int wn_rps(int CS, int WO) { return CS == 3 ? ((WO + 2) * 3 + 31 - 16) / 32 * 32 + 16 : (WO + 2) * CS; }

// PAIR (8-channel P, whose channels fill half of a 16-column tile): a column is (pixel of a horizontally adjacent pair,
// channel), the 4 reduction slots of an instruction are 4 consecutive PAIRS, and the rows are the shifted views of S over
// a window 4 columns wide, (kh, kw', cs) with kw' = 0..3: the first pixel of a pair meets kernel column kw' (kw' <= 2),
// the second one kw' - 1 (kw' >= 1).  3 x 4 x CS = 36 rows in 3 tiles serve 8 pixels, where 27 rows in 2 tiles served 4.
template <int CS, int CP, int SWAP, int WO, bool PAIR = false>
__device__ __forceinline__ void wn_body(const WnArgs& a, const int bx, const int gx) {
    static_assert(!PAIR || (CP == 8 && SWAP == 0), "pairs of pixels fill the 8 columns an 8-channel P leaves empty");
    constexpr int KWN = PAIR ? 4 : 3;                          // window columns a row of the GEMM can stand for
    constexpr int M = 3 * KWN * CS, MT = (M + 15) / 16, NT = (CP + 15) / 16, NTILE = MT * NT;
    constexpr int QX = PAIR ? WO / 8 : WO / 4;                 // reduction steps (pixel quads / quads of pairs) per image row
    constexpr int KP = PAIR ? 2 : 1;                           // pixels per reduction slot
    constexpr int MTC = (4 * CS) / 16;                         // tile holding the centre tap's rows (SWAP: the bias sums)
    constexpr int Cin = SWAP ? CP : CS, Cout = SWAP ? CS : CP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int Ho = a.Ho;
    constexpr int Wo = WO, RPS = wn_rps(CS, WO);
    const int sampleS = (Ho + 2) * RPS;
    const int pshift = (SWAP && a.up == 2) ? 1 : 0;            // P = x kept at its stored resolution
    const int PH = SWAP ? a.Hs : Ho, PW = SWAP ? a.Ws : Wo;
    (void)PW;
    const int sampleP = PH * PW * CP;
    float* s_lds = smem;
    float* p_lds = smem + ((a.ns * sampleS + 3) & ~3);

    // ---- staging: what a pass reads from global memory
    const float* s_src = SWAP ? a.dy : a.x;                    // [N][Ho][Wo][CS]
    const float* s_y = SWAP ? a.zy : nullptr;
    const bool s_act = SWAP && a.zmode == 1;
    const int s_sample = Ho * Wo * CS;                         // floats per sample (a multiple of 4: Wo is)
    const float* p_src = SWAP ? a.x : a.dy;
    const int pz = SWAP ? 0 : a.zmode;                         // 0 copy, 1 * act'(y), 2 scatter through the argmax
    const int p_h = SWAP ? a.Hs : (pz == 2 ? Ho >> 1 : Ho), p_w = SWAP ? a.Ws : (pz == 2 ? Wo >> 1 : Wo);
    const int p_sample = p_h * p_w * CP;
    const FDiv f_shw = fd_make(Ho * Wo), f_sw = fd_make(Wo), f_phw = fd_make(p_h * p_w), f_pw = fd_make(p_w);

    WN_MARK(0, false);
    WnRegs R;
    auto load_pass = [&](int n0) {
        const int ns = min(a.ns, a.N - n0);
        const int su = (ns * s_sample) >> 2, pu = (ns * p_sample) >> 2;
        const size_t sb = (size_t)n0 * s_sample, pb = (size_t)n0 * p_sample;
#pragma unroll
        for (int i = 0; i < kWnUnits; ++i) {
            const int u = min((int)threadIdx.x + i * 256, su - 1);
            R.sv[i] = *reinterpret_cast<const float4*>(s_src + sb + 4 * (size_t)u);
            if (SWAP) R.sy[i] = s_act ? *reinterpret_cast<const float4*>(s_y + sb + 4 * (size_t)u) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
#pragma unroll
        for (int i = 0; i < kWnUnits; ++i) {
            const int u = min((int)threadIdx.x + i * 256, pu - 1);
            R.pv[i] = *reinterpret_cast<const float4*>(p_src + pb + 4 * (size_t)u);
            if (!SWAP) {
                R.py[i] = pz ? *reinterpret_cast<const float4*>(a.zy + pb + 4 * (size_t)u) : make_float4(1.f, 1.f, 1.f, 1.f);
                R.pk[i] = pz == 2 ? *reinterpret_cast<const uint32_t*>(a.zidx + pb + 4 * (size_t)u) : 0u;
            }
        }
    };
    auto grad4 = [&](float4 v, float4 y, bool on) {
        if (on) {
            v.x *= act_grad_q(y.x, a.zk); v.y *= act_grad_q(y.y, a.zk);
            v.z *= act_grad_q(y.z, a.zk); v.w *= act_grad_q(y.w, a.zk);
        }
        return v;
    };
    // LDS positions of this thread's pieces: they depend on the thread and the piece only, not on the pass, so they are
    // formed once (the store phase of a pass is then the writes and the dz arithmetic, no index decoding)
    constexpr int SEL = CS % 4 == 0 ? 1 : 4;  // 3 channels: the 4 floats of a piece land on 4 positions
    int s_at[kWnUnits][SEL], p_at[kWnUnits];
#pragma unroll
    for (int i = 0; i < kWnUnits; ++i) {
        const int e = 4 * ((int)threadIdx.x + i * 256);
#pragma unroll
        for (int t = 0; t < SEL; ++t) {
            const int p = (e + t) / CS, c = (e + t) - p * CS;
            const int ss = fd(p, f_shw), pp = p - ss * (Ho * Wo);
            const int row = fd(pp, f_sw), col = pp - row * Wo;
            s_at[i][t] = ss * sampleS + (row + 1) * RPS + (col + 1) * CS + c;
        }
        if (SWAP || pz != 2) {
            p_at[i] = e;  // the image is the tensor
        } else {
            const int p = e / CP, c = e - p * CP;
            const int ss = fd(p, f_phw), pp = p - ss * (p_h * p_w);
            const int row = fd(pp, f_pw), col = pp - row * p_w;
            p_at[i] = ss * sampleP + ((2 * row) * Wo + 2 * col) * CP + c;
        }
    }
    auto store_pass = [&](int ns) {
        const int su = (ns * s_sample) >> 2, pu = (ns * p_sample) >> 2;
#pragma unroll
        for (int i = 0; i < kWnUnits; ++i) {
            const int u = threadIdx.x + i * 256;
            if (u < su) {
                const float4 v = SWAP ? grad4(R.sv[i], R.sy[i], s_act) : R.sv[i];
                if (CS % 4 == 0) {  // one pixel, 4 of its channels
                    *reinterpret_cast<float4*>(s_lds + s_at[i][0]) = v;
                } else {
                    s_lds[s_at[i][0]] = v.x; s_lds[s_at[i][1 % SEL]] = v.y;
                    s_lds[s_at[i][2 % SEL]] = v.z; s_lds[s_at[i][3 % SEL]] = v.w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kWnUnits; ++i) {
            const int u = threadIdx.x + i * 256;
            if (u < pu) {
                if (SWAP || pz != 2) {
                    const float4 v = SWAP ? R.pv[i] : grad4(R.pv[i], R.py[i], pz == 1);
                    *reinterpret_cast<float4*>(p_lds + p_at[i]) = v;
                } else {
                    const float4 v = grad4(R.pv[i], R.py[i], true);
                    const uint32_t k = R.pk[i];
                    float* q = p_lds + p_at[i];
                    const uint32_t k0 = k & 0xff, k1 = (k >> 8) & 0xff, k2 = (k >> 16) & 0xff, k3 = k >> 24;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float4 o = make_float4(k0 == (uint32_t)s ? v.x : 0.f, k1 == (uint32_t)s ? v.y : 0.f,
                                                     k2 == (uint32_t)s ? v.z : 0.f, k3 == (uint32_t)s ? v.w : 0.f);
                        *reinterpret_cast<float4*>(q + ((s >> 1) * Wo + (s & 1)) * CP) = o;
                    }
                }
            }
        }
    };

    // ---- per-lane operand offsets.  Row m = mt * 16 + li = (kh * 3 + kw) * CS + cs reads S[oh + kh][ow + kw + lk][cs]
    // (halo baked in); column n = nt * 16 + li reads P[oh][ow + lk][n].  Rows / columns past the end read finite values
    // of a neighbouring position and their results are never stored.
    int aoff[MT], boff[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + li;
        const int mm = m < M ? m : M - 1;  // unused rows read what the last used row reads (a broadcast, not a bank conflict)
        const int kh = mm / (KWN * CS), r = mm - kh * KWN * CS;
        aoff[mt] = kh * RPS + r + lk * KP * CS;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) boff[nt] = PAIR ? lk * 2 * CP + li : (lk >> pshift) * CP + nt * 16 + li;

    wn_v4f acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = wn_v4f{0.f, 0.f, 0.f, 0.f};
    float bsum[NT];  // SWAP = 0: column sums of P = dz;  SWAP = 1: [0] = row sums of the centre-tap tile of S = dz
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

    const int hq_log = a.hq_log, hq_mask = (1 << hq_log) - 1;  // Ho / 4 rows per wave and sample, a power of two

    int n0 = bx * a.ns;
    if (n0 < a.N) load_pass(n0);
    // zero once: the halo is never overwritten (the pad after the P image only feeds unused columns)
    {
        const int total = ((a.ns * sampleS + 3) & ~3) + a.ns * sampleP + 64;
        for (int i = threadIdx.x; i < (total >> 2); i += 256) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    WN_MARK(1, false);
    [[maybe_unused]] int pass = 0;
    for (; n0 < a.N; n0 += gx * a.ns, ++pass) {
        const int ns = min(a.ns, a.N - n0);
        __syncthreads();  // zero fill landed / the previous pass's operand reads are done
        WN_MARK(2 + 4 * pass, false);
        WN_MARK(3 + 4 * pass, true);
        if (!WN_DBG(a, 2)) store_pass(ns);
        __syncthreads();
        WN_MARK(4 + 4 * pass, false);
        if (n0 + gx * a.ns < a.N && !WN_DBG(a, 4)) load_pass(n0 + gx * a.ns);  // in flight during this pass's arithmetic

        if (WN_DBG(a, 1)) continue;
        // A wave owns the image rows oh = wave, wave + 4, ...; within a row every operand address is a per-lane register
        // (row base + the lane's constant) plus a compile-time offset, and the reads of the next row are issued before
        // the matrix-core instructions of the current one (two register sets).
        auto run = [&](auto ps_c) {
            constexpr int PS = decltype(ps_c)::value;
            constexpr int PWc = WO >> PS;
            const int R = ns << hq_log;
            float av0[QX][MT], bv0[QX][NT], av1[QX][MT], bv1[QX][NT];
            auto load_row = [&](float (&av)[QX][MT], float (&bv)[QX][NT], int r) {
                const int ss = r >> hq_log, oh = wave + 4 * (r & hq_mask);
                const float* sq = s_lds + ss * sampleS + oh * RPS;
                const float* pq = p_lds + ss * sampleP + (oh >> PS) * (PWc * CP);
#pragma unroll
                for (int q = 0; q < QX; ++q) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[q][nt] = pq[boff[nt] + ((4 * KP * q) >> PS) * CP];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) av[q][mt] = sq[aoff[mt] + 4 * KP * q * CS];
                }
            };
            auto mfma_row = [&](const float (&av)[QX][MT], const float (&bv)[QX][NT]) {
#pragma unroll
                for (int q = 0; q < QX; ++q) {
                    if (SWAP) bsum[0] += av[q][MTC];
                    else {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bsum[nt] += bv[q][nt];
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][mt], bv[q][nt], acc[mt][nt], 0, 0, 0);
                }
            };
            // R is even (Ho / 4 is a power of two >= 2): the body has no branches, so the waits before each row's matrix
            // instructions count only that row's reads; the last turn re-reads its own row instead of running past the end
            // few tiles: the next row's reads go between this row's matrix instructions; many tiles (the 16 / 32-channel
            // pair, 72 accumulator registers): all reads first, which keeps the register count under the cap
            constexpr bool MIX = NTILE <= 8;
            load_row(av0, bv0, 0);
            if (MIX) __builtin_amdgcn_sched_barrier(0);
            for (int r = 0; r < R; r += 2) {
                load_row(av1, bv1, r + 1);
                if (!MIX) __builtin_amdgcn_sched_barrier(0);
                mfma_row(av0, bv0);
                if (MIX) wn_interleave<QX * MT * NT, QX * (MT + NT)>();
                __builtin_amdgcn_sched_barrier(0);
                load_row(av0, bv0, min(r + 2, R - 1));
                if (!MIX) __builtin_amdgcn_sched_barrier(0);
                mfma_row(av1, bv1);
                if (MIX) wn_interleave<QX * MT * NT, QX * (MT + NT)>();
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (pshift) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 0>{});
        WN_MARK(5 + 4 * pass, false);
    }

    // ---- sum the 4 waves as (w0 + w1) + (w2 + w3) through LDS (the images are dead): tiles travel in their register
    // layout as 16-byte pieces, wave 0 then places the sums in dw order, and all threads copy the slab out.
    float4* xch = reinterpret_cast<float4*>(smem);  // [2][NTILE + 1][64]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {  // the 4 pixel lanes of a channel
        bsum[nt] += __shfl_xor(bsum[nt], 16, 64);
        bsum[nt] += __shfl_xor(bsum[nt], 32, 64);
        if (PAIR) bsum[nt] += __shfl_xor(bsum[nt], 8, 64);  // the two pixels of a pair
    }
    auto put = [&](int slot) {
        float4* d = xch + (size_t)slot * (NTILE + 1) * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                d[(mt * NT + nt) * 64] = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
        d[NTILE * 64] = make_float4(bsum[0], NT > 1 ? bsum[NT - 1] : 0.f, 0.f, 0.f);
    };
    auto add = [&](int slot) {
        const float4* d = xch + (size_t)slot * (NTILE + 1) * 64 + lane;
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
    float* img = smem;
    if (wave == 0) {
        if constexpr (PAIR) {
            // element (kh, kw', cs) x (e, co) belongs to kernel column kw' - e: the first pixels' values are placed,
            // then the second pixels' are added (LDS operations of one wave complete in order)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = mt * 16 + 4 * lk + r;
                        const int kh = m / (KWN * CS), rem = m - kh * KWN * CS, kwp = rem / CS, cs = rem - kwp * CS;
                        const int kw = kwp - e;
                        if (m < M && (li >> 3) == e && kw >= 0 && kw <= 2) {
                            const int at = ((kh * 3 + kw) * Cin + cs) * Cout + (li & 7);
                            img[at] = e == 0 ? acc[mt][0][r] : img[at] + acc[mt][0][r];
                        }
                    }
        } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = nt * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mt * 16 + 4 * lk + r;
                    const int tap = m / CS, cs = m - tap * CS;
                    const int at = SWAP ? ((8 - tap) * Cin + n) * Cout + cs : (tap * Cin + cs) * Cout + n;
                    if (m < M && n < CP) img[at] = acc[mt][nt][r];
                }
            }
        }
        if (lk == 0) {
            if (SWAP) {
                const int cs = li - (4 * CS) % 16;
                if (cs >= 0 && cs < CS) img[9 * Cin * Cout + cs] = bsum[0];
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (nt * 16 + li < CP) img[9 * Cin * Cout + nt * 16 + li] = bsum[nt];  // PAIR: lanes 0..7 hold both pixels' sum
            }
        }
        if (lane < a.slab_floats - (9 * Cin * Cout + Cout)) img[9 * Cin * Cout + Cout + lane] = 0.f;
    }
    __syncthreads();
    float* out = a.slab + (size_t)bx * a.slab_floats;
    for (int i = threadIdx.x; i < (a.slab_floats >> 2); i += 256)
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(img)[i];
    WN_MARK(31, true);
}


// ---------------------------------------------------------------------------------------------------- stored-pixel form
// A conv behind a 2x nearest upsampling (x_up[q] = x[q >> 1]), S = dz at the upsampled resolution, P = x as stored.
// With q = p + k - 1 = 2 r + e the sum over the upsampled pixels p regroups over the stored pixels r:
//   G[ta][tb][co][ci] = sum over r of dz[2 r - 1 + (ta, tb)][co] * x[r][ci]          (a 4 x 4 window of dz per stored pixel)
//   dw[kh][kw][ci][co] = sum over eh, ew in {0, 1} of G[2 - kh + eh][2 - kw + ew][co][ci]
// 16 instead of 4 x 9 products per stored pixel.  The GEMM rows are (ta, tb, co): wave `ta` owns the rows of its window
// row for every pixel of the pass (the waves split the rows of the result, not the pixels, so their sums need no
// exchange), the 4 reduction slots of an instruction are 4 consecutive stored pixels of an image row.  The four window
// positions with ta, tb in {1, 2} tile the dz image exactly once: their operand values, summed per lane, are the bias
// gradient.  Pixel strides of 24 floats for a 16-channel S and 48 for a 32-channel P keep the two pixel slots that share
// an LDS cycle on different banks.
constexpr int wn_sp_ps(int CS) { return CS == 16 ? 24 : CS; }
constexpr int wn_sp_pps(int CP) { return CP == 32 ? 48 : CP; }

template <int CS, int CP, int WO>
__device__ __forceinline__ void wn_body_sp(const WnArgs& a, const int bx, const int gx) {
    constexpr int TPW = CS == 3 ? 1 : CS / 4;                  // 16-row tiles of a wave: (tb, co) of its window row
    constexpr int NT = (CP + 15) / 16;
    constexpr int Ws = WO / 2, QX = Ws / 4, SW = WO + 2;
    constexpr int PS = wn_sp_ps(CS), PPS = wn_sp_pps(CP);
    constexpr int Cin = CP, Cout = CS;
    static_assert(QX >= 1 && (CS == 3 || CS % 4 == 0), "stored rows hold whole pixel quads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int Ho = a.Ho, Hs = a.Hs;
    const int sampleS = (Ho + 2) * SW * PS, sampleP = Hs * Ws * PPS;
    float* s_lds = smem;
    float* p_lds = smem + ((a.ns * sampleS + 3) & ~3);
    const int s_sample = Ho * WO * CS, p_sample = Hs * Ws * CP;
    const bool s_act = a.zmode == 1;
    const FDiv f_shw = fd_make(Ho * WO);

    WN_MARK(0, false);
    WnRegs R;
    auto load_pass = [&](int n0) {
        const int ns = min(a.ns, a.N - n0);
        const int su = (ns * s_sample) >> 2, pu = (ns * p_sample) >> 2;
        const size_t sb = (size_t)n0 * s_sample, pb = (size_t)n0 * p_sample;
#pragma unroll
        for (int i = 0; i < kWnUnits; ++i) {
            const int u = min((int)threadIdx.x + i * 256, su - 1);
            R.sv[i] = *reinterpret_cast<const float4*>(a.dy + sb + 4 * (size_t)u);
            R.sy[i] = s_act ? *reinterpret_cast<const float4*>(a.zy + sb + 4 * (size_t)u) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
#pragma unroll
        for (int i = 0; i < kWnUnits; ++i) {
            const int u = min((int)threadIdx.x + i * 256, pu - 1);
            R.pv[i] = *reinterpret_cast<const float4*>(a.x + pb + 4 * (size_t)u);
        }
    };
    constexpr int SEL = CS % 4 == 0 ? 1 : 4;
    int s_at[kWnUnits][SEL], p_at[kWnUnits];
#pragma unroll
    for (int i = 0; i < kWnUnits; ++i) {
        const int e = 4 * ((int)threadIdx.x + i * 256);
#pragma unroll
        for (int t = 0; t < SEL; ++t) {
            const int p = (e + t) / CS, c = (e + t) - p * CS;
            const int ss = fd(p, f_shw), pp = p - ss * (Ho * WO);
            const int row = pp / WO, col = pp - row * WO;
            s_at[i][t] = ss * sampleS + ((row + 1) * SW + col + 1) * PS + c;
        }
        p_at[i] = (e / CP) * PPS + e % CP;
    }
    auto store_pass = [&](int ns) {
        const int su = (ns * s_sample) >> 2, pu = (ns * p_sample) >> 2;
#pragma unroll
        for (int i = 0; i < kWnUnits; ++i) {
            if ((int)threadIdx.x + i * 256 < su) {
                float4 v = R.sv[i];
                if (s_act) {
                    v.x *= act_grad_q(R.sy[i].x, a.zk); v.y *= act_grad_q(R.sy[i].y, a.zk);
                    v.z *= act_grad_q(R.sy[i].z, a.zk); v.w *= act_grad_q(R.sy[i].w, a.zk);
                }
                if (CS % 4 == 0) {
                    *reinterpret_cast<float4*>(s_lds + s_at[i][0]) = v;
                } else {
                    s_lds[s_at[i][0]] = v.x; s_lds[s_at[i][1 % SEL]] = v.y;
                    s_lds[s_at[i][2 % SEL]] = v.z; s_lds[s_at[i][3 % SEL]] = v.w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kWnUnits; ++i)
            if ((int)threadIdx.x + i * 256 < pu) *reinterpret_cast<float4*>(p_lds + p_at[i]) = R.pv[i];
    };

    // row t * 16 + li of the wave's block is (tb, co) = divmod(t * 16 + li, CS) (3 channels: rows 12..15 are unused and
    // read finite neighbours); reduction slot lk is stored pixel 4 q + lk of the row, whose window starts at dz column
    // 2 (4 q + lk) of the halo image
    int aoff[TPW], boff[NT];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int m = t * 16 + li, tb = m / CS, cs = m - tb * CS;
        aoff[t] = wave * SW * PS + tb * PS + cs + lk * 2 * PS;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) boff[nt] = lk * PPS + nt * 16 + li;

    wn_v4f acc[TPW][NT];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = wn_v4f{0.f, 0.f, 0.f, 0.f};
    float bs = 0.f;
    const int hs_log = a.hq_log, hs_mask = (1 << hs_log) - 1;  // Hs is a power of two

    int n0 = bx * a.ns;
    if (n0 < a.N) load_pass(n0);
    {
        const int total = ((a.ns * sampleS + 3) & ~3) + a.ns * sampleP + 64;
        for (int i = threadIdx.x; i < (total >> 2); i += 256) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    WN_MARK(1, false);
    [[maybe_unused]] int pass = 0;
    for (; n0 < a.N; n0 += gx * a.ns, ++pass) {
        const int ns = min(a.ns, a.N - n0);
        __syncthreads();
        WN_MARK(2 + 4 * pass, false);
        WN_MARK(3 + 4 * pass, true);
        if (!WN_DBG(a, 2)) store_pass(ns);
        __syncthreads();
        WN_MARK(4 + 4 * pass, false);
        if (n0 + gx * a.ns < a.N && !WN_DBG(a, 4)) load_pass(n0 + gx * a.ns);
        if (WN_DBG(a, 1)) continue;
        // two stored rows per turn, the next pair's reads issued ahead of the current pair's matrix instructions
        const int R = ns << hs_log;  // a multiple of 4
        float av0[2][QX][TPW], bv0[2][QX][NT], av1[2][QX][TPW], bv1[2][QX][NT];
        auto load_rows = [&](float (&av)[2][QX][TPW], float (&bv)[2][QX][NT], int r0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = r0 + j, ss = r >> hs_log, oh = r & hs_mask;
                const float* sq = s_lds + ss * sampleS + 2 * oh * (SW * PS);
                const float* pq = p_lds + ss * sampleP + oh * (Ws * PPS);
#pragma unroll
                for (int q = 0; q < QX; ++q) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[j][q][nt] = pq[boff[nt] + 4 * q * PPS];
#pragma unroll
                    for (int t = 0; t < TPW; ++t) av[j][q][t] = sq[aoff[t] + 8 * q * PS];
                }
            }
        };
        auto mfma_rows = [&](const float (&av)[2][QX][TPW], const float (&bv)[2][QX][NT]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < QX; ++q) {
                    if (CS == 16) bs += av[j][q][1 % TPW] + av[j][q][2 % TPW];
                    else if (CS == 8) bs += li >= 8 ? av[j][q][0] : av[j][q][1 % TPW];
                    else bs += av[j][q][0];
#pragma unroll
                    for (int t = 0; t < TPW; ++t)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][q][t], bv[j][q][nt], acc[t][nt], 0, 0, 0);
                }
        };
        load_rows(av0, bv0, 0);
        __builtin_amdgcn_sched_barrier(0);
        for (int r = 0; r < R; r += 4) {
            load_rows(av1, bv1, r + 2);
            mfma_rows(av0, bv0);
            wn_interleave<2 * QX * TPW * NT, 2 * QX * (TPW + NT)>();
            __builtin_amdgcn_sched_barrier(0);
            load_rows(av0, bv0, min(r + 4, R - 2));
            mfma_rows(av1, bv1);
            wn_interleave<2 * QX * TPW * NT, 2 * QX * (TPW + NT)>();
            __builtin_amdgcn_sched_barrier(0);
        }
        WN_MARK(5 + 4 * pass, false);
    }

    // ---- G into LDS as [window position][ci][co] (the images are dead), then every thread folds four of its values into
    // an element of dw and writes it to the slab; the bias sums of the two inner window rows meet in LDS as well
    bs += __shfl_xor(bs, 16, 64);
    bs += __shfl_xor(bs, 32, 64);
    __syncthreads();
    float* gimg = smem;
    float* bimg = smem + 16 * CS * CP;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + li;
            if (CS % 4 == 0) {
                const int m = t * 16 + 4 * lk, tb = m / CS, cs = m - tb * CS;
                if (n < CP)
                    *reinterpret_cast<float4*>(gimg + ((wave * 4 + tb) * CP + n) * CS + cs) =
                        make_float4(acc[t][nt][0], acc[t][nt][1], acc[t][nt][2], acc[t][nt][3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = t * 16 + 4 * lk + r, tb = m / CS, cs = m - tb * CS;
                    if (tb < 4 && n < CP) gimg[((wave * 4 + tb) * CP + n) * CS + cs] = acc[t][nt][r];
                }
            }
        }
    if (lk == 0 && (wave == 1 || wave == 2)) bimg[(wave - 1) * 16 + li] = bs;
    __syncthreads();
    float* out = a.slab + (size_t)bx * a.slab_floats;
    constexpr int NW = 9 * Cin * Cout;
    for (int e = threadIdx.x; e < NW; e += 256) {
        const int co = e % Cout, ci = (e / Cout) % Cin, tap = e / (Cin * Cout);
        const int kh = tap / 3, kw = tap - 3 * kh;
        const float* g0 = gimg + (((2 - kh) * 4 + (2 - kw)) * CP + ci) * CS + co;
        out[e] = (g0[0] + g0[CP * CS]) + (g0[4 * CP * CS] + g0[5 * CP * CS]);
    }
    if ((int)threadIdx.x < Cout) {
        const int c = threadIdx.x;
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < 2; ++wv)
#pragma unroll
            for (int l = 0; l < 16; ++l) {
                // lanes whose summed rows were the window columns tb in {1, 2} of channel c
                const bool mine = CS == 16 ? l == c : (CS == 8 ? (l & 7) == c : (l >= 3 && l < 9 && l % 3 == c));
                if (mine) t += bimg[wv * 16 + l];
            }
        out[NW + c] = t;
    } else if ((int)threadIdx.x - Cout < a.slab_floats - (NW + Cout)) {
        out[NW + threadIdx.x] = 0.f;
    }
    WN_MARK(31, true);
}

constexpr int kWnMaxJobs = 8;
struct WnMultiArgs {
    int n_jobs;
    int first[kWnMaxJobs + 1];
    int shape[kWnMaxJobs];  // 0: 3 / 8 channels on 32-wide images, 1: 8 / 16 on 16-wide, 2: 16 / 32 on 8-wide;  + 4 when S = dz (SWAP),
                            // + 8 for S = dz in the stored-pixel form of an upsampled layer
    WnArgs job[kWnMaxJobs];
};

__global__ __launch_bounds__(256, 2) void wgrad_nhwc_multi(WnMultiArgs m) {
    int j = 0;
    while (j + 1 < m.n_jobs && (int)blockIdx.x >= m.first[j + 1]) ++j;
    const int bx = blockIdx.x - m.first[j], gx = m.first[j + 1] - m.first[j];
    // The job's arguments are read from the kernel-argument segment at a wave-uniform offset (scalar loads); indexing
    // the by-value table directly would make the compiler copy all of it to per-lane scratch memory first.
    typedef const __attribute__((address_space(4))) WnMultiArgs* KArgs;
    const KArgs mp = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    WnArgs a;
    __builtin_memcpy(&a, &mp->job[j], sizeof(WnArgs));
    switch (m.shape[j]) {
        case 0: wn_body<3, 8, 0, 32, true>(a, bx, gx); break;
        case 1: wn_body<8, 16, 0, 16>(a, bx, gx); break;
        case 2: wn_body<16, 32, 0, 8>(a, bx, gx); break;
        case 4: wn_body<3, 8, 1, 32>(a, bx, gx); break;
        case 5: wn_body<8, 16, 1, 16>(a, bx, gx); break;
        case 6: wn_body<16, 32, 1, 8>(a, bx, gx); break;
        case 8: wn_body_sp<3, 8, 32>(a, bx, gx); break;
        case 9: wn_body_sp<8, 16, 16>(a, bx, gx); break;
        default: wn_body_sp<16, 32, 8>(a, bx, gx); break;
    }
}

// dw / db (+)= the slabs of all workgroups, summed in a fixed order: 16 groups of a block split the slabs, 64 lanes of
// a group own 64 consecutive slab elements (coalesced), the groups meet in LDS.
struct WnReduceJob {
    const float* slab;
    float* dw;
    float* db;
    int slab_floats, n_dw, n_db, n_split, accumulate;
};
struct WnReduceArgs {
    int n_jobs;
    int first[kWnMaxJobs + 1];
    WnReduceJob job[kWnMaxJobs];
};
__global__ __launch_bounds__(1024) void wn_reduce_multi(WnReduceArgs m) {
    __shared__ float red[16][64];
    int j = 0;
    while (j + 1 < m.n_jobs && (int)blockIdx.x >= m.first[j + 1]) ++j;
    const WnReduceJob& a = m.job[j];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = (blockIdx.x - m.first[j]) * 64 + lane;
    float s = 0.f;
    if (e < a.n_dw + a.n_db) {
        int sp = grp;
        for (; sp + 7 * 16 < a.n_split; sp += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a.slab[(size_t)(sp + u * 16) * a.slab_floats + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; sp < a.n_split; sp += 16) s += a.slab[(size_t)sp * a.slab_floats + e];
    }
    red[grp][lane] = s;
    __syncthreads();
    if (grp != 0 || e >= a.n_dw + a.n_db) return;
    float* o = e < a.n_dw ? a.dw + e : (a.db != nullptr ? a.db + (e - a.n_dw) : nullptr);
    if (o == nullptr) return;
    float t = a.accumulate ? *o : 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][lane];
    *o = t;
}

// ---------------------------------------------------------------------------------------------------- host side
struct WnPlan {
    bool ok;
    int shape, ns, gx, slab_floats, hq_log;
    size_t lds_bytes;
};

static WnPlan plan_wn(const sv_wgrad_batch_job& q) {
    WnPlan p{};
    const sv_conv_geom& g = q.g;
    if (g.KH != 3 || g.KW != 3 || g.SH != 1 || g.SW != 1 || g.up_h != g.up_w || (g.up_h != 1 && g.up_h != 2)) return p;
    if (g.Ho != g.H || g.Wo != g.W || g.pad_t != 1 || g.pad_l != 1) return p;
    const int up = g.up_h, Ho = g.H, Wo = g.W, Hs = Ho / up, Ws = Wo / up;
    const int zmode = q.idx != nullptr ? 2 : 1;
    if ((Wo & 3) || (zmode == 2 && ((Ho | Wo) & 1))) return p;
    int swap, cs, cp;
    if (g.Cin < g.Cout) { swap = 0; cs = g.Cin; cp = g.Cout; }
    else { swap = 1; cs = g.Cout; cp = g.Cin; }
    int sh;
    if (cs == 3 && cp == 8) sh = 0;
    else if (cs == 8 && cp == 16) sh = 1;
    else if (cs == 16 && cp == 32) sh = 2;
    else return p;
    if (swap == 0 && up != 1) return p;     // an upsampled x is only read as P
    if (swap == 1 && zmode == 2) return p;  // a pooled dz is only scattered as P
    const int mt = (9 * cs + 15) / 16, nt = (cp + 15) / 16, ntile = mt * nt;
    if (Wo != (sh == 0 ? 32 : (sh == 1 ? 16 : 8))) return p;  // the row loops are unrolled for these widths
    const int hq = Ho >> 2;                                    // image rows per wave
    if ((Ho & 3) || hq < 2 || (hq & (hq - 1))) return p;       // an even number of rows per wave
    p.hq_log = __builtin_ctz(hq);
    // an upsampled layer in the stored-pixel form: the stored image holds whole groups of 4 rows, a power of two of them
    const bool sp = swap == 1 && up == 2 && opt(OPT_WGRAD_STORED) != 0 && (Hs & 3) == 0 && (Hs & (Hs - 1)) == 0;
    if (sp) p.hq_log = __builtin_ctz(Hs);
    const int64_t sampleS = sp ? (int64_t)(Ho + 2) * (Wo + 2) * wn_sp_ps(cs) : (int64_t)(Ho + 2) * wn_rps(cs, Wo);
    const int64_t sampleP = sp ? (int64_t)Hs * Ws * wn_sp_pps(cp) : (swap ? (int64_t)Hs * Ws * cp : (int64_t)Ho * Wo * cp);
    const int64_t s_units = (int64_t)Ho * Wo * cs / 4;
    const int64_t p_units = (swap ? (int64_t)Hs * Ws * cp : (zmode == 2 ? (int64_t)(Ho / 2) * (Wo / 2) * cp : (int64_t)Ho * Wo * cp)) / 4;
    auto lds = [&](int ns) { return (size_t)((((int64_t)ns * sampleS + 3) & ~(int64_t)3) + ns * sampleP + 64) * sizeof(float); };
    // 128 workgroups per layer measured best at batch 1024 (768 for the six layers of the default model: fewer leave CUs
    // idle, more pay the fold and slab traffic again); a caller asking for fewer gets fewer
    int gx = q.blocks > 0 && q.blocks <= 512 ? q.blocks : 128;
    if (gx > g.N) gx = g.N;
    int ns = 1;
    while (ns < 8 && lds(ns + 1) <= 64 * 1024 && (ns + 1) * s_units <= 256 * kWnUnits && (ns + 1) * p_units <= 256 * kWnUnits &&
           (int64_t)(ns + 1) * gx * 2 <= g.N)  // at least two passes per workgroup, so loads overlap arithmetic
        ++ns;
    if (lds(ns) > 64 * 1024 || ns * s_units > 256 * kWnUnits || ns * p_units > 256 * kWnUnits) return p;
    if ((int64_t)gx * ns > g.N) gx = (g.N + ns - 1) / ns;
    p.shape = sh + (sp ? 8 : 4 * swap);
    p.ns = ns;
    p.gx = gx;
    p.slab_floats = (9 * g.Cin * g.Cout + g.Cout + 3) & ~3;
    size_t need = lds(ns);
    const size_t xch = sp ? (size_t)(16 * cs * cp + 32) * sizeof(float) : (size_t)2 * (ntile + 1) * 64 * sizeof(float4);
    if (xch > need) need = xch;
    if ((size_t)p.slab_floats * sizeof(float) > need) need = (size_t)p.slab_floats * sizeof(float);
    p.lds_bytes = (need + 15) & ~(size_t)15;
    p.ok = true;
    return p;
}

// Whether launch_wgrad_nhwc_batch takes these jobs.
bool wgrad_nhwc_batch_ok(const sv_wgrad_batch_job* jobs, int n_jobs) {
    if (n_jobs < 1 || n_jobs > kWnMaxJobs) return false;
    if (opt(OPT_WGRAD_NHWC) == 0) return false;
    for (int j = 0; j < n_jobs; ++j) {
        const WnPlan p = plan_wn(jobs[j]);
        if (!p.ok) return false;
        if (jobs[j].workspace == nullptr || jobs[j].workspace_bytes < (size_t)p.gx * p.slab_floats * sizeof(float)) return false;
    }
    return true;
}

// 1: some job is outside this family or its workspace is too small for the slabs (the caller takes the planar kernels);
// 0: launched;  < 0: error.  slabs_out != nullptr: the slabs are NOT summed here; slabs_out[j] describes job j's slabs
// for whoever consumes them (the optimizer launch that reads its gradients as sums of slabs).
int launch_wgrad_nhwc_batch(const sv_wgrad_batch_job* jobs, int n_jobs, hipStream_t st, sv_grad_slabs* slabs_out) {
    if (!wgrad_nhwc_batch_ok(jobs, n_jobs)) return 1;
    WnPlan plans[kWnMaxJobs];
    for (int j = 0; j < n_jobs; ++j) plans[j] = plan_wn(jobs[j]);
    // longest-lived workgroups first (matrix-core work per workgroup), so the launch does not end on a long one
    int ord[kWnMaxJobs];
    auto cost = [&](int j) {
        const sv_conv_geom& g = jobs[j].g;
        const int cs = g.Cin < g.Cout ? g.Cin : g.Cout, cp = g.Cin < g.Cout ? g.Cout : g.Cin;
        return (double)g.H * g.W * ((9 * cs + 15) / 16) * ((cp + 15) / 16) * g.N / plans[j].gx;
    };
    for (int j = 0; j < n_jobs; ++j) ord[j] = j;
    for (int a = 1; a < n_jobs; ++a)
        for (int b = a; b > 0 && cost(ord[b]) > cost(ord[b - 1]); --b) { const int t = ord[b]; ord[b] = ord[b - 1]; ord[b - 1] = t; }
    WnMultiArgs m{};
    WnReduceArgs rm{};
    m.n_jobs = rm.n_jobs = n_jobs;
    int blocks = 0, rblocks = 0;
    size_t lds = 0;
    for (int s = 0; s < n_jobs; ++s) {
        const sv_wgrad_batch_job& q = jobs[ord[s]];
        const WnPlan& p = plans[ord[s]];
        const sv_conv_geom& g = q.g;
        WnArgs& a = m.job[s];
        a.x = (const float*)q.x; a.dy = (const float*)q.dy; a.zy = (const float*)q.y; a.zidx = q.idx; a.slab = (float*)q.workspace;
        a.N = g.N; a.Ho = g.H; a.Wo = g.W; a.up = g.up_h; a.Hs = g.H / g.up_h; a.Ws = g.W / g.up_w;
        a.zmode = q.idx != nullptr ? 2 : 1; a.with_bias = q.db != nullptr; a.ns = p.ns; a.slab_floats = p.slab_floats; a.hq_log = p.hq_log;
        a.zk = act_grad_coef(q.act, q.alpha);
#ifdef SV_TRACE
        if (const char* e = std::getenv("SPECTRUM_WGRAD_DBG")) a.dbg = atoi(e);
#endif
        m.shape[s] = p.shape;
        m.first[s] = blocks;
        blocks += p.gx;
        if (p.lds_bytes > lds) lds = p.lds_bytes;
        WnReduceJob& r = rm.job[s];
        r.slab = (const float*)q.workspace; r.dw = (float*)q.dw; r.db = (float*)q.db; r.slab_floats = p.slab_floats;
        r.n_dw = 9 * g.Cin * g.Cout; r.n_db = g.Cout; r.n_split = p.gx; r.accumulate = q.accumulate;
        rm.first[s] = rblocks;
        rblocks += (r.n_dw + r.n_db + 63) / 64;
    }
    m.first[n_jobs] = blocks;
    rm.first[n_jobs] = rblocks;
    hipLaunchKernelGGL(wgrad_nhwc_multi, dim3((unsigned)blocks), dim3(256), lds, st, m);
    int rc = check_launch("wgrad_nhwc_multi");
    if (rc) return rc;
    if (slabs_out != nullptr) {
        for (int j = 0; j < n_jobs; ++j) {
            const sv_conv_geom& g = jobs[j].g;
            slabs_out[j] = sv_grad_slabs{jobs[j].workspace, jobs[j].dw, jobs[j].db, plans[j].slab_floats, plans[j].gx,
                                         9 * g.Cin * g.Cout, g.Cout};
        }
        return SV_OK;
    }
#ifdef SV_TRACE
    if (const char* e = std::getenv("SPECTRUM_WGRAD_DBG"))  // bit 8: the launch above without the sum of its slabs
        if (atoi(e) & 8) return SV_OK;
#endif
    hipLaunchKernelGGL(wn_reduce_multi, dim3((unsigned)rblocks), dim3(1024), 0, st, rm);
    return check_launch("wn_reduce_multi");
}

}  // namespace sv

#ifdef SV_TRACE
extern "C" int sv_debug_set_wn_trace(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(sv::g_wn_trace), &buf, sizeof(buf));
}
#endif
