// Device code of the channel-last matrix-core input-gradient kernels (dgrad_nhwc.hip), shared with the launch that runs
// them behind the fused forward pass (net_fused.hip).
#pragma once
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "tile_common.h"

namespace sv {

typedef float dn_v4f __attribute__((ext_vector_type(4)));

// Order of one turn of the tile loop: NM matrix instructions with the NR LDS reads of the NEXT tile's operands placed one
// behind each of the first ones (the reads issue in the shadow of the matrix pipe).
template <int NM, int NR, int... I>
__device__ __forceinline__ void dn_interleave_seq(std::integer_sequence<int, I...>) {
    ((__builtin_amdgcn_sched_group_barrier(0x008, 1, 0), __builtin_amdgcn_sched_group_barrier(0x100, I < NR ? 1 : 0, 0)), ...);
}
template <int NM, int NR>
__device__ __forceinline__ void dn_interleave() {
    static_assert(NR <= NM, "one read per matrix instruction at most");
    dn_interleave_seq<NM, NR>(std::make_integer_sequence<int, NM>{});
}

#ifdef SV_TRACE
// diagnostic builds: thread 0 of every workgroup stores the 100 MHz wall clock at phase marks, 16 slots per workgroup
static __device__ unsigned long long* g_dn_trace = nullptr;  // per translation unit
__device__ __forceinline__ void dn_mark(int slot, bool drain) {
    if (g_dn_trace == nullptr) return;
    if (drain) __builtin_amdgcn_s_waitcnt(0);  // the phase's loads have landed
    if (threadIdx.x == 0 && slot < 16) g_dn_trace[(size_t)blockIdx.x * 16 + slot] = wall_clock64();
}
#define DN_MARK(slot, drain) dn_mark(slot, drain)
#else
#define DN_MARK(slot, drain)
#endif

struct DnArgs {
    const float* dy;      // zmode 1: [N][H][W][CZ];  zmode 2: [N][H/2][W/2][CZ]
    const float* zy;      // activation output, extents of dy
    const uint8_t* zidx;  // zmode 2: argmax slot of every pooling window
    const float* w;       // [3][3][CX][CZ]: the forward conv's HWIO filters (CX input, CZ output channels)
    float* dx;            // up == 1: [N][H][W][CX];  up == 2: [N][H/2][W/2][CX], the sums over 2x2 windows
    int N, H, up, zmode, ns, tps_log;  // tps_log: log2 of the (H / 2) * (W / 8) pixel tiles per sample
    ActGradCoef zk;
};

// 16-byte pieces per thread and pass held in registers between load and store (two samples of 3-channel dz a pass were
// measured: 13.6 against 11.9 us, its scalar staging then has nothing to hide behind)
// (the 32 x 32 image of 8-channel dz is 2 048 pieces: eight per thread)
constexpr int dn_units(int CZ, int WO) { return (CZ == 8 && WO == 32) ? 8 : 4; }

// The LDS image of dz: floats from one pixel to the next (PS) and from one row to the next (RP).  The 16 lanes that one
// LDS cycle of an operand read serves are 4 x 4 (16-byte reads), 2 x 16 (8- and 4-byte reads) (pixel, channel group)
// pairs of a 2 x 8 pixel patch; with the channels of a pixel packed (PS = CZ) and rows packed (RP = (W + 2) PS) they
// fall on a half or a quarter of the banks.  The strides below are the smallest that spread every such group over all
// 64 banks (found by enumerating the lane groups of ds_read_b128 / _b64 / _b32 over candidate strides).
constexpr int dn_pad_to(int base, int mod, int res) { return base + ((res - base % mod) + mod) % mod; }
constexpr int dn_ps(int CZ, int MODE) {
    return (CZ == 8 && MODE == 1) ? 10 : (CZ == 16 && MODE == 2) ? 20 : (CZ == 32 && MODE == 0) ? 40 : CZ;
}
constexpr int dn_rp(int CZ, int WO, int MODE) {
    const int base = (WO + 2) * dn_ps(CZ, MODE);
    if (CZ == 3) return MODE == 1 ? dn_pad_to(base, 16, 8) : base;
    if (CZ == 8) return MODE == 1 ? dn_pad_to(base, 32, 16) : dn_pad_to(base, 8, 4);
    if (CZ == 16) return MODE == 2 ? dn_pad_to(base, 64, 0) : dn_pad_to(base, 16, 8);
    if (CZ == 32) return MODE == 0 ? dn_pad_to(base, 64, 0) : base;
    return base;
}

__device__ __forceinline__ float dn_dpp_xor1(float v) {  // value of lane ^ 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ float dn_dpp_ror8(float v) {  // value of lane ^ 8 (rotation by 8 inside a row of 16 lanes)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
}

// PH (a conv behind a 2x nearest upsampling, a.up == 2): the result tile holds 16 STORED pixels instead of 16 pixels of
// the upsampled image.  The four upsampled pixels above a stored pixel (r, c) read dz rows 2r - 1 .. 2r + 2 and columns
// 2c - 1 .. 2c + 2 between them, each dz pixel of that 4 x 4 window with the SUM of the filter taps that connect it to
// one of the four (window row t reads kernel rows {2 - t, 3 - t} within 0..2, columns alike), so
//   dx[r, c][ci] = sum over the 4 x 4 window and co of dz[2r - 1 + ta, 2c - 1 + tb][co] * wsum[ta][tb][ci][co]:
// 16 window taps per stored pixel instead of 4 x 9 per stored pixel, and no 2 x 2 sums afterwards.
// MODE 2 (8 input channels, where the filters fill only half of a 16-row A operand): rows 8..15 hold the same filters
// shifted by one column, so a result column is a PAIR of horizontally adjacent pixels (even column in rows 0..7, odd
// column in rows 8..15) reading a 3 x 4 window of dz between them: 12 window taps per pixel pair instead of 2 x 9.
// The workgroup's samples are [nb, ne) in passes of a.ns, `step` samples from one pass to the next; `wg` names it in traces.
template <int CZ, int CX, int WO, int MODE>
__device__ __forceinline__ void dn_body(const DnArgs& a, const int nb, const int ne, const int step, [[maybe_unused]] const int wg) {
    constexpr bool PH = MODE == 1, WD = MODE == 2;
    static_assert(!WD || CX == 8, "the pair form fills the rows an 8-channel filter leaves empty");
    constexpr int MT = (CX + 15) / 16;
    constexpr int VEC = CZ == 3 ? 1 : (CZ == 8 ? 2 : 4);          // dz channels per lane and read
    constexpr int NH = CZ == 3 ? 1 : CZ / (4 * VEC);              // reads per tap
    constexpr int TW = (PH || WD) ? 4 : 3, TAPS = (PH ? 4 : 3) * TW;  // window of dz pixels a result column reads
    constexpr int NSTEP = CZ == 3 ? (3 * TAPS + 3) / 4 : TAPS * NH;  // reads per pixel tile
    constexpr int TX = (PH || WD) ? WO / 16 : WO / 8;
    constexpr int PS = dn_ps(CZ, MODE), RP = dn_rp(CZ, WO, MODE);
    static_assert(CZ == 3 || (PS % VEC == 0 && RP % 4 == 0), "operand reads and staging stores stay aligned");
    static_assert(!WD || CZ != 3, "pair form: channel counts that are multiples of 4");
    constexpr bool PING = NSTEP * VEC <= 36;                      // operand registers of two tiles fit
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // the thread index behind an opaque copy: what a layer derives from it is then not loop-invariant for the chain
    // kernel's layer loop, which would otherwise hoist every layer's per-lane values in front of the loop and spill them
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int pr = li >> 3, pc = li & 7;
    const int H = a.H;
    const int sampleZ = (H + 2) * RP;

    // ---- staging of dz: source extents, registers of the pass in flight
    const int zmode = a.zmode;
    const int s_h = zmode == 2 ? H >> 1 : H, s_w = zmode == 2 ? WO >> 1 : WO;
    const int s_sample = s_h * s_w * CZ;
    const FDiv f_hw = fd_make(s_h * s_w), f_w = fd_make(s_w);
    constexpr int kDnUnits = dn_units(CZ, WO);
    float4 rv[kDnUnits], ry[kDnUnits];
    uint32_t rk[kDnUnits];
    auto load_pass = [&](int n0) {
        const int ns = min(a.ns, ne - n0);
        const int su = (ns * s_sample) >> 2;
        const size_t sb = (size_t)n0 * s_sample;
#pragma unroll
        for (int i = 0; i < kDnUnits; ++i) {
            const int u = min(tid + i * 256, su - 1);
            rv[i] = *reinterpret_cast<const float4*>(a.dy + sb + 4 * (size_t)u);
            ry[i] = *reinterpret_cast<const float4*>(a.zy + sb + 4 * (size_t)u);
            rk[i] = zmode == 2 ? *reinterpret_cast<const uint32_t*>(a.zidx + sb + 4 * (size_t)u) : 0u;
        }
    };
    auto store_pass = [&](int ns) {
        const int su = (ns * s_sample) >> 2;
#pragma unroll
        for (int i = 0; i < kDnUnits; ++i) {
            const int u = tid + i * 256;
            if (u >= su) continue;
            float4 v = rv[i];
            v.x *= act_grad_q(ry[i].x, a.zk); v.y *= act_grad_q(ry[i].y, a.zk);
            v.z *= act_grad_q(ry[i].z, a.zk); v.w *= act_grad_q(ry[i].w, a.zk);
            const int e = 4 * u;
            if (CZ % 4 == 0) {
                const int p = e / CZ, c = e - p * CZ;
                const int ss = fd(p, f_hw), pp = p - ss * (s_h * s_w);
                const int row = fd(pp, f_w), col = pp - row * s_w;
                auto put = [](float* q, const float4& o) {
                    if constexpr (PS % 4 == 0) {
                        *reinterpret_cast<float4*>(q) = o;
                    } else {  // pixels start at multiples of 8 bytes only
                        *reinterpret_cast<float2*>(q) = make_float2(o.x, o.y);
                        *reinterpret_cast<float2*>(q + 2) = make_float2(o.z, o.w);
                    }
                };
                if (zmode != 2) {
                    put(smem + ss * sampleZ + (row + 1) * RP + (col + 1) * PS + c, v);
                } else {
                    float* q = smem + ss * sampleZ + (2 * row + 1) * RP + (2 * col + 1) * PS + c;
                    const uint32_t k = rk[i];
                    const uint32_t k0 = k & 0xff, k1 = (k >> 8) & 0xff, k2 = (k >> 16) & 0xff, k3 = k >> 24;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float4 o = make_float4(k0 == (uint32_t)s ? v.x : 0.f, k1 == (uint32_t)s ? v.y : 0.f,
                                                     k2 == (uint32_t)s ? v.z : 0.f, k3 == (uint32_t)s ? v.w : 0.f);
                        put(q + (s >> 1) * RP + (s & 1) * PS, o);
                    }
                }
            } else {  // 3 channels, never pooled here: the 4 floats of a piece belong to 1 or 2 pixels
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int p = (e + t) / CZ, c = (e + t) - p * CZ;
                    const int ss = fd(p, f_hw), pp = p - ss * (s_h * s_w);
                    const int row = fd(pp, f_w), col = pp - row * s_w;
                    smem[ss * sampleZ + (row + 1) * RP + (col + 1) * PS + c] = vv[t];
                }
            }
        }
    };

    // ---- the filters as A operands: row m = mt * 16 + li is input channel ci, reduction slot lk of read (tap, h) and
    // element j is output channel h * 4 VEC + VEC * lk + j of tap (a, b) = mirrored (kh, kw)
    DN_MARK(0, false);
    // Where a lane's operand values are many scattered words (WLDS), the filter tensor comes in as 16-byte pieces (one
    // round trip for the workgroup) behind the dz image and the lane picks its values out of LDS; the two instantiations
    // with the fewest values per lane read them from global memory directly (measured 1 us faster there).
    constexpr bool WLDS = !((CZ == 16 && CX == 32) || (CZ == 8 && CX == 16));
    int n0 = nb;
    if (n0 < ne) load_pass(n0);
    const int zimg4 = (a.ns * sampleZ + 3 + 16) >> 2;  // 16-byte pieces of the dz images (and the pad behind them)
    const float* wl = WLDS ? smem + 4 * zimg4 : a.w;
    if constexpr (!WLDS) {
        for (int i = tid; i < zimg4; i += 256) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        constexpr int W4 = 9 * CX * CZ / 4;
        static_assert(9 * CX * CZ % 4 == 0, "the filters are copied as 16-byte pieces");
        float4 wv[(W4 + 255) / 256];
#pragma unroll
        for (int i = 0; i < (W4 + 255) / 256; ++i)
            wv[i] = reinterpret_cast<const float4*>(a.w)[min(tid + i * 256, W4 - 1)];
        for (int i = tid; i < zimg4; i += 256) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < (W4 + 255) / 256; ++i)
            if (tid + i * 256 < W4) reinterpret_cast<float4*>(smem)[zimg4 + tid + i * 256] = wv[i];
    }
    if constexpr (WLDS) __syncthreads();
    float wreg[NSTEP][VEC][MT];
    int zoff[CZ == 3 ? NSTEP : 1];  // 3 channels: reduction index kk = 4 s + lk walks (a, b, co) flattened
    // the filter value that connects window position (ta, tb) of dz with input channel ci: one mirrored tap, or (PH) the
    // sum of the taps named above
    auto wtap = [&](int ta, int tb, int row, int co) {
        if (WD) {  // row = (pixel of the pair, channel): the odd pixel's window starts one column to the right
            const int e = row >> 3, ci = row & 7, t = tb - e;
            return (t >= 0 && t <= 2) ? wl[(((2 - ta) * 3 + (2 - t)) * CX + ci) * CZ + co] : 0.f;
        }
        const int ci = row;
        if (!PH) return wl[(((2 - ta) * 3 + (2 - tb)) * CX + ci) * CZ + co];
        // kernel rows {2 - ta, 3 - ta} and columns {2 - tb, 3 - tb} inside 0..2: four loads at clamped indices, the ones
        // outside weighted 0 (a loop over the nine taps with a condition each costs nine predicated loads per value)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kh = 2 - ta + i, kw = 2 - tb + j;
                const bool in = kh >= 0 && kh <= 2 && kw >= 0 && kw <= 2;
                const float v = wl[((min(max(kh, 0), 2) * 3 + min(max(kw, 0), 2)) * CX + ci) * CZ + co];
                t += in ? v : 0.f;
            }
        return t;
    };
    if (CZ == 3) {
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int kk = 4 * s + lk;
            const bool ok = kk < 3 * TAPS;
            const int k2 = ok ? kk : 0;
            const int ta = k2 / (3 * TW), rem = k2 - ta * (3 * TW), tb = rem / 3, co = rem - tb * 3;
            zoff[s] = ta * RP + rem;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int ci = mt * 16 + li;
                wreg[s][0][mt] = (ok && ci < CX) ? wtap(ta, tb, ci, co) : 0.f;
            }
        }
    } else {
        zoff[0] = 0;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int tap = s / NH, h = s - tap * NH;
            const int ta = tap / TW, tb = tap - ta * TW;
#pragma unroll
            for (int j = 0; j < VEC; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int ci = mt * 16 + li;
                    const int co = h * 4 * VEC + VEC * lk + j;
                    wreg[s][j][mt] = ci < (WD ? 16 : CX) ? wtap(ta, tb, ci, co) : 0.f;
                }
        }
    }
    // the lane's pixel (and channel group) inside a tile; PH: the window of stored pixel (pr, pc) starts at (2 pr, 2 pc)
    const int lz = (PH ? 2 * (pr * RP + pc * PS) : (WD ? pr * RP + 2 * pc * PS : pr * RP + pc * PS)) + (CZ == 3 ? 0 : VEC * lk);

    DN_MARK(1, false);
    const int tps_log = a.tps_log, tps_mask = (1 << tps_log) - 1;
    DN_MARK(2, false);
    [[maybe_unused]] int pass = 0;
    for (; n0 < ne; n0 += step) {
        const int ns = min(a.ns, ne - n0);
        __syncthreads();
        store_pass(ns);
        __syncthreads();
        DN_MARK(3 + 2 * pass, false);
        if (n0 + step < ne) load_pass(n0 + step);

        const int ntile = (ns << tps_log) >> 2;  // tiles of this wave: tps is a multiple of 4
        float zv0[NSTEP][VEC], zv1[NSTEP][VEC];
        auto tile_pos = [&](int i, int& ss, int& oh, int& ow0) {
            const int t = wave + 4 * i;
            ss = t >> tps_log;
            const int q = t & tps_mask;
            oh = (q / TX) * 2;
            ow0 = (q % TX) * (WD ? 16 : 8);
        };
        auto load_tile = [&](float (&zv)[NSTEP][VEC], int i) {
            int ss, oh, ow0;
            tile_pos(i, ss, oh, ow0);
            const float* zq = smem + ss * sampleZ + (PH ? 2 : 1) * (oh * RP + ow0 * PS) + lz;  // PH: (oh, ow0) are stored pixels
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (CZ == 3) {
                    zv[s][0] = zq[zoff[s]];
                } else {
                    const int tap = s / NH, h = s - tap * NH;
                    const float* q = zq + (tap / TW) * RP + (tap % TW) * PS + h * 4 * VEC;
                    if (VEC == 2) {
                        const float2 v = *reinterpret_cast<const float2*>(q);
                        zv[s][0] = v.x; zv[s][VEC - 1] = v.y;
                    } else {
                        const float4 v = *reinterpret_cast<const float4*>(q);
                        zv[s][0] = v.x; zv[s][1 % VEC] = v.y; zv[s][2 % VEC] = v.z; zv[s][3 % VEC] = v.w;
                    }
                }
            }
        };
        auto mfma_tile = [&](const float (&zv)[NSTEP][VEC], int i, bool live) {
            dn_v4f acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = dn_v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NSTEP; ++s)
#pragma unroll
                for (int j = 0; j < VEC; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s][j][mt], zv[s][j], acc[mt], 0, 0, 0);
            int ss, oh, ow0;
            tile_pos(i, ss, oh, ow0);
            if (!live) return;
            // lane (li, lk) holds input channels mt * 16 + 4 lk .. + 3 of pixel (oh + pr, ow0 + pc)
            if (WD) {  // rows 4 lk .. 4 lk + 3 = channels 4 (lk & 1) .. + 3 of the pair's pixel lk >> 1
                float* o = a.dx + (((size_t)(n0 + ss) * H + oh + pr) * WO + ow0 + 2 * pc + (lk >> 1)) * CX + 4 * (lk & 1);
                *reinterpret_cast<float4*>(o) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
            } else if (PH) {
                float* o = a.dx + (((size_t)(n0 + ss) * (H >> 1) + oh + pr) * (WO >> 1) + ow0 + pc) * CX;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (mt * 16 + 4 * lk < CX)
                        *reinterpret_cast<float4*>(o + mt * 16 + 4 * lk) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
            } else if (a.up == 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[mt][r];
                        v += dn_dpp_xor1(v);
                        v += dn_dpp_ror8(v);
                        acc[mt][r] = v;
                    }
                if (pr == 0 && (pc & 1) == 0) {
                    float* o = a.dx + (((size_t)(n0 + ss) * (H >> 1) + (oh >> 1)) * (WO >> 1) + ((ow0 + pc) >> 1)) * CX;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        if (mt * 16 + 4 * lk < CX)
                            *reinterpret_cast<float4*>(o + mt * 16 + 4 * lk) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
                }
            } else {
                float* o = a.dx + (((size_t)(n0 + ss) * H + oh + pr) * WO + ow0 + pc) * CX;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (mt * 16 + 4 * lk < CX)
                        *reinterpret_cast<float4*>(o + mt * 16 + 4 * lk) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
            }
        };
        if constexpr (PING) {
            // tiles in pairs, the next tile's reads placed between the current tile's matrix instructions; an odd count
            // runs its first tile on its own, so the loop body has no branch
            constexpr int NM = NSTEP * VEC * MT;
            int first = 0;
            if (ntile & 1) {
                load_tile(zv0, 0);
                mfma_tile(zv0, 0, true);
                first = 1;
            }
            if (first < ntile) {
                load_tile(zv0, first);
                __builtin_amdgcn_sched_barrier(0);
                for (int i = first; i < ntile; i += 2) {
                    load_tile(zv1, i + 1);
                    mfma_tile(zv0, i, true);
                    dn_interleave<NM, NSTEP>();
                    __builtin_amdgcn_sched_barrier(0);
                    load_tile(zv0, min(i + 2, ntile - 1));
                    mfma_tile(zv1, i + 1, true);
                    dn_interleave<NM, NSTEP>();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            for (int i = 0; i < ntile; ++i) {
                load_tile(zv0, i);
                mfma_tile(zv0, i, true);
            }
        }
        DN_MARK(4 + 2 * pass, false);
        ++pass;
    }
    DN_MARK(15, true);
}

// ---- the input gradients of SEVERAL consecutive layers in one launch.  A sample's gradient chain touches no other
// sample, so a workgroup can take its samples through every layer in turn: layer j + 1 reads (as dy) what the workgroup
// itself wrote as layer j's dx, and nothing orders it against other workgroups.  Saves the launch ramps and gaps between
// the layers (each layer alone runs 8..12 us, of which 2..3 us pass before its first matrix instruction).
constexpr int kDnChainMax = 8;
struct DnChainArgs {
    DnArgs job[kDnChainMax];
    int code[kDnChainMax];  // shape * 4 + mode
    int n_jobs, S;          // S: samples per workgroup, the same range in every layer
};


// The layers of the table in order, for the samples [nb, nb + c.S) of this workgroup.  `c` must be readable through
// scalar loads with a run-time index (the kernel argument segment, not a by-value parameter, which the compiler would
// first copy into every lane's scratch memory).
// ALL: every shape of the family; otherwise only the layers of the default stack (the launch that runs the chain behind
// that stack's forward pass carries no other bodies).
template <bool ALL>
__device__ __forceinline__ void dn_chain_run(const DnChainArgs& c, const int nb, const bool fence_first) {
    for (int j = 0; j < c.n_jobs; ++j) {
        const DnArgs& a = c.job[j];
        const int ne = min(a.N, nb + c.S);
        if (j > 0 || fence_first) {
            // this workgroup's stores of the previous layer's dx before its loads of them; the barrier also keeps the
            // LDS images of the previous layer alive until its last wave has left the tile loop
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        switch (c.code[j]) {
            case 0: dn_body<3, 8, 32, 0>(a, nb, ne, a.ns, blockIdx.x); break;
            case 1: dn_body<3, 8, 32, 1>(a, nb, ne, a.ns, blockIdx.x); break;
            case 4: dn_body<8, 16, 16, 0>(a, nb, ne, a.ns, blockIdx.x); break;
            case 5: dn_body<8, 16, 16, 1>(a, nb, ne, a.ns, blockIdx.x); break;
            case 8: dn_body<16, 32, 8, 0>(a, nb, ne, a.ns, blockIdx.x); break;
            case 12: dn_body<32, 16, 8, 0>(a, nb, ne, a.ns, blockIdx.x); break;
            case 16: dn_body<16, 8, 16, 0>(a, nb, ne, a.ns, blockIdx.x); break;
            case 18: dn_body<16, 8, 16, 2>(a, nb, ne, a.ns, blockIdx.x); break;
            default:
                if constexpr (ALL) {  // 8 -> 8 channels at the three image widths
                    switch (c.code[j]) {
                        case 20: dn_body<8, 8, 32, 0>(a, nb, ne, a.ns, blockIdx.x); break;
                        case 21: dn_body<8, 8, 32, 1>(a, nb, ne, a.ns, blockIdx.x); break;
                        case 22: dn_body<8, 8, 32, 2>(a, nb, ne, a.ns, blockIdx.x); break;
                        case 24: dn_body<8, 8, 16, 0>(a, nb, ne, a.ns, blockIdx.x); break;
                        case 25: dn_body<8, 8, 16, 1>(a, nb, ne, a.ns, blockIdx.x); break;
                        case 26: dn_body<8, 8, 16, 2>(a, nb, ne, a.ns, blockIdx.x); break;
                        case 28: dn_body<8, 8, 8, 0>(a, nb, ne, a.ns, blockIdx.x); break;
                        default: break;
                    }
                }
                break;
        }
    }
}

// The table of a list of jobs for workgroups of S samples each (S <= 0: the planner's choice); *grid workgroups cover
// the batch, *lds_bytes is the largest layer's need.  0: planned; 1: a layer is outside the family, the batch sizes
// differ or the list is too long.  (dgrad_nhwc.hip)
int plan_dgrad_chain(const sv_dgrad_job* jobs, int n_jobs, int S, DnChainArgs* c, int* grid, size_t* lds_bytes);
// whether a job's code is one of the default stack's (dn_chain_run<false> carries it)
inline bool dn_code_in_default_stack(int code) { return code <= 18; }

}  // namespace sv
