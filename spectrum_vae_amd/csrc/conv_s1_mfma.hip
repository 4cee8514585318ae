// Stride-1 SAME convolutions with a 3x3 or 5x5 kernel on small images (4..32 pixels wide), any channel counts up to 128,
// on the f32 matrix cores with channel-last LDS images: the layer shapes the hyper-parameter grid of run_model.py:164-219
// produces for CAEPadded (autoencoder.py:437-526) outside its few-channel default stack - 16..128 channels, 5x5 kernels,
// several convs per block.  One kernel body serves
//   forward          y[p][co]  = act(bias[co] + sum_{tap, ci} x[p + tap][ci] w[tap][ci][co])       (+ 2x2 max-pool, argmax)
//   input gradient   dx[p][ci] = sum_{tap, co} dz[p + tap][co] w[mirrored tap][ci][co]              (+ 2x2 sum: upsampled input)
// as  out[p][ro] = sum_{tap, ri} in[p + tap][ri] A[tap][ri][ro]:  the filters are the MFMA A operand (16 result channels
// x 4 reduction channels), 16 pixels (a 2x8 or 4x4 patch holding whole pooling windows) the B operand columns.
//
// A workgroup (4 waves) carries `nu` units (a sample, or a band of rows of a large image) per pass:
//  * the input of the pass sits in LDS as planes of 16 channels, [plane][unit][row][col][16], in one of two layouts.
//    HALO: every row is followed by `pad` zero pixels (which are also the next row's left margin) and every unit by
//    `pad` zero rows (the next unit's top margin), so a tap's operand address is the pixel's address plus ONE
//    wave-uniform offset and the loop over the taps has no per-lane arithmetic and no branches (1.2 - 2.3x the image).
//    Without it (deep layers, where four 4x4 x 128-channel samples must share the LDS with the filter chunks): no
//    margins at all, a lane whose tap falls outside the image reads a zero pixel instead (one select per tap and pixel
//    tile, amortised over the channel groups).  An upsampled input (UpSampling2D folded into the conv) is replicated
//    while it is written to LDS;
//  * a lane's B operand for 4 MFMAs is ONE 16-byte read (4 consecutive channels of its pixel); the row pitch carries a
//    32-byte skew so that the 16-lane groups of a ds_read_b128 over a 2x8 / 4x4 patch hit distinct banks;
//  * the filters stream through LDS in chunks [tap][16 reduction channels][result channel][16], double buffered, the
//    next chunk's global loads in flight during a chunk's MFMAs (small filter sets stay resident); a lane's A operand
//    for 4 MFMAs is one 16-byte read, 1 KB contiguous per wave: conflict-free;
//  * a wave owns TI result-channel tiles x TJ pixel tiles (TI * TJ <= 8 accumulators), the waves split pixel tiles (and
//    result channels beyond 64);
//  * results leave from the accumulators: a lane holds 4 consecutive channels of a pixel = one 16-byte NHWC store; the
//    pooling maximum / argmax and the 2x2 sum of an upsampled layer's input gradient are two cross-lane steps.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "tile_common.h"

namespace sv {

typedef float s1_v4f __attribute__((ext_vector_type(4)));

struct S1Args {
    const float* in;    // [N][H / up][W / up][RI]
    const float* w;     // the forward conv's HWIO filters
    const float* bias;  // forward (may be null)
    float* out;         // [N][H][W][RO], or [N][H / 2][W / 2][RO] with epi 1 / 2
    uint8_t* idx;       // epi 1: argmax slot of every pooling window
    int N, H, W, Hs, Ws;
    int RI, RO, ROp;    // channels of in / out; result rows the waves cover (16 * TI * WR)
    int G, GC, TC;      // 16-channel reduction groups; groups / taps per filter chunk
    int K, pad, up, epi, dgrad, act;
    float alpha;
    int BR, nb_log, nu, units;  // rows per band, log2(bands per sample), units per pass, N * bands
    int WR_log;                 // log2(waves along the result-channel tiles)
    int TW_log;                 // 3: 2x8 pixel tiles, 2: 4x4
    int tpu_log;                // log2(tiles per unit)
    int rows_lds, pitch, plane; // window rows per unit; floats per row; floats per plane
    int halo, prefix, ustride, row0;  // zero-padded layout; floats before the first row; rows per unit; first unit's row
    int w_floats, resident, steps, gsteps;  // floats per filter buffer; all filters in one chunk; chunks per pass; G / GC
    int srows_max, ppu, pmax;   // stored rows of a full window; staging pieces per unit (max) and per thread
    FDiv f_ppu, f_k;
    int dbg;  // diagnostic build only (S1_DBG): 1 skip the matrix-core loop, 2 skip the staging, 4 skip the result stores
};

// Ablation switches exist in the diagnostic build (make TRACE=1) only: they make the results invalid.
#ifdef SV_TRACE
#define S1_DBG(a, bit) ((a).dbg & (bit))
#else
#define S1_DBG(a, bit) 0
#endif

constexpr int kS1PMax = 12;  // 16-byte staging pieces per thread and pass
constexpr int kS1PChunk = 6; // ... of which this many are in registers between their load and their LDS store
constexpr int kS1WMax = 4;   // filter pieces per thread and streamed chunk

__device__ __forceinline__ float s1_dpp_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ float s1_dpp_ror8(float v) {  // lane ^ 8 inside a row of 16 lanes
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
}
__device__ __forceinline__ int s1_dpp_xor1_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ int s1_dpp_ror8_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); }

template <int TI, int TJ, int VEC, bool HALO>
__global__ __launch_bounds__(256, 2) void s1_mfma_k(S1Args a) {
    constexpr int CPG = 4 * VEC;  // floats per pixel and plane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int TW_log = a.TW_log, TW = 1 << TW_log;
    const int ty = li >> TW_log, tx = li & (TW - 1);
    const int wr = wave & ((1 << a.WR_log) - 1), wn = wave >> a.WR_log;
    const int H = a.H, W = a.W, K = a.K, pad = a.pad;
    const int pitch = a.pitch, plane = a.plane, rows_lds = a.rows_lds;
    float* img = smem;
    float* wl = smem + a.G * plane;
    const int nb_mask = (1 << a.nb_log) - 1;
    const int tpu_mask = (1 << a.tpu_log) - 1;
    const int txl = __builtin_ctz((unsigned)(W >> TW_log));  // log2(tiles per image row)
    const int TH_log = 4 - TW_log;
    const bool banded = a.nb_log > 0;

    // ---- image staging: global -> registers (load_img) -> LDS (store_img)
    float4 rv[kS1PChunk];
    auto unit_rows = [&](int u, int& n, int& wy0, int& sa, int& cnt) {  // window of unit u and its stored rows
        n = u >> a.nb_log;
        const int r0 = (u & nb_mask) * a.BR;
        wy0 = banded ? r0 - pad : 0;
        const int ca = max(wy0, 0), cb = min(banded ? r0 + a.BR + pad : H, H);
        sa = ca / a.up;
        const int sb = (cb + a.up - 1) / a.up;
        cnt = (sb - sa) * a.Ws * a.RI;  // floats
    };
    auto load_img = [&](int u0, int i0) {
        const int nu = min(a.nu, a.units - u0);
#pragma unroll
        for (int ii = 0; ii < kS1PChunk; ++ii) {
            const int i = i0 + ii;
            const int q = threadIdx.x + i * 256;
            const int ul = fd(q, a.f_ppu), r = q - ul * a.ppu;
            rv[ii] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < a.pmax && ul < nu) {
                int n, wy0, sa, cnt;
                unit_rows(u0 + ul, n, wy0, sa, cnt);
                const float* src = a.in + ((size_t)n * a.Hs + sa) * a.Ws * a.RI;
                if (VEC == 1) {  // 3 channels: scalar elements
                    if (r < cnt) rv[ii].x = src[r];
                } else if (4 * r < cnt) {
                    rv[ii] = *reinterpret_cast<const float4*>(src + 4 * r);
                }
            }
        }
    };
    auto store_img = [&](int u0, int i0) {
        const int nu = min(a.nu, a.units - u0);
        const int ws_log = __builtin_ctz((unsigned)a.Ws);
#pragma unroll
        for (int ii = 0; ii < kS1PChunk; ++ii) {
            const int i = i0 + ii;
            const int q = threadIdx.x + i * 256;
            const int ul = fd(q, a.f_ppu), r = q - ul * a.ppu;
            if (i >= a.pmax || ul >= nu) continue;
            int n, wy0, sa, cnt;
            unit_rows(u0 + ul, n, wy0, sa, cnt);
            const int e = VEC == 1 ? r : 4 * r;
            if (e >= cnt) continue;
            int pi, ch;
            if (VEC == 1) { pi = e / 3; ch = e - 3 * pi; }
            else { pi = e / a.RI; ch = e - pi * a.RI; }
            const int srow = sa + (pi >> ws_log), scol = pi & (a.Ws - 1);
            float* dst = img + (VEC == 4 ? (ch >> 4) * plane + (ch & 15) : ch) + a.prefix + (ul * a.ustride + a.row0) * pitch;
            if (a.up == 1) {
                float* d = dst + (srow - wy0) * pitch + scol * CPG;
                if (VEC == 1) d[0] = rv[ii].x;
                else *reinterpret_cast<float4*>(d) = rv[ii];
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int lr = 2 * srow + (s >> 1) - wy0;
                    if (lr < 0 || lr >= rows_lds) continue;
                    float* d = dst + lr * pitch + (2 * scol + (s & 1)) * CPG;
                    if (VEC == 1) d[0] = rv[ii].x;
                    else *reinterpret_cast<float4*>(d) = rv[ii];
                }
            }
        }
    };

    // ---- filter chunks: [tc][gc][result channel][CPG reduction channels]
    const int ROp = a.ROp, rop_log = __builtin_ctz((unsigned)ROp);
    constexpr int QN_log = VEC == 4 ? 2 : (VEC == 2 ? 1 : 0);  // log2(16-byte pieces per result channel and group)
    const int gc_log = a.GC == 2 ? 1 : 0;
    const int KK = K * K;
    // Place (in 16-byte pieces) of reduction channels 4 qq .. 4 qq + 3 of result channel ro in block blk = (tap, group) of
    // a chunk.  With 16-channel groups a 16 x 16 block is stored [qq][result channel][4], so that the A-operand read of a
    // wave - lane (li, lk) takes piece (qq = lk, ro = li) - walks 1 KB in lane order: no bank conflicts.
    auto w_at = [&](int blk, int ro, int qq) {
        return VEC == 4 ? ((blk * (ROp >> 4) + (ro >> 4)) * 4 + qq) * 16 + (ro & 15) : (blk * ROp + ro) * (1 << QN_log) + qq;
    };
    // Piece p of the chunk starting at (tap0, g0): 4 consecutive reduction channels of one result channel.  The thread ->
    // piece map follows the filters' memory order (forward: result channels are contiguous in HWIO, lanes walk them;
    // input gradient: the reduction channels are, one 16-byte load per piece); `dst` is the piece's place in the chunk.
    auto w_piece = [&](int p, int tap0, int g0, float4& v, int& dst) {
        int ro, qq, gc, tc;
        if (a.dgrad) {
            qq = p & ((1 << QN_log) - 1);
            const int r1 = p >> QN_log;
            gc = r1 & (a.GC - 1);
            const int r2 = r1 >> gc_log;
            ro = r2 & (ROp - 1);
            tc = r2 >> rop_log;
        } else {
            ro = p & (ROp - 1);
            const int r1 = p >> rop_log;
            qq = r1 & ((1 << QN_log) - 1);
            const int r2 = r1 >> QN_log;
            gc = r2 & (a.GC - 1);
            tc = r2 >> gc_log;
        }
        dst = w_at((tc << gc_log) + gc, ro, qq);
        const int tap = tap0 + tc;
        const int ri0 = (g0 + gc) * 16 + 4 * qq;
        if (a.dgrad && (a.RI & 3) == 0) {
            const bool ok = ro < a.RO && ri0 < a.RI;
            v = ok ? *reinterpret_cast<const float4*>(a.w + ((size_t)((KK - 1 - tap) * a.RO + ro) * a.RI + ri0))
                   : make_float4(0.f, 0.f, 0.f, 0.f);
            return;
        }
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ri = ri0 + e;
            const bool ok = ro < a.RO && ri < a.RI;
            const int at = a.dgrad ? ((KK - 1 - tap) * a.RO + ro) * a.RI + ri : (tap * a.RI + ri) * a.RO + ro;
            t[e] = ok ? a.w[at] : 0.f;
        }
        v = make_float4(t[0], t[1], t[2], t[3]);
    };
    const int w_pieces = a.w_floats >> 2;
    // Streamed chunks (large filter sets: channel counts that are multiples of 16 on both sides): a thread moves 16
    // floats per chunk as four 16-byte loads and four 16-byte LDS stores whose addresses are formed ONCE (`gofs`, `ldst`
    // for the chunk at tap 0, group 0); a step adds a wave-uniform offset.  Forward: the four loads are 4 consecutive
    // result channels of 4 consecutive reduction channels (HWIO rows are contiguous in the result channel); the 4 x 4
    // block is transposed in registers on its way to LDS.  Input gradient: four independent pieces along the reduction
    // channels, which HWIO stores contiguously.
    float4 wv[4];
    int gofs[4], ldst[4];
    bool w_on = false;
    if (!a.resident) {
        const int t = threadIdx.x;
        w_on = t * 4 < w_pieces;  // one unit = 4 pieces
        if (a.dgrad) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = 4 * t + e;
                const int qq = p & 3, r1 = p >> 2, gc = r1 & (a.GC - 1), ro = (r1 >> gc_log) & (ROp - 1);
                gofs[e] = ((KK - 1) * a.RO + ro) * a.RI + gc * 16 + 4 * qq;  // tap 0 reads the last filter tap
                ldst[e] = w_at(gc, ro, qq);
                if (ro >= a.RO) { gofs[e] = -1; }
            }
        } else {
            const int ro4 = t & ((ROp >> 2) - 1), r1 = t >> (rop_log - 2), riq = r1 & 3, gc = (r1 >> 2) & (a.GC - 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gofs[e] = (gc * 16 + 4 * riq + e) * a.RO + 4 * ro4;              // load e: reduction channel 4 riq + e
                ldst[e] = w_at(gc, 4 * ro4 + e, riq);                            // store e: result channel 4 ro4 + e
                if (4 * ro4 >= a.RO) gofs[e] = -1;
            }
        }
    }
    auto load_w = [&](int step) {
        if (!w_on) return;
        const int tap = step / a.gsteps, g0 = (step - tap * a.gsteps) * a.GC;
        const int soff = a.dgrad ? g0 * 16 - tap * a.RO * a.RI : (tap * a.RI + g0 * 16) * a.RO;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            wv[e] = gofs[e] >= 0 ? *reinterpret_cast<const float4*>(a.w + gofs[e] + soff) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_w = [&](int buf) {
        if (!w_on) return;
        float4* d = reinterpret_cast<float4*>(wl + buf * a.w_floats);
        if (a.dgrad) {
#pragma unroll
            for (int e = 0; e < 4; ++e) d[ldst[e]] = wv[e];
        } else {
            d[ldst[0]] = make_float4(wv[0].x, wv[1].x, wv[2].x, wv[3].x);
            d[ldst[1]] = make_float4(wv[0].y, wv[1].y, wv[2].y, wv[3].y);
            d[ldst[2]] = make_float4(wv[0].z, wv[1].z, wv[2].z, wv[3].z);
            d[ldst[3]] = make_float4(wv[0].w, wv[1].w, wv[2].w, wv[3].w);
        }
    };

    // HALO, banded units: the window rows that lie outside the image (top of the first band, bottom of the last) hold
    // another band's data from the previous pass: zeros again
    auto zero_outside = [&](int u0) {
        const int nu = min(a.nu, a.units - u0);
        const int rowf = W * CPG;
        for (int ul = 0; ul < nu; ++ul) {
            const int r0 = ((u0 + ul) & nb_mask) * a.BR, wy0 = r0 - pad;
            const int top = max(0, -wy0), bot = max(0, wy0 + rows_lds - H);  // rows above / below the image
            if (top + bot == 0) continue;
            const int total = a.G * (top + bot) * rowf;
            for (int i = threadIdx.x; i < total; i += 256) {
                const int g = i / ((top + bot) * rowf), r = i - g * (top + bot) * rowf;
                const int row = r / rowf, c = r - row * rowf;
                const int lr = row < top ? row : rows_lds - bot + (row - top);
                img[g * plane + a.prefix + (ul * a.ustride + a.row0 + lr) * pitch + c] = 0.f;
            }
        }
    };

    // ---- one-time set-up: zero pixels / margins, resident filters, bias
    if (HALO) {
        const int total = (a.G * plane) >> 2;
        for (int i = threadIdx.x; i < total; i += 256) reinterpret_cast<float4*>(img)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int g = 0; g < a.G; ++g)
            if (threadIdx.x < 16) img[g * plane + threadIdx.x] = 0.f;
    }
    if (a.resident) {
        float4* d = reinterpret_cast<float4*>(wl);
        for (int p = threadIdx.x; p < w_pieces; p += 256) {
            float4 v;
            int at;
            w_piece(p, 0, 0, v, at);
            d[at] = v;
        }
    }
    if (VEC != 4 && !HALO) {  // the padding channels of every pixel stay zero: clear the image once
        const int total = plane;
        for (int i = threadIdx.x; i < total; i += 256) img[i] = 0.f;
    }
    float bias[TI][4];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ro = (wr * TI + ti) * 16 + 4 * lk + r;
            bias[ti][r] = (a.bias != nullptr && ro < a.RO) ? a.bias[ro] : 0.f;
        }

    int u0 = blockIdx.x * a.nu;
    if (!a.resident) load_w(0);
    int wbuf = 0;  // buffer holding the chunk of the step about to run
    bool first = true;
    for (; u0 < a.units; u0 += gridDim.x * a.nu) {
        const int nu = min(a.nu, a.units - u0);
        // The pass's input: global -> registers -> LDS in one go.  (Holding the next pass's pieces in registers across
        // the arithmetic costs 48 registers and spills; the second workgroup of the CU covers this round trip.)
        if (!S1_DBG(a, 2)) load_img(u0, 0);
        __syncthreads();  // the previous pass's operand reads are done
        if (!S1_DBG(a, 2)) {
            store_img(u0, 0);
            for (int i0 = kS1PChunk; i0 < a.pmax; i0 += kS1PChunk) {
                load_img(u0, i0);
                store_img(u0, i0);
            }
            if (HALO && banded) zero_outside(u0);
        }
        if (first && !a.resident) store_w(0);
        first = false;
        __syncthreads();

        // ---- this wave's pixel tiles
        int pix_y[TJ], pix_x[TJ], pix_n[TJ], base[TJ];
        bool live[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int pt = wn * TJ + j;
            int ul = pt >> a.tpu_log;
            live[j] = ul < nu;
            if (!live[j]) ul = 0;
            const int t = pt & tpu_mask;
            const int u = u0 + ul;
            const int r0 = (u & nb_mask) * a.BR;
            const int wy0 = banded ? r0 - pad : 0;
            pix_n[j] = u >> a.nb_log;
            pix_y[j] = r0 + ((t >> txl) << TH_log) + ty;
            pix_x[j] = ((t & ((1 << txl) - 1)) << TW_log) + tx;
            base[j] = a.prefix + (ul * a.ustride + a.row0 + pix_y[j] - wy0) * pitch + pix_x[j] * CPG + lk * VEC;
        }
        s1_v4f acc[TI][TJ];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[ti][j] = s1_v4f{0.f, 0.f, 0.f, 0.f};

        for (int s = 0; s < (S1_DBG(a, 1) ? 0 : a.steps); ++s) {
            if (!a.resident) load_w(s + 1 < a.steps ? s + 1 : 0);  // the next chunk (of the next pass after the last step)
            const int tap0 = a.resident ? 0 : (s / a.gsteps) * a.TC;
            const int g0 = a.resident ? 0 : (s - (s / a.gsteps) * a.gsteps) * a.GC;
            const float* wb = wl + (a.resident ? 0 : wbuf * a.w_floats) +
                              (VEC == 4 ? ((wr * TI) * 4 + lk) * 64 + li * 4 : ((wr * TI) * 16 + li) * CPG + lk * VEC);
            // iteration it = (tap tc, group gc) of the chunk; the operand reads of iteration it + 1 are issued before the
            // matrix instructions of iteration it (two register sets)
            const int n_it = a.TC * a.GC;
            const int gcl = __builtin_ctz((unsigned)a.GC);
            int boff[TJ];
            auto read_ops = [&](int it, float (&av)[TI][VEC], float (&bv)[TJ][VEC]) {
                const int tc = it >> gcl, gc = it & (a.GC - 1);
                const float* ap = wb + it * ROp * CPG;
                const float* bp = img + (g0 + gc) * plane;
                if (HALO) {  // the tap is one wave-uniform offset: margins hold the zeros
                    const int tap = tap0 + tc;
                    const int kh = fd(tap, a.f_k), kw = tap - kh * K;
                    bp += (kh - pad) * pitch + (kw - pad) * CPG;
                } else if (gc == 0) {  // a new tap: where each pixel tile's lane reads (the zero pixel outside the image)
                    const int tap = tap0 + tc;
                    const int kh = fd(tap, a.f_k), kw = tap - kh * K;
                    const int dy = kh - pad, dx = kw - pad;
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        const bool ok = (unsigned)(pix_y[j] + dy) < (unsigned)H && (unsigned)(pix_x[j] + dx) < (unsigned)W;
                        boff[j] = ok ? base[j] + dy * pitch + dx * CPG : lk * VEC;
                    }
                }
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) {
                    if (VEC == 4) {
                        const float4 v = *reinterpret_cast<const float4*>(ap + ti * 256);
                        av[ti][0] = v.x; av[ti][1 % VEC] = v.y; av[ti][2 % VEC] = v.z; av[ti][3 % VEC] = v.w;
                    } else if (VEC == 2) {
                        const float2 v = *reinterpret_cast<const float2*>(ap + ti * 16 * CPG);
                        av[ti][0] = v.x; av[ti][1 % VEC] = v.y;
                    } else {
                        av[ti][0] = ap[ti * 16 * CPG];
                    }
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    if (VEC == 4) {
                        const float4 v = *reinterpret_cast<const float4*>(bp + (HALO ? base[j] : boff[j]));
                        bv[j][0] = v.x; bv[j][1 % VEC] = v.y; bv[j][2 % VEC] = v.z; bv[j][3 % VEC] = v.w;
                    } else if (VEC == 2) {
                        const float2 v = *reinterpret_cast<const float2*>(bp + (HALO ? base[j] : boff[j]));
                        bv[j][0] = v.x; bv[j][1 % VEC] = v.y;
                    } else {
                        bv[j][0] = bp[HALO ? base[j] : boff[j]];
                    }
                }
            };
            auto mfma_ops = [&](const float (&av)[TI][VEC], const float (&bv)[TJ][VEC]) {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
#pragma unroll
                    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                        for (int j = 0; j < TJ; ++j)
                            acc[ti][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ti][e], bv[j][e], acc[ti][j], 0, 0, 0);
            };
            float av0[TI][VEC], bv0[TJ][VEC], av1[TI][VEC], bv1[TJ][VEC];
            // the loop body has no branch (HALO), so the wait before a set's matrix instructions counts only that set's
            // reads and the other set's stay in flight; the last one or two iterations run behind the loop
            read_ops(0, av0, bv0);
            int it = 0;
            for (; it + 2 < n_it; it += 2) {
                read_ops(it + 1, av1, bv1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_ops(av0, bv0);
                __builtin_amdgcn_sched_barrier(0);
                read_ops(it + 2, av0, bv0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_ops(av1, bv1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it + 2 == n_it) {
                read_ops(it + 1, av1, bv1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_ops(av0, bv0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_ops(av1, bv1);
            } else {
                mfma_ops(av0, bv0);
            }
            if (!a.resident) {
                store_w(wbuf ^ 1);  // last read one step ago; every wave has passed that step's barrier
                wbuf ^= 1;
                __syncthreads();
            }
        }

        // ---- results
        const int Ho = a.epi ? H >> 1 : H, Wo = a.epi ? W >> 1 : W;
        const bool corner = ((ty | tx) & 1) == 0;
        const int slot = ((ty & 1) << 1) | (tx & 1);
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            const int ro0 = (wr * TI + ti) * 16 + 4 * lk;
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                float v[4];
                int sl[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[ti][j][r];
                    if (!a.dgrad) v[r] = act_apply(v[r] + bias[ti][r], a.act, a.alpha);
                    sl[r] = slot;
                }
                if (a.epi == 1) {  // maximum of the 2x2 window and the first slot holding it
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        {
                            const float ov = s1_dpp_xor1(v[r]);
                            const int os = s1_dpp_xor1_i(sl[r]);
                            const bool take = ov > v[r] || (ov == v[r] && os < sl[r]);
                            v[r] = take ? ov : v[r];
                            sl[r] = take ? os : sl[r];
                        }
                        {
                            const float ov = TW_log == 3 ? s1_dpp_ror8(v[r]) : __shfl_xor(v[r], 4, 64);
                            const int os = TW_log == 3 ? s1_dpp_ror8_i(sl[r]) : __shfl_xor(sl[r], 4, 64);
                            const bool take = ov > v[r] || (ov == v[r] && os < sl[r]);
                            v[r] = take ? ov : v[r];
                            sl[r] = take ? os : sl[r];
                        }
                    }
                } else if (a.epi == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] += s1_dpp_xor1(v[r]);
                        v[r] += TW_log == 3 ? s1_dpp_ror8(v[r]) : __shfl_xor(v[r], 4, 64);
                    }
                }
                if (!live[j] || ro0 >= a.RO || (a.epi && !corner) || S1_DBG(a, 4)) continue;
                const int oy = a.epi ? pix_y[j] >> 1 : pix_y[j], ox = a.epi ? pix_x[j] >> 1 : pix_x[j];
                const size_t at = (((size_t)pix_n[j] * Ho + oy) * Wo + ox) * a.RO + ro0;
                if ((a.RO & 3) == 0) {
                    *reinterpret_cast<float4*>(a.out + at) = make_float4(v[0], v[1], v[2], v[3]);
                    if (a.epi == 1)
                        *reinterpret_cast<uint32_t*>(a.idx + at) =
                            (uint32_t)sl[0] | ((uint32_t)sl[1] << 8) | ((uint32_t)sl[2] << 16) | ((uint32_t)sl[3] << 24);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ro0 + r < a.RO) {
                            a.out[at + r] = v[r];
                            if (a.epi == 1) a.idx[at + r] = (uint8_t)sl[r];
                        }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- host side
struct S1Plan {
    bool ok;
    int ti, tj, vec;
    S1Args a;
    size_t lds;
    int grid;
};

constexpr size_t kS1LdsBudget = 80 * 1024;  // two workgroups share a CU's 160 KB

// RI -> RO channels on H x W (logical) images, K x K taps, input stored at 1 / up, epi 0 store, 1 pool, 2 sum.
static S1Plan plan_s1(int N, int H, int W, int RI, int RO, int K, int up, int epi) {
    S1Plan p{};
    p.ok = false;
    if (!(K == 3 || K == 5) || !(up == 1 || up == 2)) return p;
    if (!(W == 4 || W == 8 || W == 16 || W == 32) || H < 4 || (H & 1) || H > 64) return p;
    if (W == 4 && (H & 3)) return p;
    if (up == 2 && ((H | W) & 1)) return p;
    if (N < 1 || (int64_t)N * H * W * (RI > RO ? RI : RO) >= (1ll << 31)) return p;
    int vec, G;
    if (RI == 3) { vec = 1; G = 1; }
    else if (RI == 8) { vec = 2; G = 1; }
    else if (RI % 16 == 0 && RI <= 128) { vec = 4; G = RI / 16; }
    else return p;
    if (!(RO == 3 || ((RO & 3) == 0 && RO >= 4 && RO <= 128))) return p;
    if (epi == 1 && (RO & 3)) return p;
    const int R = (RO + 15) / 16;
    const int ti = R == 1 ? 1 : (R == 2 ? 2 : 4);
    const int wr = R > 4 ? 2 : 1, wn = 4 / wr;
    const int cpg = 4 * vec, pad = (K - 1) / 2;
    const int tw_log = W >= 8 ? 3 : 2, th = 16 >> tw_log;
    const int tps = H * W / 16;
    if (tps & (tps - 1)) return p;  // H a power of two as well
    S1Args& a = p.a;
    for (int tj = 8 / ti; tj >= 1; tj >>= 1) {
        if (ti == 1 && tj < 4) break;  // instantiated pairs: (1,8) (1,4) (2,4) (2,2) (4,2) (4,1)
        const int T = wn * tj;  // tiles per pass
        int BR, nb, nu;
        if (tps >= T) {
            BR = T * 16 / W; nb = H / BR; nu = 1;
            if (BR % th || BR * nb != H || (nb & (nb - 1))) continue;
        } else {
            BR = H; nb = 1; nu = T / tps;
        }
        if (nu > N * nb) nu = N * nb;  // fewer units than a pass holds: the spare tiles idle
        const int rows_lds = nb > 1 ? BR + 2 * pad : H;
        const int rop = 16 * ti * wr;
        const size_t w_all = (size_t)K * K * G * rop * cpg * sizeof(float);
        const int resident = w_all <= 36 * 1024;
        const int GC = resident ? G : (G >= 2 ? 2 : 1), TC = resident ? K * K : 1;
        const int w_floats = TC * GC * rop * cpg;
        if (!resident && w_floats / 4 > 256 * kS1WMax) continue;
        // the zero-padded layout where it fits (16-channel groups only), else the bare one
        int halo = vec == 4 ? 1 : 0, pitch = 0, plane = 0, prefix = 0, ustride = 0, row0 = 0;
        size_t lds = 0;
        for (; halo >= 0; --halo) {
            pitch = (W + (halo ? pad : 0)) * cpg + 8;
            prefix = halo ? (pad * cpg > 16 ? pad * cpg : 16) : 16;
            ustride = halo && nb == 1 ? H + pad : rows_lds;
            row0 = halo && nb == 1 ? pad : 0;
            plane = (prefix + (nu * ustride + row0) * pitch + 3) & ~3;
            lds = ((size_t)G * plane + (size_t)(resident ? 1 : 2) * w_floats) * sizeof(float);
            if (lds <= kS1LdsBudget) break;
        }
        if (halo < 0) continue;
        const int srows_max = up == 1 ? rows_lds : rows_lds / 2 + 1;
        const int ppu = vec == 1 ? srows_max * (W / up) * RI : srows_max * (W / up) * RI / 4;
        const int pmax = (nu * ppu + 255) / 256;
        if (pmax > kS1PMax) continue;
        a.N = N; a.H = H; a.W = W; a.Hs = H / up; a.Ws = W / up;
        a.RI = RI; a.RO = RO; a.ROp = rop; a.G = G; a.GC = GC; a.TC = TC;
        a.K = K; a.pad = pad; a.up = up; a.epi = epi;
        a.BR = BR; a.nb_log = __builtin_ctz((unsigned)nb); a.nu = nu; a.units = N * nb;
        a.WR_log = wr == 2 ? 1 : 0; a.TW_log = tw_log; a.tpu_log = __builtin_ctz((unsigned)(BR * W / 16));
        a.rows_lds = rows_lds; a.pitch = pitch; a.plane = plane;
        a.halo = halo; a.prefix = prefix; a.ustride = ustride; a.row0 = row0;
        a.w_floats = w_floats; a.resident = resident; a.gsteps = G / GC; a.steps = resident ? 1 : K * K * (G / GC);
        a.srows_max = srows_max; a.ppu = ppu; a.pmax = pmax;
        a.f_ppu = fd_make(ppu); a.f_k = fd_make(K);
        p.ti = ti; p.tj = tj; p.vec = vec; p.lds = lds;
        const int passes = (a.units + nu - 1) / nu;
        p.grid = passes < 512 ? passes : 512;
        p.ok = true;
        return p;
    }
    return p;
}

template <int TI, int TJ, int VEC, bool HALO>
static int s1_launch(const S1Plan& p, hipStream_t st) {
    static bool raised = false;  // dynamic LDS beyond 64 KB needs the opt-in once per kernel
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&s1_mfma_k<TI, TJ, VEC, HALO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kS1LdsBudget);
        if (e != hipSuccess) return fail(SV_E_LAUNCH, "s1_mfma: raising the LDS limit failed: %s", hipGetErrorString(e));
        raised = true;
    }
    hipLaunchKernelGGL((s1_mfma_k<TI, TJ, VEC, HALO>), dim3((unsigned)p.grid), dim3(256), p.lds, st, p.a);
    return check_launch("s1_mfma");
}

static int s1_dispatch(S1Plan& p, hipStream_t st) {
#ifdef SV_TRACE
    if (const char* e = getenv("SPECTRUM_S1_DBG")) p.a.dbg = atoi(e);
#endif
    if (p.vec == 1) return p.tj == 8 ? s1_launch<1, 8, 1, false>(p, st) : s1_launch<1, 4, 1, false>(p, st);
    if (p.vec == 2) return p.tj == 8 ? s1_launch<1, 8, 2, false>(p, st) : s1_launch<1, 4, 2, false>(p, st);
    if (p.a.halo) {
        if (p.ti == 1) return p.tj == 8 ? s1_launch<1, 8, 4, true>(p, st) : s1_launch<1, 4, 4, true>(p, st);
        if (p.ti == 2) return p.tj == 4 ? s1_launch<2, 4, 4, true>(p, st) : s1_launch<2, 2, 4, true>(p, st);
        return p.tj == 2 ? s1_launch<4, 2, 4, true>(p, st) : s1_launch<4, 1, 4, true>(p, st);
    }
    if (p.ti == 1) return p.tj == 8 ? s1_launch<1, 8, 4, false>(p, st) : s1_launch<1, 4, 4, false>(p, st);
    if (p.ti == 2) return p.tj == 4 ? s1_launch<2, 4, 4, false>(p, st) : s1_launch<2, 2, 4, false>(p, st);
    return p.tj == 2 ? s1_launch<4, 2, 4, false>(p, st) : s1_launch<4, 1, 4, false>(p, st);
}

static bool s1_geom(const sv_conv_geom& g) {
    return g.KH == g.KW && (g.KH == 3 || g.KH == 5) && g.SH == 1 && g.SW == 1 && g.pad_t == (g.KH - 1) / 2 &&
           g.pad_l == (g.KW - 1) / 2 && g.Ho == g.H && g.Wo == g.W && g.up_h == g.up_w && (g.up_h == 1 || g.up_h == 2) &&
           opt(OPT_S1_MFMA) != 0;
}

// which: 0 forward, 1 forward + pool, 2 input gradient (of an explicit dz)
bool s1_mfma_supported(const sv_conv_geom& g, int which) {
    if (!s1_geom(g)) return false;
    if (which == 2) return plan_s1(g.N, g.H, g.W, g.Cout, g.Cin, g.KH, 1, g.up_h == 2 ? 2 : 0).ok;
    return plan_s1(g.N, g.H, g.W, g.Cin, g.Cout, g.KH, g.up_h, which == 1 ? 1 : 0).ok;
}

// 1: geometry outside the family; 0: launched; < 0: error
int launch_s1_fwd(const sv_conv_geom& g, const float* x, const float* w, const float* bias, float* y, uint8_t* idx,
                  int act, float alpha, hipStream_t st) {
    if (!s1_geom(g)) return 1;
    S1Plan p = plan_s1(g.N, g.H, g.W, g.Cin, g.Cout, g.KH, g.up_h, idx != nullptr ? 1 : 0);
    if (!p.ok) return 1;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return 1;
    p.a.in = x; p.a.w = w; p.a.bias = bias; p.a.out = y; p.a.idx = idx; p.a.dgrad = 0; p.a.act = act; p.a.alpha = alpha;
    return s1_dispatch(p, st);
}

// dx of the conv described by g from an explicit dz [N][H][W][Cout]; a folded upsampling turns into the 2x2 sum
int launch_s1_dgrad(const sv_conv_geom& g, const float* dz, const float* w, float* dx, hipStream_t st) {
    if (!s1_geom(g)) return 1;
    S1Plan p = plan_s1(g.N, g.H, g.W, g.Cout, g.Cin, g.KH, 1, g.up_h == 2 ? 2 : 0);
    if (!p.ok) return 1;
    if ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(dx)) & 15) return 1;
    p.a.in = dz; p.a.w = w; p.a.bias = nullptr; p.a.out = dx; p.a.idx = nullptr; p.a.dgrad = 1; p.a.act = SV_ACT_NONE;
    p.a.alpha = 0.f;
    return s1_dispatch(p, st);
}

}  // namespace sv
