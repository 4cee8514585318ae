// Input gradients of 3x3 stride-1 SAME convs with few channels on the f32 matrix cores, channel-last LDS image (the
// activation-gradient chain of the tape in run_model.py:281-289 through the conv stack of autoencoder.py:437-526).
//
//   dx[p][ci] = sum over taps (kh, kw) and co of dz[p - (kh - 1, kw - 1)][co] * w[kh][kw][ci][co]
//
// as a GEMM with the input channels as rows (the filters are the A operand and stay in registers for the whole
// launch), 16 pixels (a 2 x 8 patch) as columns and (tap, co) as the reduction.  dz = dy * act'(y), scattered through the
// pooling argmax where the layer was pooled, is formed while it is copied into an NHWC LDS image with a zero halo; a
// lane's B operand for 4 (2 for 8-channel dz) consecutive reduction steps is ONE 16-byte (8-byte) read of consecutive
// channels of its pixel, at a per-tile base plus a compile-time tap offset.  The result tile holds 4 consecutive
// channels of a pixel per lane: it is stored as 16-byte pieces in NHWC order, after the 2 x 2 sum (two DPP adds) when
// the conv read its input through nearest upsampling.  No partial sums leave the workgroup: results are exact sums in
// a fixed order.
#include "dgrad_nhwc_body.h"

namespace sv {

template <int CZ, int CX, int WO, int MODE>
__global__ __launch_bounds__(256, 2) void dgrad_nhwc(DnArgs a) {
    dn_body<CZ, CX, WO, MODE>(a, blockIdx.x * a.ns, a.N, gridDim.x * a.ns, blockIdx.x);
}

__global__ __launch_bounds__(256, 2) void dgrad_chain_k(DnChainArgs) {
    // the table is indexed by a loop counter: read it where the launch put it (the kernel argument segment, scalar loads)
#if defined(__HIP_DEVICE_COMPILE__)
    const DnChainArgs& c = *(const DnChainArgs*)(const void*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    const DnChainArgs c{};  // host pass: the body is never run
#endif
    dn_chain_run<true>(c, blockIdx.x * c.S, false);
}

// ---------------------------------------------------------------------------------------------------- host side
template <int CZ, int CX, int WO>
static int dn_launch(const DnArgs& a, int grid, size_t lds, hipStream_t st, int mode) {
    if constexpr (WO >= 16) {  // a tile is 2 x 8 stored pixels (mode 1) / 2 x 8 pixel pairs (mode 2): 16 columns at least
        if (mode == 1) {
            hipLaunchKernelGGL((dgrad_nhwc<CZ, CX, WO, 1>), dim3((unsigned)grid), dim3(256), lds, st, a);
            return check_launch("dgrad_nhwc");
        }
        if constexpr (CX == 8 && CZ != 3) {
            if (mode == 2) {
                hipLaunchKernelGGL((dgrad_nhwc<CZ, CX, WO, 2>), dim3((unsigned)grid), dim3(256), lds, st, a);
                return check_launch("dgrad_nhwc");
            }
        }
    }
    hipLaunchKernelGGL((dgrad_nhwc<CZ, CX, WO, 0>), dim3((unsigned)grid), dim3(256), lds, st, a);
    return check_launch("dgrad_nhwc");
}

// The kernel form of one layer: shape and mode (-> template arguments), tiles per sample, staging units per sample.
struct DnPlan {
    int shape, mode, tps, zmode;
    int64_t sampleZ, units;
    size_t lds(int ns) const { return (size_t)(ns * sampleZ + 3 + 16) / 4 * 16 + w_bytes; }
    size_t w_bytes;
};

// 0: planned; 1: shape outside this family
static int dn_plan(const sv_conv_geom& g, const float* y, const uint8_t* idx, DnPlan& p) {
    if (opt(OPT_DGRAD_NHWC) == 0) return 1;
    if (g.KH != 3 || g.KW != 3 || g.SH != 1 || g.SW != 1 || g.up_h != g.up_w || (g.up_h != 1 && g.up_h != 2)) return 1;
    if (g.Ho != g.H || g.Wo != g.W || g.pad_t != 1 || g.pad_l != 1) return 1;
    const int H = g.H, W = g.W, CZ = g.Cout, CX = g.Cin;
    const int zmode = idx != nullptr ? 2 : 1;
    int shape;
    if (CZ == 3 && CX == 8 && W == 32) shape = 0;
    else if (CZ == 8 && CX == 16 && W == 16) shape = 1;
    else if (CZ == 16 && CX == 32 && W == 8) shape = 2;
    else if (CZ == 32 && CX == 16 && W == 8) shape = 3;
    else if (CZ == 16 && CX == 8 && W == 16) shape = 4;
    else if (CZ == 8 && CX == 8 && W == 32) shape = 5;
    else if (CZ == 8 && CX == 8 && W == 16) shape = 6;
    else if (CZ == 8 && CX == 8 && W == 8) shape = 7;
    else return 1;
    if (CZ == 3 && zmode == 2) return 1;
    if ((H & 1) || y == nullptr) return 1;
    // an upsampled conv as a 4 x 4 window over the stored pixels (PH in the kernel) where the stored image holds tiles
    bool ph = g.up_h == 2 && (shape == 0 || shape == 1 || shape == 5 || shape == 6) && (H % 4) == 0 && (W % 16) == 0;
    if (opt(OPT_DGRAD_PHASE) == 0) ph = false;
    int tps = ph ? (H / 4) * (W / 16) : (H / 2) * (W / 8);
    if (ph && (tps < 4 || (tps & (tps - 1)))) { ph = false; tps = (H / 2) * (W / 8); }
    // 8 input channels without upsampling: pixel pairs as result columns (MODE 2 in the kernel)
    bool wd = !ph && g.up_h == 1 && (shape == 4 || shape == 5 || shape == 6) && (W % 16) == 0;
    if (opt(OPT_DGRAD_PAIR) == 0) wd = false;
    if (wd) {
        const int t2 = (H / 2) * (W / 16);
        if (t2 >= 4 && !(t2 & (t2 - 1))) tps = t2;
        else wd = false;
    }
    if (tps < 4 || (tps & (tps - 1))) return 1;
    p.shape = shape;
    p.mode = ph ? 1 : (wd ? 2 : 0);
    p.tps = tps;
    p.zmode = zmode;
    p.sampleZ = (int64_t)(H + 2) * dn_rp(CZ, W, p.mode);
    p.units = (zmode == 2 ? (int64_t)(H / 2) * (W / 2) * CZ : (int64_t)H * W * CZ) / 4;
    p.w_bytes = (size_t)9 * CX * CZ * sizeof(float);
    return 0;
}

static void dn_fill(DnArgs& a, const DnPlan& p, const sv_conv_geom& g, const float* w, const float* dy, const float* y,
                    const uint8_t* idx, float* dx, int act, float alpha, int ns) {
    a = DnArgs{};
    a.dy = dy; a.zy = y; a.zidx = idx; a.w = w; a.dx = dx;
    a.N = g.N; a.H = g.H; a.up = g.up_h; a.zmode = p.zmode; a.ns = ns; a.tps_log = __builtin_ctz(p.tps);
    a.zk = act_grad_coef(act, alpha);
}

// dx of the conv described by g (its input gradient), dz formed from (dy, y, idx).  1: shape outside this family (the
// caller takes the planar kernels); 0: launched; < 0: error.
int launch_dgrad_nhwc(const sv_conv_geom& g, const float* w, const float* dy, const float* y, const uint8_t* idx, float* dx,
                      int act, float alpha, hipStream_t st) {
    DnPlan p;
    if (dn_plan(g, y, idx, p)) return 1;
    const int shape = p.shape, CZ = g.Cout;
    const bool ph = p.mode == 1, wd = p.mode == 2;
    const int64_t units = p.units;
    auto lds = [&](int ns) { return p.lds(ns); };
    // workgroups (measured at batch 1024): one sample each for the 32-wide 3-channel image (few registers, 14 KB of LDS:
    // four workgroups share a CU and hide each other's load latencies), two per CU for the 16-wide image with 8-channel
    // dz, one per CU elsewhere, where more workgroups only repeat the filter loads
    int grid = shape == 0 ? 1024 : ((shape == 1 || shape == 3 || shape == 4 || shape >= 5) ? 512 : 256);
    if (ph) grid = 512;  // a quarter of the tiles per sample: fewer, longer-lived workgroups
    if (opt(OPT_DGRAD_GRID) > 0) grid = opt(OPT_DGRAD_GRID);
    if (grid > g.N) grid = g.N;
    int ns = 1;
    // one pass per workgroup where the samples fit (LDS, staging registers): a second pass costs its staging and two
    // barriers again, and two workgroups per CU already overlap each other's phases
    const int per_pass = 256 * dn_units(CZ, g.W);
    while (ns < 8 && lds(ns + 1) <= 64 * 1024 && (ns + 1) * units <= per_pass && (int64_t)(ns + 1) * grid <= g.N) ++ns;
    if (lds(ns) > 64 * 1024 || ns * units > per_pass) return 1;
    if ((int64_t)grid * ns > g.N) grid = (g.N + ns - 1) / ns;
    DnArgs a;
    dn_fill(a, p, g, w, dy, y, idx, dx, act, alpha, ns);
    const size_t bytes = lds(ns);
    switch (shape) {
        case 0: return dn_launch<3, 8, 32>(a, grid, bytes, st, ph ? 1 : 0);
        case 1: return dn_launch<8, 16, 16>(a, grid, bytes, st, ph ? 1 : 0);
        case 2: return dn_launch<16, 32, 8>(a, grid, bytes, st, 0);
        case 3: return dn_launch<32, 16, 8>(a, grid, bytes, st, 0);
        case 4: return dn_launch<16, 8, 16>(a, grid, bytes, st, wd ? 2 : 0);
        case 5: return dn_launch<8, 8, 32>(a, grid, bytes, st, p.mode);
        case 6: return dn_launch<8, 8, 16>(a, grid, bytes, st, p.mode);
        default: return dn_launch<8, 8, 8>(a, grid, bytes, st, 0);
    }
}

int plan_dgrad_chain(const sv_dgrad_job* jobs, int n_jobs, int S, DnChainArgs* c, int* grid, size_t* lds_bytes) {
    if (n_jobs < 1 || n_jobs > kDnChainMax || opt(OPT_DGRAD_CHAIN) == 0) return 1;
    const int N = jobs[0].g.N;
    if (S <= 0) {
        int g = opt(OPT_DGRAD_CHAIN) > 1 ? opt(OPT_DGRAD_CHAIN) : 512;
        if (g > N) g = N;
        S = (N + g - 1) / g;
    }
    *grid = (N + S - 1) / S;
    *c = DnChainArgs{};
    size_t bytes = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const sv_dgrad_job& q = jobs[j];
        DnPlan p;
        if (q.g.N != N || dn_plan(q.g, (const float*)q.y, q.idx, p)) return 1;
        const int per_pass = 256 * dn_units(q.g.Cout, q.g.W);
        int ns = 1;
        while (ns < S && p.lds(ns + 1) <= 64 * 1024 && (ns + 1) * p.units <= per_pass) ++ns;
        if (p.lds(ns) > 64 * 1024 || ns * p.units > per_pass) return 1;
        dn_fill(c->job[j], p, q.g, (const float*)q.w, (const float*)q.dy, (const float*)q.y, q.idx, (float*)q.dx, q.act,
                q.alpha, ns);
        c->code[j] = p.shape * 4 + p.mode;
        if (p.lds(ns) > bytes) bytes = p.lds(ns);
    }
    c->n_jobs = n_jobs;
    c->S = S;
    *lds_bytes = bytes;
    return 0;
}

// Several layers' input gradients as one launch (dgrad_chain_k).  launch == false: only tell whether the list qualifies.
// 0: launched / qualifies; 1: a layer is outside the family, the batch sizes differ or the list is too long; < 0: error.
int launch_dgrad_chain(const sv_dgrad_job* jobs, int n_jobs, bool launch, hipStream_t st) {
    DnChainArgs c;
    int grid;
    size_t bytes;
    if (plan_dgrad_chain(jobs, n_jobs, 0, &c, &grid, &bytes)) return 1;
    if (!launch) return 0;
    hipLaunchKernelGGL(dgrad_chain_k, dim3((unsigned)grid), dim3(256), bytes, st, c);
    return check_launch("dgrad_chain");
}

}  // namespace sv

#ifdef SV_TRACE
extern "C" int sv_debug_set_dn_trace(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(sv::g_dn_trace), &buf, sizeof(buf));
}
#endif
