// C-ABI entry points for the convolution family and library-wide helpers.
#include <stdarg.h>

#include <atomic>

#include "common.h"

namespace sv {

thread_local char g_last_error[512] = "";

// ---- plan-time options: a process-wide table of ints, -1 = the planner's own choice
static std::atomic<int> g_opt[OPT_COUNT];
static const char* const kOptNames[OPT_COUNT] = {
    "dgrad_nhwc", "dgrad_phase", "dgrad_pair", "dgrad_grid", "dgrad_valu", "wgrad_nhwc", "wgrad_valu", "netfwd_generic",
    "netfwd_pair", "netfwd_phase", "netfwd_valu", "netfwd_s", "netfwd_grid", "small_convt", "s1_mfma", "wgrad_stored",
    "dgrad_chain"};
static const bool g_opt_init = [] {
    for (int i = 0; i < OPT_COUNT; ++i) g_opt[i].store(-1, std::memory_order_relaxed);
    return true;
}();
int opt(int id) { return g_opt[id].load(std::memory_order_relaxed); }
static int opt_index(const char* name) {
    if (name != nullptr)
        for (int i = 0; i < OPT_COUNT; ++i)
            if (strcmp(name, kOptNames[i]) == 0) return i;
    return -1;
}

// conv_direct.hip
int validate_geom(const sv_conv_geom* g, const char* who);
size_t conv_direct_scratch(const sv_conv_geom& g, int mode);
int launch_fwd_direct(const sv_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act,
                      float alpha, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0);
int launch_dgrad_direct(const sv_conv_geom& g, const float* dz, const float* w, const float* bias, float* dx,
                        int act, float alpha, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0);
size_t wgrad_direct_workspace(const sv_conv_geom& g);
int launch_wgrad_direct(const sv_conv_geom& g, const float* x, const float* dz, float* dw, int accumulate, void* ws,
                        size_t ws_bytes, hipStream_t st, const float* xmask = nullptr);
bool conv_masked_supported(const sv_conv_geom& g);
int launch_fwd_masked(const sv_conv_geom& g, const float* x, const float* mask, const float* w, const float* bias, float* y,
                      int act, float alpha, hipStream_t st, void* ws, size_t ws_bytes);
// bf16 storage: activations / filters / activation gradients bf16, bias and kernel gradients f32
int launch_fwd_bf16(const sv_conv_geom& g, const void* x, const void* w, const float* bias, void* y, int act, float alpha,
                    hipStream_t st, void* ws, size_t ws_bytes);
int launch_dgrad_bf16(const sv_conv_geom& g, const void* dz, const void* w, const float* bias, void* dx, int act,
                      float alpha, hipStream_t st, void* ws, size_t ws_bytes);
int launch_wgrad_bf16(const sv_conv_geom& g, const void* x, const void* dz, float* dw, int accumulate, void* ws,
                      size_t ws_bytes, hipStream_t st);

// conv_tiled.hip
bool tiled_s1_supported(int N, int Hs, int Ws, int Cin, int Cout, int K, int UP);
int launch_tiled_s1(const float* x, const float* w, const float* bias, float* y, uint8_t* idx, int N, int Hs, int Ws,
                    int Cin, int Cout, int K, int UP, int wflip, int epi, int act, float alpha, hipStream_t st,
                    int zmode = 0, const float* zy = nullptr, const uint8_t* zidx = nullptr, int zact = 0,
                    float zalpha = 0.f);
size_t tiled_wgrad_workspace(int N, int Hs, int Ws, int Cin, int Cout, int K, int UP);
bool tiled_wgrad_supported(int N, int Hs, int Ws, int Cin, int Cout, int K, int UP);
int launch_tiled_wgrad(const float* x, const float* dz, float* dw, float* db, int accumulate, void* ws,
                       size_t ws_bytes, int N, int Hs, int Ws, int Cin, int Cout, int K, int UP, hipStream_t st,
                       int zmode = 0, const float* zy = nullptr, const uint8_t* zidx = nullptr, int zact = 0,
                       float zalpha = 0.f, int defer = 0, int with_bias = 0);
int launch_wgrad_finish(const sv_wgrad_job* jobs, int n_jobs, hipStream_t st);
int launch_wgrad_batch(const sv_wgrad_batch_job* jobs, int n_jobs, int reduce, hipStream_t st);
// conv_s1_mfma.hip: stride-1 SAME convs on the matrix cores with channel-last LDS images (1: geometry not covered)
bool s1_mfma_supported(const sv_conv_geom& g, int which);
int launch_s1_fwd(const sv_conv_geom& g, const float* x, const float* w, const float* bias, float* y, uint8_t* idx, int act,
                  float alpha, hipStream_t st);
int launch_s1_dgrad(const sv_conv_geom& g, const float* dz, const float* w, float* dx, hipStream_t st);
// wgrad_s1_mfma.hip
bool wgrad_s1_mfma_supported(const sv_conv_geom& g);
size_t wgrad_s1_mfma_workspace(const sv_conv_geom& g);
int launch_wgrad_s1_mfma(const sv_conv_geom& g, const float* x, const float* dz, float* dw, float* db, int accumulate,
                         void* ws, size_t ws_bytes, hipStream_t st);
int launch_form_dz(const float* dy, const float* y, const uint8_t* idx, float* dz, int N, int H, int W, int C, int act,
                   float alpha, hipStream_t st);
bool wgrad_nhwc_batch_ok(const sv_wgrad_batch_job* jobs, int n_jobs);  // wgrad_nhwc.hip
int launch_wgrad_nhwc_batch(const sv_wgrad_batch_job* jobs, int n_jobs, hipStream_t st, sv_grad_slabs* slabs_out);
int launch_dgrad_chain(const sv_dgrad_job* jobs, int n_jobs, bool launch, hipStream_t st);
int launch_dgrad_nhwc(const sv_conv_geom& g, const float* w, const float* dy, const float* y, const uint8_t* idx, float* dx,
                      int act, float alpha, hipStream_t st);

// Stride-1 SAME conv with an odd square kernel and an isotropic folded upsampling: the LDS-tiled family.
static bool tiled_shape(const sv_conv_geom& g) {
    return g.KH == g.KW && (g.KH == 3 || g.KH == 5) && g.SH == 1 && g.SW == 1 && g.pad_t == (g.KH - 1) / 2 &&
           g.pad_l == (g.KW - 1) / 2 && g.Ho == g.H && g.Wo == g.W && g.up_h == g.up_w && (g.up_h == 1 || g.up_h == 2);
}

// Whether a stride-1 SAME conv goes to the channel-last matrix-core kernel rather than the planar vector kernels: from 16
// channels on both sides its 16-wide tiles are full (option s1_mfma: 1 forces it wherever it applies, 0 turns it off).
static bool prefer_s1(const sv_conv_geom& g) {
    const int o = opt(OPT_S1_MFMA);
    if (o == 0) return false;
    if (o == 1) return true;
    // the two 16 / 32-channel layers of the run_model.py default stack have kernels built for their shape
    // (dgrad_nhwc.hip, wgrad_nhwc.hip; measured about twice as fast there)
    if (g.KH == 3 && g.W == 8 && ((g.Cin == 16 && g.Cout == 32) || (g.Cin == 32 && g.Cout == 16))) return false;
    return g.Cin % 16 == 0 && g.Cout % 16 == 0;
}

// The backward of such a layer through the matrix-core kernels: dz = dy * act'(y) (scattered through the pooling argmax)
// is formed ONCE in the workspace and feeds both the input-gradient and the kernel-gradient kernel.
static size_t s1_bwd_dz_bytes(const sv_conv_geom& g) {
    return (((size_t)g.N * g.H * g.W * g.Cout * sizeof(float)) + 255) & ~(size_t)255;
}
static bool s1_bwd_ok(const sv_conv_geom& g) {
    return prefer_s1(g) && s1_mfma_supported(g, 2) && wgrad_s1_mfma_supported(g);
}
static size_t s1_bwd_workspace(const sv_conv_geom& g) { return s1_bwd_dz_bytes(g) + wgrad_s1_mfma_workspace(g); }

}  // namespace sv

using namespace sv;

extern "C" {

const char* sv_last_error(void) { return g_last_error; }

int sv_version(void) { return 1; }

int sv_set_option(const char* name, int value) {
    const int i = opt_index(name);
    SV_REQUIRE(i >= 0, "sv_set_option: unknown option '%s'", name ? name : "(null)");
    g_opt[i].store(value, std::memory_order_relaxed);
    return SV_OK;
}

int sv_get_option(const char* name, int* value) {
    const int i = opt_index(name);
    SV_REQUIRE(i >= 0 && value != nullptr, "sv_get_option: unknown option '%s' or null result", name ? name : "(null)");
    *value = opt(i);
    return SV_OK;
}

int sv_conv2d_fwd(const sv_conv_geom* g, const void* x, const void* w, const void* bias, void* y, int act,
                  float alpha, int dtype, sv_stream_t stream) {
    return sv_conv2d_fwd_ws(g, x, w, bias, y, act, alpha, nullptr, 0, dtype, stream);
}

size_t sv_conv2d_scratch_bytes(const sv_conv_geom* g, int op) {
    if (validate_geom(g, "sv_conv2d_scratch_bytes") || op < 0 || op > 3) return 0;
    if (tiled_shape(*g) || g->up_h != 1 || g->up_w != 1) return 0;
    // a transposed conv's forward is the input-gradient kernel of the described conv and vice versa
    return conv_direct_scratch(*g, (op == SV_CONV_OP_DGRAD || op == SV_CONV_OP_TRANSPOSE_FWD) ? 1 : 0);
}

int sv_conv2d_fwd_ws(const sv_conv_geom* g, const void* x, const void* w, const void* bias, void* y, int act,
                     float alpha, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_conv2d_fwd");
    int rc = validate_geom(g, "sv_conv2d_fwd");
    if (rc) return rc;
    SV_REQUIRE(x && w && y, "sv_conv2d_fwd: null tensor");
    if (dtype == SV_BF16) {
        SV_REQUIRE(g->up_h == 1 && g->up_w == 1, "sv_conv2d_fwd: folded upsampling is f32 only");
        return launch_fwd_bf16(*g, x, w, (const float*)bias, y, act, alpha, to_stream(stream), workspace, workspace_bytes);
    }
    if (tiled_shape(*g)) {
        if (prefer_s1(*g)) {
            rc = launch_s1_fwd(*g, (const float*)x, (const float*)w, (const float*)bias, (float*)y, nullptr, act, alpha,
                               to_stream(stream));
            if (rc <= 0) return rc;
        }
        rc = launch_tiled_s1((const float*)x, (const float*)w, (const float*)bias, (float*)y, nullptr, g->N,
                             g->H / g->up_h, g->W / g->up_w, g->Cin, g->Cout, g->KH, g->up_h, 0, 0, act, alpha,
                             to_stream(stream));
        if (rc <= 0) return rc;
    }
    return launch_fwd_direct(*g, (const float*)x, (const float*)w, (const float*)bias, (float*)y, act, alpha,
                             to_stream(stream), workspace, workspace_bytes);
}

int sv_conv2d_dgrad(const sv_conv_geom* g, const void* dz, const void* w, void* dx, int dtype, sv_stream_t stream) {
    return sv_conv2d_dgrad_ws(g, dz, w, dx, nullptr, 0, dtype, stream);
}

int sv_conv2d_dgrad_ws(const sv_conv_geom* g, const void* dz, const void* w, void* dx, void* workspace,
                       size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_conv2d_dgrad");
    int rc = validate_geom(g, "sv_conv2d_dgrad");
    if (rc) return rc;
    SV_REQUIRE(dz && w && dx, "sv_conv2d_dgrad: null tensor");
    if (dtype == SV_BF16) {
        SV_REQUIRE(g->up_h == 1 && g->up_w == 1, "sv_conv2d_dgrad: folded upsampling is f32 only");
        return launch_dgrad_bf16(*g, dz, w, nullptr, dx, SV_ACT_NONE, 0.f, to_stream(stream), workspace, workspace_bytes);
    }
    if (tiled_shape(*g)) {
        if (prefer_s1(*g)) {
            rc = launch_s1_dgrad(*g, (const float*)dz, (const float*)w, (float*)dx, to_stream(stream));
            if (rc <= 0) return rc;
        }
        // a conv of dz (Cout channels, logical resolution) with mirrored taps and swapped channel roles; a folded
        // upsampling turns into a 2x2 sum of the result
        rc = launch_tiled_s1((const float*)dz, (const float*)w, nullptr, (float*)dx, nullptr, g->N, g->H, g->W, g->Cout,
                             g->Cin, g->KH, 1, 1, g->up_h == 2 ? 2 : 0, SV_ACT_NONE, 0.f, to_stream(stream));
        if (rc <= 0) return rc;
    }
    return launch_dgrad_direct(*g, (const float*)dz, (const float*)w, nullptr, (float*)dx, SV_ACT_NONE, 0.f,
                               to_stream(stream), workspace, workspace_bytes);
}

int sv_conv2d_masked_supported(const sv_conv_geom* g) {
    if (validate_geom(g, "sv_conv2d_masked_supported")) return 0;
    return g->up_h == 1 && g->up_w == 1 && !tiled_shape(*g) && conv_masked_supported(*g) ? 1 : 0;
}

int sv_conv2d_fwd_masked(const sv_conv_geom* g, const void* x, const void* mask, const void* w, const void* bias, void* y,
                         int act, float alpha, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_conv2d_fwd_masked");
    int rc = validate_geom(g, "sv_conv2d_fwd_masked");
    if (rc) return rc;
    SV_REQUIRE(x && mask && w && y, "sv_conv2d_fwd_masked: null tensor");
    SV_REQUIRE(sv_conv2d_masked_supported(g), "sv_conv2d_fwd_masked: geometry outside the implicit-GEMM family");
    return launch_fwd_masked(*g, (const float*)x, (const float*)mask, (const float*)w, (const float*)bias, (float*)y, act,
                             alpha, to_stream(stream), workspace, workspace_bytes);
}

int sv_conv2d_wgrad_masked(const sv_conv_geom* g, const void* x, const void* mask, const void* dz, void* dw, int accumulate,
                           void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_conv2d_wgrad_masked");
    int rc = validate_geom(g, "sv_conv2d_wgrad_masked");
    if (rc) return rc;
    SV_REQUIRE(x && mask && dz && dw, "sv_conv2d_wgrad_masked: null tensor");
    SV_REQUIRE(sv_conv2d_masked_supported(g), "sv_conv2d_wgrad_masked: geometry outside the implicit-GEMM family");
    return launch_wgrad_direct(*g, (const float*)x, (const float*)dz, (float*)dw, accumulate, workspace, workspace_bytes,
                               to_stream(stream), (const float*)mask);
}

size_t sv_conv2d_wgrad_workspace_bytes(const sv_conv_geom* g) {
    if (validate_geom(g, "sv_conv2d_wgrad_workspace_bytes")) return 0;
    size_t need = wgrad_direct_workspace(*g);
    if (tiled_shape(*g)) {
        const size_t t = tiled_wgrad_workspace(g->N, g->H / g->up_h, g->W / g->up_w, g->Cin, g->Cout, g->KH, g->up_h);
        if (t > need) need = t;
        if (s1_bwd_ok(*g) && s1_bwd_workspace(*g) > need) need = s1_bwd_workspace(*g);
    }
    return need;
}

int sv_conv2d_wgrad(const sv_conv_geom* g, const void* x, const void* dz, void* dw, int accumulate, void* workspace,
                    size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_conv2d_wgrad");
    int rc = validate_geom(g, "sv_conv2d_wgrad");
    if (rc) return rc;
    SV_REQUIRE(x && dz && dw, "sv_conv2d_wgrad: null tensor");
    if (dtype == SV_BF16) {
        SV_REQUIRE(g->up_h == 1 && g->up_w == 1, "sv_conv2d_wgrad: folded upsampling is f32 only");
        return launch_wgrad_bf16(*g, x, dz, (float*)dw, accumulate, workspace, workspace_bytes, to_stream(stream));
    }
    if (tiled_shape(*g)) {
        rc = launch_tiled_wgrad((const float*)x, (const float*)dz, (float*)dw, nullptr, accumulate, workspace,
                                workspace_bytes, g->N, g->H / g->up_h, g->W / g->up_w, g->Cin, g->Cout, g->KH, g->up_h,
                                to_stream(stream));
        if (rc <= 0) return rc;
    }
    return launch_wgrad_direct(*g, (const float*)x, (const float*)dz, (float*)dw, accumulate, workspace,
                               workspace_bytes, to_stream(stream));
}

// ---- fused forms of the stride-1 SAME family
int sv_conv2d_fused_supported(const sv_conv_geom* g, int pooled) {
    if (validate_geom(g, "sv_conv2d_fused_supported") || !tiled_shape(*g)) return 0;
    if (pooled && ((g->Ho | g->Wo) & 1)) return 0;
    const int Hs = g->H / g->up_h, Ws = g->W / g->up_w;
    return tiled_s1_supported(g->N, Hs, Ws, g->Cin, g->Cout, g->KH, g->up_h) &&
           tiled_s1_supported(g->N, g->H, g->W, g->Cout, g->Cin, g->KH, 1) &&
           tiled_wgrad_supported(g->N, Hs, Ws, g->Cin, g->Cout, g->KH, g->up_h);
}

int sv_conv2d_s1_mfma_bwd(const sv_conv_geom* g) {
    if (validate_geom(g, "sv_conv2d_s1_mfma_bwd") || !tiled_shape(*g)) return 0;
    return s1_bwd_ok(*g) ? 1 : 0;
}

int sv_conv2d_pool_fwd(const sv_conv_geom* g, const void* x, const void* w, const void* bias, void* y_pooled,
                       uint8_t* idx, int act, float alpha, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_conv2d_pool_fwd");
    int rc = validate_geom(g, "sv_conv2d_pool_fwd");
    if (rc) return rc;
    SV_REQUIRE(x && w && y_pooled, "sv_conv2d_pool_fwd: null tensor");
    SV_REQUIRE(sv_conv2d_fused_supported(g, 1), "sv_conv2d_pool_fwd: shape is outside the fused family");
    SV_REQUIRE(idx != nullptr, "sv_conv2d_pool_fwd: null argmax tensor");
    if (prefer_s1(*g)) {
        rc = launch_s1_fwd(*g, (const float*)x, (const float*)w, (const float*)bias, (float*)y_pooled, idx, act, alpha,
                           to_stream(stream));
        if (rc <= 0) return rc;
    }
    rc = launch_tiled_s1((const float*)x, (const float*)w, (const float*)bias, (float*)y_pooled, idx, g->N,
                         g->H / g->up_h, g->W / g->up_w, g->Cin, g->Cout, g->KH, g->up_h, 0, 1, act, alpha,
                         to_stream(stream));
    return rc > 0 ? fail(SV_E_UNSUPPORTED, "sv_conv2d_pool_fwd: shape not covered") : rc;
}

int sv_conv2d_bwd(const sv_conv_geom* g, const void* x, const void* w, const void* dy, const void* y,
                  const uint8_t* idx, void* dx, void* dw, void* db, int act, float alpha, void* workspace,
                  size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_conv2d_bwd");
    int rc = validate_geom(g, "sv_conv2d_bwd");
    if (rc) return rc;
    SV_REQUIRE(w && dy && y, "sv_conv2d_bwd: null tensor");
    SV_REQUIRE(dw != nullptr || dx != nullptr, "sv_conv2d_bwd: nothing to compute (dw and dx are both NULL)");
    SV_REQUIRE(dw == nullptr || x != nullptr, "sv_conv2d_bwd: the kernel gradient needs x");
    const int pooled = idx != nullptr;
    SV_REQUIRE(sv_conv2d_fused_supported(g, pooled), "sv_conv2d_bwd: shape is outside the fused family");
    if (s1_bwd_ok(*g) && workspace != nullptr && workspace_bytes >= s1_bwd_workspace(*g) &&
        (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) {
        float* dz = (float*)workspace;
        rc = launch_form_dz((const float*)dy, (const float*)y, idx, dz, g->N, g->H, g->W, g->Cout, act, alpha, to_stream(stream));
        if (rc) return rc;
        if (dx != nullptr) {
            rc = launch_s1_dgrad(*g, dz, (const float*)w, (float*)dx, to_stream(stream));
            if (rc) return rc > 0 ? fail(SV_E_UNSUPPORTED, "sv_conv2d_bwd: input gradient not covered") : rc;
        }
        if (dw != nullptr) {
            rc = launch_wgrad_s1_mfma(*g, (const float*)x, dz, (float*)dw, (float*)db, 1, (char*)workspace + s1_bwd_dz_bytes(*g),
                                      workspace_bytes - s1_bwd_dz_bytes(*g), to_stream(stream));
            if (rc) return rc > 0 ? fail(SV_E_UNSUPPORTED, "sv_conv2d_bwd: kernel gradient not covered") : rc;
        }
        return SV_OK;
    }
    const int zmode = pooled ? 2 : 1;
    if (dw != nullptr) {
        rc = launch_tiled_wgrad((const float*)x, (const float*)dy, (float*)dw, (float*)db, 1, workspace,
                                workspace_bytes, g->N, g->H / g->up_h, g->W / g->up_w, g->Cin, g->Cout, g->KH, g->up_h,
                                to_stream(stream), zmode, (const float*)y, idx, act, alpha);
        if (rc) return rc > 0 ? fail(SV_E_UNSUPPORTED, "sv_conv2d_bwd: shape not covered") : rc;
    }
    if (dx == nullptr) return SV_OK;
    rc = launch_dgrad_nhwc(*g, (const float*)w, (const float*)dy, (const float*)y, idx, (float*)dx, act, alpha,
                           to_stream(stream));  // the channel-last matrix-core kernel for the layer shapes it covers
    if (rc <= 0) return rc;
    rc = launch_tiled_s1((const float*)dy, (const float*)w, nullptr, (float*)dx, nullptr, g->N, g->H, g->W, g->Cout,
                         g->Cin, g->KH, 1, 1, g->up_h == 2 ? 2 : 0, SV_ACT_NONE, 0.f, to_stream(stream), zmode,
                         (const float*)y, idx, act, alpha);
    return rc > 0 ? fail(SV_E_UNSUPPORTED, "sv_conv2d_bwd: shape not covered") : rc;
}

static bool dgrad_chain_jobs_valid(const sv_dgrad_job* jobs, int n_jobs) {
    if (jobs == nullptr || n_jobs < 1 || n_jobs > 8) return false;
    for (int j = 0; j < n_jobs; ++j) {
        if (validate_geom(&jobs[j].g, "sv_conv2d_dgrad_chain")) return false;
        if (!jobs[j].w || !jobs[j].dy || !jobs[j].y || !jobs[j].dx) return false;
    }
    return true;
}

int sv_conv2d_dgrad_chain_supported(const sv_dgrad_job* jobs, int n_jobs) {
    return dgrad_chain_jobs_valid(jobs, n_jobs) && launch_dgrad_chain(jobs, n_jobs, false, nullptr) == 0 ? 1 : 0;
}

int sv_conv2d_dgrad_chain(const sv_dgrad_job* jobs, int n_jobs, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_conv2d_dgrad_chain");
    SV_REQUIRE(dgrad_chain_jobs_valid(jobs, n_jobs), "sv_conv2d_dgrad_chain: bad job list (1..8 jobs, tensors set)");
    const int rc = launch_dgrad_chain(jobs, n_jobs, true, to_stream(stream));
    return rc > 0 ? fail(SV_E_UNSUPPORTED, "sv_conv2d_dgrad_chain: a job is outside the few-channel 3x3 family, or the "
                                           "batch sizes differ") : rc;
}

int sv_conv2d_bwd_partial(const sv_conv_geom* g, const void* x, const void* dy, const void* y, const uint8_t* idx,
                          int act, float alpha, int with_bias, void* workspace, size_t workspace_bytes, int dtype,
                          sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_conv2d_bwd_partial");
    int rc = validate_geom(g, "sv_conv2d_bwd_partial");
    if (rc) return rc;
    SV_REQUIRE(x && dy && y, "sv_conv2d_bwd_partial: null tensor");
    const int pooled = idx != nullptr;
    SV_REQUIRE(sv_conv2d_fused_supported(g, pooled), "sv_conv2d_bwd_partial: shape is outside the fused family");
    rc = launch_tiled_wgrad((const float*)x, (const float*)dy, nullptr, nullptr, 1, workspace, workspace_bytes, g->N,
                            g->H / g->up_h, g->W / g->up_w, g->Cin, g->Cout, g->KH, g->up_h, to_stream(stream),
                            pooled ? 2 : 1, (const float*)y, idx, act, alpha, 1, with_bias);
    return rc > 0 ? fail(SV_E_UNSUPPORTED, "sv_conv2d_bwd_partial: shape not covered") : rc;
}

int sv_conv2d_wgrad_finish(const sv_wgrad_job* jobs, int n_jobs, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_conv2d_wgrad_finish");
    SV_REQUIRE(jobs != nullptr && n_jobs >= 0, "sv_conv2d_wgrad_finish: null job list");
    for (int j = 0; j < n_jobs; ++j) {
        int rc = validate_geom(&jobs[j].g, "sv_conv2d_wgrad_finish");
        if (rc) return rc;
        SV_REQUIRE(jobs[j].workspace && jobs[j].dw, "sv_conv2d_wgrad_finish: job %d: null workspace or dw", j);
        SV_REQUIRE(jobs[j].blocks >= 0 && jobs[j].blocks <= 512, "sv_conv2d_wgrad_finish: job %d: blocks must be 0..512", j);
        SV_REQUIRE(sv_conv2d_fused_supported(&jobs[j].g, 0), "sv_conv2d_wgrad_finish: job %d is outside the fused family", j);
    }
    for (int j0 = 0; j0 < n_jobs; j0 += 8) {
        const int n = n_jobs - j0 < 8 ? n_jobs - j0 : 8;
        int rc = launch_wgrad_finish(jobs + j0, n, to_stream(stream));
        if (rc) return rc;
    }
    return SV_OK;
}

int sv_conv2d_wgrad_batch(const sv_wgrad_batch_job* jobs, int n_jobs, int reduce, int dtype, sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_conv2d_wgrad_batch");
    SV_REQUIRE(jobs != nullptr && n_jobs >= 0, "sv_conv2d_wgrad_batch: null job list");
    for (int j = 0; j < n_jobs; ++j) {
        int rc = validate_geom(&jobs[j].g, "sv_conv2d_wgrad_batch");
        if (rc) return rc;
        SV_REQUIRE(jobs[j].x && jobs[j].dy && jobs[j].y && jobs[j].dw, "sv_conv2d_wgrad_batch: job %d: null tensor", j);
        SV_REQUIRE(jobs[j].blocks >= 0 && jobs[j].blocks <= 512, "sv_conv2d_wgrad_batch: job %d: blocks must be 0..512", j);
        SV_REQUIRE(sv_conv2d_fused_supported(&jobs[j].g, jobs[j].idx != nullptr),
                   "sv_conv2d_wgrad_batch: job %d is outside the fused family", j);
    }
    for (int j0 = 0; j0 < n_jobs; j0 += 8) {
        const int n = n_jobs - j0 < 8 ? n_jobs - j0 : 8;
        int rc = launch_wgrad_batch(jobs + j0, n, reduce, to_stream(stream));
        if (rc) return rc;
    }
    return SV_OK;
}

static bool wgrad_batch_jobs_valid(const sv_wgrad_batch_job* jobs, int n_jobs) {
    if (jobs == nullptr || n_jobs < 1 || n_jobs > 8) return false;
    for (int j = 0; j < n_jobs; ++j) {
        if (validate_geom(&jobs[j].g, "sv_conv2d_wgrad_batch_slabs")) return false;
        if (!(jobs[j].x && jobs[j].dy && jobs[j].y && jobs[j].dw)) return false;
        if (jobs[j].blocks < 0 || jobs[j].blocks > 512) return false;
    }
    return true;
}

int sv_conv2d_wgrad_batch_slabs_supported(const sv_wgrad_batch_job* jobs, int n_jobs) {
    return wgrad_batch_jobs_valid(jobs, n_jobs) && wgrad_nhwc_batch_ok(jobs, n_jobs) ? 1 : 0;
}

int sv_conv2d_wgrad_batch_slabs(const sv_wgrad_batch_job* jobs, int n_jobs, sv_grad_slabs* slabs_out, int dtype,
                                sv_stream_t stream) {
    SV_F32_ONLY(dtype, "sv_conv2d_wgrad_batch_slabs");
    SV_REQUIRE(slabs_out != nullptr, "sv_conv2d_wgrad_batch_slabs: null result table");
    SV_REQUIRE(wgrad_batch_jobs_valid(jobs, n_jobs), "sv_conv2d_wgrad_batch_slabs: bad job list (1..8 jobs, tensors set)");
    const int rc = launch_wgrad_nhwc_batch(jobs, n_jobs, to_stream(stream), slabs_out);
    return rc > 0 ? fail(SV_E_UNSUPPORTED, "sv_conv2d_wgrad_batch_slabs: a job is outside the few-channel 3x3 family") : rc;
}

// Conv2DTranspose forward is the input-gradient kernel of the described conv, plus bias and activation.
int sv_conv2d_transpose_fwd(const sv_conv_geom* g, const void* x, const void* w, const void* bias, void* y, int act,
                            float alpha, int dtype, sv_stream_t stream) {
    return sv_conv2d_transpose_fwd_ws(g, x, w, bias, y, act, alpha, nullptr, 0, dtype, stream);
}

int sv_conv2d_transpose_fwd_ws(const sv_conv_geom* g, const void* x, const void* w, const void* bias, void* y, int act,
                               float alpha, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_conv2d_transpose_fwd");
    int rc = validate_geom(g, "sv_conv2d_transpose_fwd");
    if (rc) return rc;
    SV_REQUIRE(g->up_h == 1 && g->up_w == 1, "sv_conv2d_transpose_fwd: folded upsampling is not defined here");
    SV_REQUIRE(x && w && y, "sv_conv2d_transpose_fwd: null tensor");
    if (dtype == SV_BF16)
        return launch_dgrad_bf16(*g, x, w, (const float*)bias, y, act, alpha, to_stream(stream), workspace, workspace_bytes);
    return launch_dgrad_direct(*g, (const float*)x, (const float*)w, (const float*)bias, (float*)y, act, alpha,
                               to_stream(stream), workspace, workspace_bytes);
}

// ... and its input-gradient is the described conv run forward without epilogue.
int sv_conv2d_transpose_dgrad(const sv_conv_geom* g, const void* dz, const void* w, void* dx, int dtype,
                              sv_stream_t stream) {
    return sv_conv2d_transpose_dgrad_ws(g, dz, w, dx, nullptr, 0, dtype, stream);
}

int sv_conv2d_transpose_dgrad_ws(const sv_conv_geom* g, const void* dz, const void* w, void* dx, void* workspace,
                                 size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_conv2d_transpose_dgrad");
    int rc = validate_geom(g, "sv_conv2d_transpose_dgrad");
    if (rc) return rc;
    SV_REQUIRE(g->up_h == 1 && g->up_w == 1, "sv_conv2d_transpose_dgrad: folded upsampling is not defined here");
    SV_REQUIRE(dz && w && dx, "sv_conv2d_transpose_dgrad: null tensor");
    if (dtype == SV_BF16)
        return launch_fwd_bf16(*g, dz, w, nullptr, dx, SV_ACT_NONE, 0.f, to_stream(stream), workspace, workspace_bytes);
    return launch_fwd_direct(*g, (const float*)dz, (const float*)w, nullptr, (float*)dx, SV_ACT_NONE, 0.f,
                             to_stream(stream), workspace, workspace_bytes);
}

// dw[kh,kw,co_T,ci_T] = sum dz_T[(H,W,Cin) side] * x_T[(Ho,Wo,Cout) side]: the conv kernel-gradient with the
// layer's output gradient in the role of the conv input and the layer's input in the role of the conv's dz.
int sv_conv2d_transpose_wgrad(const sv_conv_geom* g, const void* x, const void* dz, void* dw, int accumulate,
                              void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream) {
    SV_F32_OR_BF16(dtype, "sv_conv2d_transpose_wgrad");
    int rc = validate_geom(g, "sv_conv2d_transpose_wgrad");
    if (rc) return rc;
    SV_REQUIRE(g->up_h == 1 && g->up_w == 1, "sv_conv2d_transpose_wgrad: folded upsampling is not defined here");
    SV_REQUIRE(x && dz && dw, "sv_conv2d_transpose_wgrad: null tensor");
    if (dtype == SV_BF16)
        return launch_wgrad_bf16(*g, dz, x, (float*)dw, accumulate, workspace, workspace_bytes, to_stream(stream));
    return launch_wgrad_direct(*g, (const float*)dz, (const float*)x, (float*)dw, accumulate, workspace,
                               workspace_bytes, to_stream(stream));
}

// ---- deferred reductions
int sv_deferred_reduce_begin(void) {
    deferred_begin();
    return SV_OK;
}

int sv_deferred_reduce_flush(int launch, sv_stream_t stream) { return deferred_flush(to_stream(stream), launch != 0); }

}  // extern "C"
