// An entry point that allocates nothing: the kernel gradient of a convolution with its partial-sum slabs in a
// device array that is part of the library's code object.  For callers that cannot call hipMalloc (code under HIP
// graph capture, tight serving / fine-tuning loops, harnesses that forbid allocation) and have no workspace to pass.
#include <mutex>

#include "common.h"

namespace sv {

constexpr size_t kStaticScratchBytes = (size_t)16 << 20;  // 16 MiB per GPU, held from library load to process end

// The partial-sum slabs of sv_conv2d_wgrad_static.  Every GPU the library's code object is loaded on has its own copy
// at its own address.  In host code this name is only a handle for hipGetSymbolAddress: it is never passed to a
// kernel or to a memory routine.
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch` on each GPU, looked up once per GPU

// Device address of this GPU's copy of `scratch` (the current device's), or nullptr with the error text set.  The first
// call on a GPU looks the address up, which may load this file's code object and its 16 MiB variable: that must not
// happen while `st` is being captured into a graph, so there the lookup is refused (the caller resolves the address
// beforehand through sv_static_scratch_bytes()).
static int static_scratch_for_current_device(hipStream_t st, bool have_stream, void** out) {
    *out = nullptr;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDevices)
        return fail(SV_E_LAUNCH, "sv_conv2d_wgrad_static: no usable current device (%s, device %d)", hipGetErrorString(e), dev);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        if (have_stream) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)
                return fail(SV_E_INVALID,
                            "sv_conv2d_wgrad_static: first use on device %d inside a stream capture; call "
                            "sv_static_scratch_bytes() on this device before capturing", dev);
        }
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));  // resolves against the current device's code object
        if (e != hipSuccess || p == nullptr)
            return fail(SV_E_LAUNCH, "sv_conv2d_wgrad_static: hipGetSymbolAddress failed on device %d: %s", dev,
                        hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return SV_OK;
}

}  // namespace sv

using namespace sv;

extern "C" {

// Also resolves the array's address on the current device (once per GPU), so that a caller about to record
// sv_conv2d_wgrad_static into a graph can get the lookup out of the way: call this once per GPU before capturing.
size_t sv_static_scratch_bytes(void) {
    void* p = nullptr;
    (void)static_scratch_for_current_device(nullptr, false, &p);
    return kStaticScratchBytes;
}

int sv_conv2d_wgrad_static(const sv_conv_geom* g, const void* x, const void* dz, void* dw, int accumulate, int dtype,
                           sv_stream_t stream) {
    const size_t need = sv_conv2d_wgrad_workspace_bytes(g);  // validates the geometry (0: invalid)
    if (need == 0) return fail(SV_E_INVALID, "sv_conv2d_wgrad_static: invalid geometry");
    if (need > kStaticScratchBytes)
        return fail(SV_E_WORKSPACE,
                    "sv_conv2d_wgrad_static: this geometry needs %zu bytes of partial sums, the built-in scratch holds "
                    "%zu: use sv_conv2d_wgrad with a caller workspace",
                    need, kStaticScratchBytes);
    void* ptr = nullptr;
    const int rc0 = static_scratch_for_current_device(to_stream(stream), true, &ptr);
    if (rc0) return rc0;
    // Nothing in the slabs has to start at zero: every partial sum that the reduction uses is written by this call's
    // producer kernel first (padding slots are read and dropped), so the previous call's contents never show.
    // Inside a sv_deferred_reduce_begin / _flush bracket the sum of the slabs would be queued with its source in the
    // shared array, which the next call overwrites before the flush: the bracket is set aside for this call, the sum
    // launches at once.
    const bool was = deferred_suspend();
    const int rc = sv_conv2d_wgrad(g, x, dz, dw, accumulate, ptr, kStaticScratchBytes, dtype, stream);
    deferred_resume(was);
    return rc;
}

}  // extern "C"
