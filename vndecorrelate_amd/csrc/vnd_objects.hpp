// vnd_objects.hpp - includes, the error channel and the two objects of the C ABI (context, tap table).
#pragma once
#include "vnd_kernels.hpp"
#include "vnd_epilogue.hpp"
#include "vnd_moments.hpp"
#include "vnd_haas.hpp"
#include "vnd_win.hpp"
#include <atomic>
#include <functional>
#include "../../include/vnd_amd.h"
#include "../../include/vnd_amd_internal.h"

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace vnd;

// Every entry point works on its context's device whatever the caller's current device is, and leaves the caller's current
// device as it found it (a torch user's next allocation must not land on another GPU because this library was called).
struct DeviceScope {
    int prev = -1;
    bool ok = true;
    explicit DeviceScope(int dev)
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); ok = hipSetDevice(dev) == hipSuccess; return; }
        if (cur != dev) { ok = hipSetDevice(dev) == hipSuccess; if (ok) prev = cur; }
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// ------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------
static thread_local std::string g_err;

static vnd_status fail(vnd_status st, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return st;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(VND_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                           \
    } while (0)

// ------------------------------------------------------------------------------
// objects
// ------------------------------------------------------------------------------
struct vnd_ctx {
    int device = 0;
    hipDeviceProp_t prop{};
    int lds_limit = 65536;        // bytes of LDS one workgroup may use
    hipStream_t stream = nullptr; // used by the *_host entry points
    hipStream_t stream2 = nullptr;    // second lane of the chunked host pipeline
    std::vector<hipEvent_t> up_events;    // "piece k is on the device" marks of the time-chunked pipeline (made on first use)
    float *scratch_x = nullptr, *scratch_y = nullptr;
    size_t scratch_elems = 0;
    char *work = nullptr;         // grow-only workspace of the *_host entry points
    size_t work_bytes = 0;
    int variant = -1;
    int variant_nofuse = 0;       // tuning: 1 = keep the decorrelate epilogue as separate passes
    // One *_host call at a time per context: they share the stream, the staging buffers and the
    // workspace.  The reference's functions are re-entrant (decorrelation.py:630-660), and ctypes /
    // cgo / JNI callers run without a global lock, so the library serialises them itself.
    std::mutex host_mutex;
    // kernels already opted in to > 64 KiB of dynamic LDS on THIS context's device
    // (hipFuncSetAttribute applies to the current device's copy of the function)
    std::mutex raised_mutex;
    std::map<const void *, size_t> raised;      // kernel -> dynamic LDS bytes it has been allowed
    // pacing slots of the window kernel (vnd_win_kernel.inc, VWArgs::pace): [2048 CU indices][2] tile counters, allocated and zeroed
    // by vnd_ctx_create - never in a launch path (a *_dev call may sit inside a stream capture); read-only here afterwards
    unsigned *pace = nullptr;
};

typedef std::lock_guard<std::mutex> HostLock;

struct vnd_taps {
    vnd_ctx *ctx = nullptr;
    int32_t C = 0, total = 0, total_segs = 0, max_index = 0, apply_gain = 0;
    bool has_seg = false, has_flags = false;
    bool unit_weights = false;    // every weight is +-1: x*w is exact, so fma(x, w, acc) == acc + x*w bit for bit
    bool nonfinite = false;       // an inf/NaN weight: only the direct kernel drops (rather than zero-fills) the tail terms
    bool lds_images = true;       // false: indices too large for the LDS kernels' byte offsets (direct kernel only)
    std::vector<int32_t> tap_off, idx, seg_off, seg_end;
    std::vector<float> w, seg_gain;
    std::vector<uint8_t> flags;
    // device image
    Tap *d_taps = nullptr;
    FastTap *d_taps_fast = nullptr, *d_taps_ord = nullptr;
    int32_t *d_fast_off = nullptr, *d_fast_even = nullptr;
    int32_t *d_tap_off = nullptr, *d_seg_off = nullptr, *d_seg_end = nullptr;
    float *d_seg_gain = nullptr;
    uint8_t *d_flags = nullptr;
    // fast mode, specialised per table (vnd_spec.hpp): modules are compiled on first use
    SpecTable spec_table;          // effective weights (segment gain folded in)
    bool spec_ok = false;          // the table is within the specialised kernel's scope
    bool spec_exact_ok = false;    // ... also in VND_MODE_EXACT (no empty segment)
    bool win_exact_pays = false;   // ... and its exact mode takes the window form (stereo tables)
    std::mutex spec_mutex;                    // guards spec_modules (never held across a hipRTC build)
    std::condition_variable spec_built;       // a module left its `building` state
    std::map<SpecConfig, std::unique_ptr<SpecModule>> spec_modules;
};

