// vnd_amd.hip - C ABI (include/vnd_amd.h) over the gfx950 kernels.
// Host-side only decides launch geometry; all arithmetic lives in vnd_kernels.hpp.
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
    // pacing slots of the window kernel (vnd_win_kernel.inc, VWArgs::pace): [2048 CU indices][2] tile counters, made on first use
    std::mutex pace_mutex;
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
    std::mutex spec_mutex;
    std::map<SpecConfig, std::unique_ptr<SpecModule>> spec_modules;
};

// ------------------------------------------------------------------------------
// launch geometry
// ------------------------------------------------------------------------------
struct Plan {
    bool direct = false;
    bool bc = false;                            // mono input fanned out: one staged plane per workgroup
    int nt = 256, cg = 1, r = 1;                // r = frame pairs per lane; tile = 2 * nt * r frames
    int W = 0;
    size_t lds_bytes = 0;
    uint32_t nblocks = 0;
    int tiles = 0, groups = 1;
};

typedef void (*kern_t)(const KArgs);

constexpr int kOrderedThreads = 256;

template <int CG, int MODE>
static kern_t ordered_by_r(int r)
{
    switch (r) {
    case 1: return conv_ordered_kernel<kOrderedThreads, CG, 1, MODE>;
    case 2: return conv_ordered_kernel<kOrderedThreads, CG, 2, MODE>;
    case 4: return conv_ordered_kernel<kOrderedThreads, CG, 4, MODE>;
    case 8: return conv_ordered_kernel<kOrderedThreads, CG, 8, MODE>;
    default: return nullptr;
    }
}

static kern_t ordered_kernel(int cg, int r, int mode)
{
    const bool exact = mode == VND_MODE_EXACT;
    switch (cg) {
    case 1: return exact ? ordered_by_r<1, 0>(r) : ordered_by_r<1, 1>(r);
    case 2: return exact ? ordered_by_r<2, 0>(r) : ordered_by_r<2, 1>(r);
    default: return exact ? ordered_by_r<4, 0>(r) : ordered_by_r<4, 1>(r);
    }
}

template <int NT, int CG>
static kern_t fast_by_r(int r)
{
    switch (r) {
    case 1: return conv_fast_kernel<NT, CG, 1>;
    case 2: return conv_fast_kernel<NT, CG, 2>;
    case 3: return conv_fast_kernel<NT, CG, 3>;
    case 4: return conv_fast_kernel<NT, CG, 4>;
    case 6: return conv_fast_kernel<NT, CG, 6>;
    case 8: return conv_fast_kernel<NT, CG, 8>;
    default: return nullptr;
    }
}

template <int NT>
static kern_t fast_by_cg(int cg, int r)
{
    switch (cg) {
    case 1: return fast_by_r<NT, 1>(r);
    case 2: return fast_by_r<NT, 2>(r);
    default: return fast_by_r<NT, 4>(r);
    }
}

static kern_t fast_kernel(int nt, int cg, int r)
{
    switch (nt) {
    case 128: return fast_by_cg<128>(cg, r);
    case 256: return fast_by_cg<256>(cg, r);
    case 512: return fast_by_cg<512>(cg, r);
    default: return fast_by_cg<1024>(cg, r);
    }
}

// fan-out instantiations (mono input, two output channels per workgroup, 256 threads)
static kern_t fast_bc_kernel(int r, bool epi)
{
    switch (r) {
    case 1: return epi ? nullptr : conv_fast_kernel<256, 2, 1, false, true>;
    case 2: return epi ? conv_fast_kernel<256, 2, 2, true, true> : conv_fast_kernel<256, 2, 2, false, true>;
    case 3: return epi ? nullptr : conv_fast_kernel<256, 2, 3, false, true>;
    case 4: return epi ? conv_fast_kernel<256, 2, 4, true, true> : conv_fast_kernel<256, 2, 4, false, true>;
    case 6: return epi ? nullptr : conv_fast_kernel<256, 2, 6, false, true>;
    case 8: return epi ? conv_fast_kernel<256, 2, 8, true, true> : conv_fast_kernel<256, 2, 8, false, true>;
    default: return nullptr;
    }
}

// fused-epilogue instantiations of the fast kernel (256 threads)
template <int CG>
static kern_t fast_epi_by_r(int r)
{
    switch (r) {
    case 2: return conv_fast_kernel<256, CG, 2, true>;
    case 4: return conv_fast_kernel<256, CG, 4, true>;
    case 8: return conv_fast_kernel<256, CG, 8, true>;
    default: return nullptr;
    }
}

static kern_t fast_epi_kernel(const Plan &p)
{
    if (p.direct || p.nt != 256) return nullptr;
    if (p.bc) return fast_bc_kernel(p.r, true);       // its exchange buffer has room of its own
    const int T = 2 * p.nt * p.r;
    if (p.W - T < T / 2 + 1) return nullptr;          // the exchange buffer must fit the halo part
    switch (p.cg) {
    case 1: return fast_epi_by_r<1>(p.r);
    case 2: return fast_epi_by_r<2>(p.r);
    default: return fast_epi_by_r<4>(p.r);
    }
}

template <int MODE>
static kern_t ordered_bc_by_r(int r)
{
    switch (r) {
    case 1: return conv_ordered_kernel<kOrderedThreads, 2, 1, MODE, true>;
    case 2: return conv_ordered_kernel<kOrderedThreads, 2, 2, MODE, true>;
    case 4: return conv_ordered_kernel<kOrderedThreads, 2, 4, MODE, true>;
    case 8: return conv_ordered_kernel<kOrderedThreads, 2, 8, MODE, true>;
    default: return nullptr;
    }
}

// VND_MODE_EXACT on a table of +-1 weights (every class-path table) runs the fma kernels: the
// product is exact, so the single rounding of fma(x, +-1, acc) is the rounding of acc +- x, and
// the segment gain and segment add stay separate operations in both instantiations.
static int arithmetic_of(const vnd_taps *t, int mode)
{
    return (mode == VND_MODE_EXACT && t->unit_weights) ? VND_MODE_FMA : mode;
}

// ordered kernel with the pointwise epilogue applied before the store (two channels per workgroup)
template <int MODE, bool BC>
static kern_t ordered_epi_by_r(int r)
{
    switch (r) {
    case 1: return conv_ordered_kernel<kOrderedThreads, 2, 1, MODE, BC, true>;
    case 2: return conv_ordered_kernel<kOrderedThreads, 2, 2, MODE, BC, true>;
    case 4: return conv_ordered_kernel<kOrderedThreads, 2, 4, MODE, BC, true>;
    case 8: return conv_ordered_kernel<kOrderedThreads, 2, 8, MODE, BC, true>;
    default: return nullptr;
    }
}

static kern_t ordered_epi_kernel(const Plan &p, int arithmetic)
{
    if (p.direct || p.cg != 2 || p.nt != kOrderedThreads) return nullptr;
    const bool exact = arithmetic == VND_MODE_EXACT;
    if (p.bc) return exact ? ordered_epi_by_r<0, true>(p.r) : ordered_epi_by_r<1, true>(p.r);
    return exact ? ordered_epi_by_r<0, false>(p.r) : ordered_epi_by_r<1, false>(p.r);
}

static kern_t pick_kernel(const Plan &p, int mode)
{
    if (p.bc)
        return mode == VND_MODE_FAST ? fast_bc_kernel(p.r, false)
                                     : (mode == VND_MODE_EXACT ? ordered_bc_by_r<0>(p.r) : ordered_bc_by_r<1>(p.r));
    return mode == VND_MODE_FAST ? fast_kernel(p.nt, p.cg, p.r) : ordered_kernel(p.cg, p.r, mode);
}

static int halo_of(int max_index) { return (max_index + 2 + 15) & ~15; }

// bc: one plane, then the fast kernel's exchange buffer [cg][T/2 + 1] (rounded up to 16 B)
static size_t lds_need(int nt, int cg, int r, int max_index, bool bc = false)
{
    const size_t T = (size_t)2 * nt * r;
    if (bc) return ((T + halo_of(max_index)) + (((size_t)cg * (T / 2 + 1) + 3) & ~(size_t)3)) * sizeof(float);
    return (size_t)cg * (T + halo_of(max_index)) * sizeof(float);
}

// Tile sizes a mode supports, largest first (frame pairs per lane).
static const int kFastR[] = {8, 6, 4, 3, 2, 1};
static const int kOrderedR[] = {8, 4, 2, 1};

// variant word (vnd_set_variant): bits 0-4 frame pairs per lane (0 = auto),
// bits 8-11 channels per workgroup (0 = auto), bit 12 direct,
// bits 16-17 threads per workgroup of the fast kernel (0: 256, 1: 128, 2: 512, 3: 1024).
// Cx = interleaved input channels (== C for the plain call; a divisor of C for a fan-out).
static Plan make_plan(const vnd_ctx *ctx, const vnd_taps *t, int64_t batch, int64_t n, int C, int mode, int Cx)
{
    Plan p;
    const int v = ctx->variant;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const bool fast = mode == VND_MODE_FAST;
    // A term whose tap reaches past the end of the stream DROPS in the reference (decorrelation.py:656-658).
    // The LDS kernels read such a sample as 0.0f, which is the same thing for a finite weight only
    // (0 * inf = NaN), so a table with a non-finite weight takes the direct kernel, which tests the index.
    const bool force_direct = (v >= 0 && ((v >> 12) & 1)) || t->nonfinite || !t->lds_images;
    int cg = (v >= 0 && ((v >> 8) & 15)) ? ((v >> 8) & 15) : 0;
    if (cg == 0) cg = (C % 2 == 0) ? 2 : 1;
    if (C % cg != 0 || (cg != 1 && cg != 2 && cg != 4)) cg = 1;
    int nt = kOrderedThreads;
    if (fast) {
        const int sel = v >= 0 ? ((v >> 16) & 3) : 0;
        nt = sel == 1 ? 128 : sel == 2 ? 512 : sel == 3 ? 1024 : 256;
    }
    // a workgroup's cg output channels must come from cg consecutive input channels, or all from
    // the one channel of a mono input (bc: staged once)
    bool bc = false;
    if (Cx != C) {
        if (Cx == 1 && C % 2 == 0 && !(v >= 0 && ((v >> 8) & 15) == 1)) { bc = true; cg = 2; nt = 256; }
        else if (Cx % cg != 0) cg = (Cx % 2 == 0 && cg >= 2) ? 2 : 1;
    }
    const int *sizes = fast ? kFastR : kOrderedR;
    const int nsizes = fast ? (int)(sizeof kFastR / sizeof *kFastR) : (int)(sizeof kOrderedR / sizeof *kOrderedR);
    const size_t limit = (size_t)ctx->lds_limit - 1024;       // the kernels' static LDS (reduction scratch) shares the 160 KiB
    auto fits = [&](int r_) { return lds_need(nt, cg, r_, t->max_index, bc) <= limit; };

    int r = (v >= 0) ? (v & 31) : 0;
    if (r != 0) {
        bool known = false;
        for (int i = 0; i < nsizes; ++i) known |= sizes[i] == r;
        if (!known) r = 0;
    }
    if (r == 0) {
        // 4 pairs per lane (tile 2048) measured best wherever it leaves every CU >= 6 workgroups
        // (LDS-bound residency); smaller tiles for small problems, so the grid still fills the chip.
        r = 1;
        const size_t budget = limit / 4;
        for (int i = 0; i < nsizes; ++i) {
            if (sizes[i] > 4) continue;
            const int64_t T = (int64_t)2 * nt * sizes[i];
            const int64_t blocks = batch * ((n + T - 1) / T) * (C / cg);
            if (blocks >= (int64_t)cus * 6 && lds_need(nt, cg, sizes[i], t->max_index, bc) <= budget) { r = sizes[i]; break; }
        }
    }
    // shrink until the tile fits one workgroup's LDS at all
    while (!fits(r)) {
        if (bc) { bc = false; cg = 1; continue; }      // one plane per output channel, plain staging
        if (cg > 1) { cg /= 2; continue; }
        int smaller = 0;
        for (int i = 0; i < nsizes; ++i) if (sizes[i] < r) { smaller = sizes[i]; break; }
        if (smaller) { r = smaller; continue; }
        if (nt > 128 && fast) { nt /= 2; continue; }
        break;
    }
    if (force_direct || !fits(r)) {
        p.direct = true;
        const int64_t total = batch * n * C;
        int64_t blocks = (total + kDirectThreads - 1) / kDirectThreads;
        p.nblocks = (uint32_t)std::min<int64_t>(std::max<int64_t>(blocks, 1), (int64_t)cus * 32);
        return p;
    }
    const int64_t T = (int64_t)2 * nt * r;
    p.nt = nt; p.cg = cg; p.r = r; p.bc = bc;
    p.W = (int)T + halo_of(t->max_index);
    p.lds_bytes = lds_need(nt, cg, r, t->max_index, bc);
    p.tiles = (int)((n + T - 1) / T);
    p.groups = C / cg;
    p.nblocks = (uint32_t)(batch * p.tiles * p.groups);
    return p;
}

static vnd_status check_shape(const vnd_ctx *ctx, const vnd_taps *t, int64_t batch, int64_t n,
                              int32_t C, int32_t mode, int32_t Cx = 0)
{
    if (!ctx || !t) return fail(VND_ERR_INVALID, "null context or tap table");
    if (batch < 0 || n < 0) return fail(VND_ERR_INVALID, "negative batch or frame count");
    if (C != t->C)
        return fail(VND_ERR_INVALID, "signal has %d channels but the tap table has %d", C, t->C);
    if (t->ctx != ctx && t->ctx->device != ctx->device)
        return fail(VND_ERR_INVALID, "the tap table lives on device %d, the context on device %d", t->ctx->device,
                    ctx->device);
    if (Cx != 0 && (Cx < 0 || C % Cx != 0))
        return fail(VND_ERR_INVALID, "%d input channels do not divide the tap table's %d channels", Cx, C);
    if (mode != VND_MODE_EXACT && mode != VND_MODE_FMA && mode != VND_MODE_FAST)
        return fail(VND_ERR_INVALID, "unknown mode %d", mode);
    if (n > (int64_t)1 << 40 || batch * n * C / std::max<int64_t>(n, 1) > (int64_t)1 << 40)
        return fail(VND_ERR_UNSUPPORTED, "problem too large");
    return VND_OK;
}

// ------------------------------------------------------------------------------
// the specialised fast kernel (vnd_spec.hpp, vnd_spec_kernel.inc)
// ------------------------------------------------------------------------------
struct SpecPlan {
    bool use = false;
    bool eager = true;              // false: too small a launch to build the kernel for - taken only if its code object exists
    SpecConfig cfg;
    int tiles_total = 0, tiles_per_span = 0, spans = 0;
    uint32_t nblocks = 0, units = 0;
    // a small launch's CU chunks (window form, stereo): chunk_tiles consecutive tiles per CU, its first-dispatched workgroup takes
    // chunk_len0 of them, the second the rest (0: uniform spans)
    int chunk_tiles = 0, chunk_len0 = 0, chunks_per_stream = 0, cus_per_xcd = 0, stagger_ticks = 0;
    const char *why = "";           // when !use: the reason, for vnd_describe_launch
};

static bool spec_disabled_by_env()
{
    static const bool off = [] { const char *e = getenv("VND_SPEC"); return e && e[0] == '0'; }();
    return off;
}

struct EpiFuse {                 // non-null => launch the fused-epilogue instantiation
    double *partials;
    int ms_encode, use_width, normalize;
    float w_mid, w_side;
    double *sink = nullptr;      // moments sink: [tiles][groups][8]; the output is reduced, not written
    // exact RMS sums, block-parallel form: where the convolution may leave the per-block sums of squares ([batch][4][nblocks] doubles,
    // the predictions rms_par_tally_kernel starts from) - the window form's store phase has x and the finished y at hand; *blk_done
    // says whether it did (else rms_par_sum_kernel reads both arrays for them)
    double *blk_sum = nullptr;
    int nblocks = 0;
    int rows_major = 0;          // 1: [stream][block][x0 x1 y0 y1] - rows for epilogue_reduce_kernel (the fused fast stage's sums)
    int *path = nullptr;         // out: 0 a generic kernel ran, 1 the per-table kernel and it left the block sums, 2 the per-table kernel without them
};

// variant word, specialised kernel: bit 25 forces the generic kernel; bits 26-27 prefetch depth
// (0 = auto), bits 28-30 spans per resident slot ("rounds", 0 = auto); bits 0-4 = pairs per lane as ever;
// bits 20-22 shortest span in tiles (0 = auto, 8) and bit 23 "specialise however little work there
// is" - the two that let the tests drive span seams and tiny signals through this kernel.
static SpecPlan make_spec_plan(const vnd_ctx *ctx, const vnd_taps *t, const float *x, const float *y, int64_t batch,
                               int64_t n, int C, int Cx, int mode, const EpiFuse *epi)
{
    SpecPlan p;
    const int v = ctx->variant;
    if (mode != VND_MODE_FAST && mode != VND_MODE_EXACT) { p.why = "neither the fast nor the exact mode"; return p; }
    // a fused epilogue is within scope when it is the pointwise steps alone (no sums, no moments sink) on a stereo output:
    // they ride in the per-table kernels' store phase (VS_EPI)
    // (... with the normaliser's sums too where the caller offers room for per-block sums: the window form's store phase leaves them)
    const bool pointwise = epi != nullptr && (!epi->normalize || epi->blk_sum != nullptr) && epi->sink == nullptr && C == 2;
    // fan-out: a mono input through a stereo table is in scope (one LDS plane, VS_BC); wider fan-outs are not
    const bool bc = Cx == 1 && C == 2;
    if ((epi != nullptr && !pointwise) || (Cx != C && !bc)) { p.why = "fused epilogue or fan-out launch"; return p; }
    if (!(mode == VND_MODE_EXACT ? t->spec_exact_ok : t->spec_ok)) { p.why = "table outside the specialised kernel's scope"; return p; }
    // VND_MODE_EXACT specialises by default as well: with the shifted plane copies (odd offsets as aligned pairs) the per-table
    // kernel is ahead of the generic ordered one by 24 % on a function-path table, 37 % on a class-path one and 23-50 % on a mono
    // input fanned out (cfg2 pool; tools/exact_geometry_try.py, tools/fanout_spec_try.py).  VND_SPEC_EXACT=0 keeps the generic kernel.
    if (mode == VND_MODE_EXACT && !(v >= 0 && ((v >> 15) & 1))) {
        static const bool off = [] { const char *e = getenv("VND_SPEC_EXACT"); return e && e[0] == '0'; }();
        if (off) { p.why = "exact mode specialisation switched off"; return p; }
    }
    const bool force = v >= 0 && ((v >> 23) & 1);
    if (spec_disabled_by_env() || (v >= 0 && ((v >> 25) & 1))) { p.why = "disabled"; return p; }
    // access shape: 16 bytes per frame pair (stereo) or 8 per frame, from every stream's first sample
    const uintptr_t align = C == 2 ? 16 : 8, align_x = bc ? 8 : align;
    if (((uintptr_t)y & (align - 1)) || ((uintptr_t)x & (align_x - 1))) { p.why = "unaligned base"; return p; }
    if (batch > 1 && (((uint64_t)n * C * 4) % align != 0 || ((uint64_t)n * Cx * 4) % align_x != 0)) { p.why = "unaligned streams"; return p; }
    const int rr_hint = (v >= 0 && (v & 31) != 0 && (v & 31) <= 8) ? (v & 31) : 0;
    const int dd_hint = v >= 0 ? ((v >> 26) & 3) : 0;
    // variant bits 5-7: window form off (1), or 16 (2), 32 (3), 64 (4) frames per lane; VND_WIN_M: the default (32; 0 = pair-read kernel)
    const int win_env = spec_env("VND_WIN_M", 32);
    const int vw = v >= 0 ? ((v >> 5) & 7) : 0;
    const int win_m = vw == 1 ? 0 : (vw == 2 ? 16 : (vw == 3 ? 32 : (vw == 4 ? 64 : win_env)));
    // VND_MODE_EXACT in the window form: tables whose weights let the sign ride in the add (finite) - all in spec scope
    const int win_exact_env = spec_env("VND_WIN_EXACT", 1);       // 0: never, 1: where it pays (the table knows), 2: always
    const bool win_exact = vw >= 2 || win_exact_env == 2 || (win_exact_env == 1 && t->win_exact_pays);
    // 1536-frame tiles (cfg4's 32-tile streams included: 0.167 vs 0.179 ms) unless a span would be shorter than 12 of them
    for (int attempt = 0; attempt < 2; ++attempt) {
    // (wider signals - a workgroup per channel pair, 8 bytes per frame - measured best with the 1024-frame tiles)
    // the WINDOW form (vnd_win.hpp: a lane owns win_m consecutive frames and reads the union of its taps' windows once):
    // stereo outputs, fast mode
    bool picked = false;
    // (a mono input fanned out keeps the pair-read form unless forced: there the two channels' taps share the reads of the
    //  one plane at equal offsets, the window form makes a pass per channel - 0.163 against 0.169 ms for 128 x 10 s,
    //  tools/fanout_win_try.py)
    // (wider signals - a workgroup per channel PAIR, VW_C - keep the pair-read kernel unless variant bits 5-7 or VND_WIN_WIDE=1
    //  ask for the window form: there a workgroup moves 8 bytes of every frame, the memory pipeline's time per useful byte
    //  is 2-4x a stereo signal's and the window form's few waves per CU do not hide it - cfg5 0.54 ms against 0.45, while
    //  the same tables on planar channel pairs run 0.33 against 0.39: tools/c8_win_try.py, profiles/r03_cfg5_request_floor.txt)
    const int win_wide_env = spec_env("VND_WIN_WIDE", 0);
    const bool win_c = C == 2 || (C % 2 == 0 && (win_wide_env != 0 || vw >= 2));
    // signals of 4k channels: the window form on channel QUADS / OCTETS (VW_Q, vw_span_qc: a workgroup moves 16 / 32 bytes of every
    // frame, a wave per channel) - VND_WIN_QUAD=0 keeps the pair-read kernel (or, with VND_WIN_WIDE=1 / variant bits 5-7, the
    // window form on channel pairs)
    const bool win_quad = C % 4 == 0 && Cx == C && !pointwise && spec_env("VND_WIN_QUAD", 1) != 0;
    // (a geometry whose build failed or spilled is remembered in the table's module map: skipped, the next best taken)
    const bool nt_big = batch * n * C * (int64_t)sizeof(float) >= ((int64_t)spec_env("VND_NT_MIN_MB", 64) << 20);
    auto nt_stores_of = [&](const SpecConfig &c) {
        // a channel pair (or quad) is a piece of a frame: let L2 merge the pieces - unless the quad IS the frame
        if (C != 2 && !(c.win_q && C == 4 * c.win_q) && !spec_env("VND_FORCE_NT", 0)) return 0;
        return nt_big ? 1 : 0;
    };
    auto rejected = [&](const SpecConfig &c0) {
        SpecConfig c = c0;
        c.nt_stores = nt_stores_of(c0); c.exact = mode == VND_MODE_EXACT ? 1 : 0; c.epi = pointwise ? 1 : 0; c.bc = bc ? 1 : 0;
        std::lock_guard<std::mutex> g(const_cast<vnd_taps *>(t)->spec_mutex);
        auto it = t->spec_modules.find(c);
        return it != t->spec_modules.end() && it->second->failed;
    };
    const bool win_mode_ok = win_m > 0 && rr_hint == 0 && (mode == VND_MODE_FAST || win_exact || (win_quad && win_exact_env != 0));
    // (8k channels: two neighbouring quads - with 8 channels whole frames, whole cache lines - per workgroup of 512 lanes when that ring
    //  fits, else and for 4k channels a quad per workgroup of 256; a wave per CHANNEL, 32-frame runs: vw_span_qc)
    const int quad_m = vw >= 2 ? win_m : spec_env("VND_WIN_QUAD_M", 32);
    if (win_mode_ok && win_quad && C % 8 == 0 && spec_env("VND_WIN_OCTET", 1) != 0)
        picked = win_pick_config(t->spec_table, (size_t)ctx->lds_limit, quad_m, attempt == 1, false, &p.cfg, rejected, 2, false, mode == VND_MODE_EXACT);
    if (!picked && win_mode_ok && win_quad)
        picked = win_pick_config(t->spec_table, (size_t)ctx->lds_limit, quad_m, attempt == 1, false, &p.cfg, rejected, 1, false, mode == VND_MODE_EXACT);
    // plain stereo: the waves SPLIT over the two channels (VW_S, vw_span_s: a lane carries ONE channel's accumulators).
    // VND_WIN_SPLIT: 0 never; 1 (default) where it pays; 2 always, with the frames per lane of the plain form.
    //  * 32-frame runs, three waves per SIMD (three workgroups of 256 lanes per CU): cfg3 fast +2.0 / +2.4 % on two boxes, but
    //    cfg3 kappa 1 -5 %, cfg2 fast -3.5 %, exact modes -3 ... +4 %; 384 lanes (six waves on four SIMDs) -17 %: not taken;
    //  * 64-FRAME runs (half the LDS reads per FMA: every 16-byte window read costs the SIMD ~1.45 packed-FMA slots,
    //    profiles/r03_fp32_issue_rate.txt) fit two waves per SIMD only in this form: VND_MODE_EXACT on function-path tables
    //    +14-16 % at cfg3 (0.579 -> 0.497 ms), +6 % at cfg2 - taken there; class-path tables and the fast mode spill at 64
    //    frames (rejected builds fall back to the plain form) (tools/win_split_try.py, profiles/r03_split_waves.txt)
    // a mono input fanned out, fast mode: the plain form with ONE read stream for both output channels (win_taps_function_merged:
    // the two channels' taps lie almost alike, their windows' union is little more than one channel's - 1.48 B of LDS per FMA)
    if (!picked && win_mode_ok && bc && mode == VND_MODE_FAST && vw == 0)
        picked = win_pick_config(t->spec_table, (size_t)ctx->lds_limit, win_m, attempt == 1, true, &p.cfg, rejected);
    const int split_env = spec_env("VND_WIN_SPLIT", 1);
    // (a mono input fanned out rides the same form: its one channel staged into both plane sets, VW_BC - cfg1's shape 0.177 -> 0.15 ms
    //  for 128 x 10 s against the pair-read form, tools/fanout_win_try.py; VND_WIN_SPLIT_FANOUT=0 keeps that)
    const bool split_scope = C == 2 && (Cx == 2 || (bc && spec_env("VND_WIN_SPLIT_FANOUT", 1) != 0)) && !pointwise;
    //    In the FAST mode (E and P: 128 accumulator registers) the 64-frame split form needs its refill loaded late (VW_LATE: 15
    //    of a wave's 16 accesses per tile at the start of the store phase that consumes them, not a tile ahead) and the per-access
    //    constants kept out of the tile loop's registers: cfg3 +3-4.5 % (0.457 -> 0.437 ms), cfg2 +3.8 % (0.195 -> 0.188 ms,
    //    tools/split64_fast_probe.py); a table whose build spills all the same falls back to the plain 32-frame form
    const bool exact_now = mode == VND_MODE_EXACT;
    if (!picked && win_mode_ok && split_scope && split_env == 1 && vw == 0 &&
        ((exact_now && !t->spec_table.has_seg) || mode == VND_MODE_FAST))
        picked = win_pick_config(t->spec_table, (size_t)ctx->lds_limit, 64, attempt == 1, bc, &p.cfg, rejected, 0, true, exact_now);
    if (!picked && win_mode_ok && split_scope && split_env == 2)
        picked = win_pick_config(t->spec_table, (size_t)ctx->lds_limit, win_m, attempt == 1, bc, &p.cfg, rejected, 0, true, exact_now);
    if (!picked && win_mode_ok && win_c && (!bc || vw >= 2))
        picked = win_pick_config(t->spec_table, (size_t)ctx->lds_limit, win_m, attempt == 1, bc, &p.cfg, rejected);
    if (!picked && !spec_pick_config(t->spec_table, (size_t)ctx->lds_limit, rr_hint, dd_hint, &p.cfg, attempt == 1 || C != 2, bc, mode == VND_MODE_EXACT)) { p.why = "halo does not fit the ring"; return p; }
    const int64_t T = p.cfg.tile();
    const int64_t tiles_total = (n + T - 1) / T;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    // resident workgroups per CU: LDS-bound, within the 32 waves a CU holds
    const int64_t per_cu = p.cfg.win ? p.cfg.win_per_cu
                                     : std::min<int64_t>(std::min<int64_t>(16, 2048 / p.cfg.nt), (int64_t)(160 * 1024) / (int64_t)p.cfg.lds_bytes());
    const int64_t resident = (int64_t)cus * std::max<int64_t>(per_cu, 1);
    const int64_t units = batch * (p.cfg.win_q ? C / (4 * p.cfg.win_q) : C / 2);      // (stream, channel pair) - or channel quad / octet
    // a workgroup needs a span long enough to amortise filling its ring: 8 tiles when there are 16 and more per resident
    // slot; with less, shorter spans (down to 2 tiles) so that the chip still fills - a lone 60 s stream then runs 1.1x
    // (fast) to 1.75x (exact, 128 taps) faster than through the generic kernels, tools/single_stream_try.py - and below
    // about two million frames per channel pair the generic kernels (many small workgroups) stay ahead
    int64_t min_span = (v >= 0 && ((v >> 20) & 7)) ? ((v >> 20) & 7) : 8;
    if (!(v >= 0 && ((v >> 20) & 7)))
        min_span = std::min<int64_t>(8, std::max<int64_t>(2, units * tiles_total / (2 * resident)));
    const int64_t pair_frames = batch * (C / 2) * n;           // the work, in frames per channel pair (whatever a workgroup's unit is)
    if (!force && pair_frames < 2000000) { p.why = "too little work for persistent workgroups"; return p; }
    p.eager = force || pair_frames >= 12000000;                // enough work to be worth building the kernel for
    // Spans per stream: the workgroups are equally long, so the grid should fill the resident slots
    // a whole number of times ("rounds") - 1.5 rounds cost as much as 2.  Fewest spans (longest
    // rings) whose last round is at least 95 % full, else the fullest.
    const int rounds = (v >= 0 && ((v >> 28) & 7)) ? ((v >> 28) & 7) : 0;
    const int64_t max_spans = std::max<int64_t>(1, tiles_total / min_span);
    int64_t spans = 1;
    if (rounds > 0) {
        spans = std::min(std::max<int64_t>(1, resident * rounds / units), max_spans);
    } else {
        double best = -1.0;
        const int64_t limit = std::min<int64_t>(max_spans, std::max<int64_t>(1, 4 * resident / units + 1));
        for (int64_t sp = 1; sp <= limit; ++sp) {
            const int64_t per = (tiles_total + sp - 1) / sp;
            const int64_t wgs = units * ((tiles_total + per - 1) / per);
            const double fill = (double)wgs / (double)(((wgs + resident - 1) / resident) * resident);
            if (fill > best + 1e-9) { best = fill; spans = sp; }
            if (fill >= 0.95) { spans = sp; break; }
        }
    }
    int64_t per_span = (tiles_total + spans - 1) / spans;
    // descriptor offsets are 32-bit: keep a span (plus what it prefetches) under 2 GiB
    const int64_t max_tiles = ((int64_t)0x7fffffff / (T * C * 4)) - p.cfg.pp - p.cfg.dd - 1;
    if (max_tiles < 1) { p.why = "tile too large"; return p; }
    per_span = std::min(per_span, max_tiles);
    spans = (tiles_total + per_span - 1) / per_span;
    if (units * spans > 0x7fffffffLL) { p.why = "grid too large"; return p; }
    p.cfg.nt_stores = nt_stores_of(p.cfg);
    p.cfg.exact = mode == VND_MODE_EXACT ? 1 : 0;
    p.cfg.epi = pointwise ? 1 : 0;
    p.cfg.bc = bc ? 1 : 0;
    // exact mode counts VS_LA in steps of RR to 2*RR reads: the LDS queue holds 15, three steps fill it
    if (p.cfg.exact && !p.cfg.win && spec_env("VND_SPEC_LA", -1) < 0) p.cfg.la = 3;
    p.tiles_total = (int)tiles_total; p.tiles_per_span = (int)per_span; p.spans = (int)spans;
    // one round of workgroups: at most the resident slots, each walking units w, w + nblocks, ...
    p.units = (uint32_t)(units * spans);
    p.nblocks = (uint32_t)std::min<int64_t>(units * spans, resident);
    // ---- one round of workgroups that does not fill evenly: CU chunks ----------------------------------------------
    // cfg4's N = 8 shard (128 one-second streams: 768 tiles of 8192 frames) is 3 tiles per CU.  Uniform spans of 2 tiles make 384
    // workgroups: every CU gets one, half of them a second - and a CU's second workgroup runs in what the first leaves of the SIMDs
    // and the memory pipeline (phase stamps, profiles/r04_shard_timeline.txt: its tile period is 1.2x the first's), so those CUs
    // finish 4-5 us after the others.  Instead every CU takes a CHUNK of consecutive tiles of one stream and splits it between its
    // two co-resident workgroups - the longer piece to the one dispatched first.  The split minimises a small model of the two
    // (prologue 0.55 / 0.8 of a tile period, period 1 / 1.2); taken only when the model puts it ahead of the uniform plan.
    p.chunk_tiles = 0;
    if (p.cfg.win && C == 2 && per_cu >= 2 && cus % 8 == 0 && units <= cus && cus % units == 0 && units * spans <= resident &&
        !(v >= 0 && ((v >> 28) & 7)) && spec_env("VND_WIN_CHUNKS", 1) != 0) {
        const int64_t cps = cus / units;                              // chunks per stream: one per CU
        const int64_t w = (tiles_total + cps - 1) / cps;              // tiles per chunk
        auto cost2 = [](int64_t a, int64_t b) { return std::max(0.55 + (double)a, b > 0 ? 0.8 + 1.2 * (double)b : 0.0); };
        // the uniform plan: its workgroups land on the CUs in dispatch order - every CU one, then a second one on the first few
        const int64_t wgs = units * spans, doubled = std::max<int64_t>(0, wgs - cus);
        const double uniform = doubled > 0 ? cost2(per_span, per_span) : cost2(per_span, 0);
        int64_t best_len0 = 0;
        double best = 1e30;
        for (int64_t a0 = (w + 1) / 2; a0 <= w; ++a0) {
            const double c = cost2(a0, w - a0);
            if (c < best - 1e-9) { best = c; best_len0 = a0; }
        }
        const int len0_env = spec_env("VND_WIN_CHUNK_LEN0", 0);      // (tuning: force the split)
        if (len0_env > 0 && len0_env < w) { best_len0 = len0_env; best = -1.0; }
        if (w >= 2 && best < uniform - 1e-9 && best_len0 < w && w * (cps - 1) < tiles_total) {
            p.chunk_tiles = (int)w; p.chunk_len0 = (int)best_len0; p.chunks_per_stream = (int)cps; p.cus_per_xcd = cus / 8;
            p.stagger_ticks = std::max(0, spec_env("VND_WIN_STAGGER_TICKS", 300));      // 3 us: about the first workgroup's ring fill - the later one loads while that one computes (tools/ablate/run_r4b.sh, run_r4c.sh)
            p.units = (uint32_t)(2 * cus);
            p.nblocks = p.units;
        }
    }
    p.use = true;
    // (window form: 8192-frame tiles down to 3 per span - 256 one-second streams 42.6 us with them, 45.7 with 4096-frame
    //  tiles; at 2 per span - 128 such streams - the smaller tiles win, 26.3 against 28.1 us: tools/shard_try.py)
    if (per_span >= (p.cfg.win ? 3 : 12) || rr_hint > 0 || p.chunk_tiles > 0) break;
    }
    return p;
}

// compiled on first use, once per (table, geometry); a failed build is remembered and the generic
// kernel takes over (the reason stays readable through vnd_describe_launch)
// cache_only: a launch too small to be worth a 1.5-5 s build takes the per-table kernel only when its code object is
// already there - in this table's map or in the disk cache (looked up once) - and the generic kernel otherwise
static SpecModule *spec_module(vnd_ctx *ctx, const vnd_taps *t_, const SpecConfig &cfg, bool cache_only = false)
{
    vnd_taps *t = const_cast<vnd_taps *>(t_);
    std::lock_guard<std::mutex> g(t->spec_mutex);
    auto it = t->spec_modules.find(cfg);
    if (it == t->spec_modules.end()) {
        std::unique_ptr<SpecModule> m(new SpecModule);
        spec_compile(t->spec_table, cfg, ctx->device, ctx->lds_limit, m.get(), cache_only);
        it = t->spec_modules.emplace(cfg, std::move(m)).first;
    } else if (it->second->pending && !cache_only) {
        spec_compile(t->spec_table, cfg, ctx->device, ctx->lds_limit, it->second.get(), false);
    }
    return it->second->pending ? nullptr : it->second.get();
}

static vnd_status launch_spec(vnd_ctx *ctx, const vnd_taps *t, const SpecPlan &p, const float *x, float *y, int64_t n,
                              hipStream_t stream, bool *launched, const EpiFuse *epi = nullptr, bool *built = nullptr)
{
    *launched = false;
    SpecModule *m = spec_module(ctx, t, p.cfg, !p.eager);
    if (built) *built = !(m && m->failed);                   // false: a build was tried and failed (not: none was tried)
    if (!m || m->failed) return VND_OK;                      // generic kernel instead
    SpecArgs a{};
    a.x = x; a.y = y; a.n = n;
    a.tiles_total = p.tiles_total; a.tiles_per_span = p.tiles_per_span; a.spans = p.spans; a.nblocks = p.nblocks;
    a.units = p.units;
    a.chunk_tiles = p.chunk_tiles; a.chunk_len0 = p.chunk_len0; a.chunks_per_stream = p.chunks_per_stream; a.cus_per_xcd = p.cus_per_xcd;
    a.cus_per_xcd = std::max(1, (ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256) / 8);
    // pacing: one full round of workgroups, two per CU (their co-residency lasts the whole launch), plain stereo forms
    if (p.cfg.win && !p.cfg.win_q && p.chunk_tiles == 0 && p.cfg.win_per_cu == 2 && spec_env("VND_WIN_PACE", 1) != 0 &&
        p.nblocks > (uint32_t)(8 * a.cus_per_xcd) && p.nblocks <= (uint32_t)(2 * 8 * a.cus_per_xcd) && p.units >= p.nblocks &&
        // (long launches only: with a few tiles per workgroup the bias it corrects has no time to build up, and handing the later
        //  workgroup the priority costs - cfg4's N = 4 shard, 3 tiles each: 42.8 -> 48.8 us; cfg3's 17 tiles: +4.7 %)
        (int64_t)p.units * p.tiles_per_span >= (int64_t)p.nblocks * spec_env("VND_WIN_PACE_MIN_TILES", 16)) {
        std::lock_guard<std::mutex> g(ctx->pace_mutex);
        if (!ctx->pace && hipMalloc((void **)&ctx->pace, 2048 * 2 * sizeof(unsigned)) == hipSuccess) {
            if (hipMemset(ctx->pace, 0, 2048 * 2 * sizeof(unsigned)) != hipSuccess) { (void)hipFree(ctx->pace); ctx->pace = nullptr; }
        }
        (void)hipGetLastError();
        a.pace = ctx->pace;
    }
    a.stagger_ticks = p.stagger_ticks; a.chunk_prio = 1;
    if (epi != nullptr && p.cfg.epi) {
        a.epi_ms_encode = epi->ms_encode; a.epi_use_width = epi->use_width; a.epi_w_mid = epi->w_mid; a.epi_w_side = epi->w_side;
        // a wave of the plain 32-frame form owns one 2048-frame block of the sums (kParFrames)
        if (epi->blk_sum != nullptr && p.cfg.win == 32 && !p.cfg.win_s && !p.cfg.win_q && p.cfg.tile() % kParFrames == 0) {
            a.epi_blk_sum = epi->blk_sum; a.epi_nblocks = epi->nblocks; a.epi_rows_major = epi->rows_major;
        }
    }
    void *params[] = {&a};
    hipError_t e = hipModuleLaunchKernel(m->fn, p.nblocks, 1, 1, p.cfg.nt, 1, 1, (unsigned)p.cfg.lds_bytes(), stream, params,
                                         nullptr);
    if (e != hipSuccess) {
        std::lock_guard<std::mutex> g(const_cast<vnd_taps *>(t)->spec_mutex);
        m->failed = true;
        m->log = std::string("launch failed: ") + hipGetErrorString(e);
        (void)hipGetLastError();
        return VND_OK;
    }
    *launched = true;
    if (epi != nullptr && epi->path != nullptr) *epi->path = a.epi_blk_sum != nullptr ? 1 : 2;
    return VND_OK;
}


static vnd_status launch(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch,
                         int64_t n, int32_t C, int32_t mode, hipStream_t stream, const EpiFuse *epi = nullptr,
                         int32_t Cx = 0)
{
    if (batch == 0 || n == 0) return VND_OK;
    if (Cx == 0) Cx = C;
    for (int attempt = 0; attempt < 8; ++attempt) {
        const SpecPlan sp = make_spec_plan(ctx, t, x, y, batch, n, C, Cx, mode, epi);
        if (!sp.use) break;
        bool launched = false, built = true;
        vnd_status st = launch_spec(ctx, t, sp, x, y, n, stream, &launched, epi, &built);
        if (st != VND_OK || launched) return st;
        if (!sp.cfg.win || built) break;                          // (a failed window build: plan again, that geometry is skipped now)
    }
    const Plan p = make_plan(ctx, t, batch, n, C, mode, Cx);
    KArgs a{};
    a.x = x; a.y = y; a.taps = t->d_taps; a.taps_fast = t->d_taps_fast; a.taps_ord = t->d_taps_ord; a.fast_off = t->d_fast_off; a.fast_even = t->d_fast_even; a.tap_off = t->d_tap_off;
    a.seg_off = t->has_seg ? t->d_seg_off : nullptr;
    a.seg_end = t->d_seg_end; a.seg_gain = t->d_seg_gain;
    a.chan_flags = t->has_flags ? t->d_flags : nullptr;
    a.n = n; a.C = C; a.Cx = Cx; a.apply_gain = t->apply_gain;
    // an output beyond what the L2 + Infinity Cache could hand to a consumer is streamed past them
    // (only where a workgroup writes whole frames: pieces of a frame written past the caches by different
    // workgroups reach HBM as separate partial writes)
    a.stream_out = (batch * n * C * (int64_t)sizeof(float) >= ((int64_t)64 << 20) && !spec_env("VND_NO_NT", 0)) ? 1 : 0;
    if (!p.direct && p.cg != C && !spec_env("VND_FORCE_NT", 0)) a.stream_out = 0;
    a.nblocks = p.nblocks;
    if (p.direct) {
        a.tiles = (int32_t)batch; a.groups = 1; a.W = 0;
        kern_t k = arithmetic_of(t, mode) == VND_MODE_EXACT ? conv_direct_kernel<0> : conv_direct_kernel<1>;
        if (mode == VND_MODE_FAST) a.taps = t->d_taps;      // direct kernel keeps the table's association
        hipLaunchKernelGGL(k, dim3(p.nblocks), dim3(kDirectThreads), 0, stream, a);
    } else {
        if ((int64_t)batch * p.tiles * p.groups > 0x7fffffffLL)
            return fail(VND_ERR_UNSUPPORTED, "grid too large; split the batch");
        a.tiles = p.tiles; a.groups = p.groups; a.W = p.W;
        kern_t k = !epi ? pick_kernel(p, arithmetic_of(t, mode))
                 : (mode == VND_MODE_FAST ? fast_epi_kernel(p) : ordered_epi_kernel(p, arithmetic_of(t, mode)));
        if (!k) return fail(VND_ERR_UNSUPPORTED, "no kernel for this tile shape");
        if (epi) {
            a.epi_partials = epi->partials; a.epi_ms_encode = epi->ms_encode; a.epi_use_width = epi->use_width;
            a.epi_normalize = epi->normalize; a.epi_w_mid = epi->w_mid; a.epi_w_side = epi->w_side;
            a.sink_partials = epi->sink;
        }
        if (p.lds_bytes > 65536) {           // opt in to > 64 KiB of dynamic LDS, once per (device, kernel)
            // ask for what the launch needs, not for the whole LDS: a kernel's static LDS (reduction
            // scratch of the epilogue instantiations) counts against the same 160 KiB
            std::lock_guard<std::mutex> g(ctx->raised_mutex);
            size_t &have = ctx->raised[(const void *)k];
            if (have < p.lds_bytes) {
                HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes));
                have = p.lds_bytes;
            }
        }
        hipLaunchKernelGGL(k, dim3(p.nblocks), dim3(p.nt), p.lds_bytes, stream, a);
    }
    HIP_TRY(hipGetLastError());
    return VND_OK;
}

static void free_taps_dev(vnd_taps *t)
{
    if (t->d_taps) (void)hipFree(t->d_taps);
    if (t->d_taps_fast) (void)hipFree(t->d_taps_fast);
    if (t->d_taps_ord) (void)hipFree(t->d_taps_ord);
    if (t->d_fast_off) (void)hipFree(t->d_fast_off);
    if (t->d_fast_even) (void)hipFree(t->d_fast_even);
    if (t->d_tap_off) (void)hipFree(t->d_tap_off);
    if (t->d_seg_off) (void)hipFree(t->d_seg_off);
    if (t->d_seg_end) (void)hipFree(t->d_seg_end);
    if (t->d_seg_gain) (void)hipFree(t->d_seg_gain);
    if (t->d_flags) (void)hipFree(t->d_flags);
}

template <typename T>
static hipError_t upload(T **dst, const T *src, size_t count)
{
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc((void **)dst, bytes);
    if (e != hipSuccess) return e;
    if (count) e = hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

// ------------------------------------------------------------------------------
// ABI
// ------------------------------------------------------------------------------
extern "C" {

int vnd_abi_version(void) { return VND_ABI_VERSION; }

const char *vnd_last_error(void) { return g_err.c_str(); }

vnd_status vnd_device_count(int32_t *count)
{
    if (!count) return fail(VND_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(VND_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return VND_OK;
}

vnd_status vnd_ctx_create(int32_t device, vnd_ctx **out)
{
    if (!out) return fail(VND_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(VND_ERR_NO_DEVICE, "no HIP device visible: the velvet-noise kernels need an MI355X (gfx950)");
    if (device < 0 || device >= n) return fail(VND_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    vnd_ctx *c = new (std::nothrow) vnd_ctx;
    if (!c) return fail(VND_ERR_NOMEM, "out of host memory");
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->prop, device) != hipSuccess) {
        delete c;
        return fail(VND_ERR_HIP, "cannot open device %d", device);
    }
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        std::string arch = c->prop.gcnArchName;
        delete c;
        return fail(VND_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, arch.c_str());
    }
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, device) == hipSuccess && optin > 0)
        c->lds_limit = optin;
    else
        c->lds_limit = (int)c->prop.sharedMemPerBlock;
    if (c->lds_limit < 65536) c->lds_limit = 65536;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) {
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(VND_ERR_HIP, "hipStreamCreate failed");
    }
    *out = c;
    return VND_OK;
}

vnd_status vnd_ctx_destroy(vnd_ctx *c)
{
    if (!c) return VND_OK;
    (void)hipSetDevice(c->device);
    if (c->scratch_x) (void)hipFree(c->scratch_x);
    if (c->scratch_y) (void)hipFree(c->scratch_y);
    if (c->work) (void)hipFree(c->work);
    if (c->pace) (void)hipFree(c->pace);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    for (hipEvent_t ev : c->up_events) (void)hipEventDestroy(ev);
    delete c;
    return VND_OK;
}

vnd_status vnd_ctx_info(const vnd_ctx *c, char *name, int32_t len, int32_t *cus, int64_t *hbm, int32_t *lds)
{
    if (!c) return fail(VND_ERR_INVALID, "null context");
    if (name && len > 0) snprintf(name, (size_t)len, "%s (%s)", c->prop.name, c->prop.gcnArchName);
    if (cus) *cus = c->prop.multiProcessorCount;
    if (hbm) *hbm = (int64_t)c->prop.totalGlobalMem;
    if (lds) *lds = c->lds_limit;
    return VND_OK;
}

vnd_status vnd_taps_create(vnd_ctx *ctx, int32_t C, const int32_t *tap_offsets, const int32_t *tap_index,
                           const float *tap_weight, const int32_t *seg_offsets, const int32_t *seg_end,
                           const float *seg_gain, const uint8_t *chan_flags, int32_t apply_gain,
                           vnd_taps **out)
{
    if (!out) return fail(VND_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!ctx) return fail(VND_ERR_INVALID, "null context");
    if (C <= 0 || C > 65535) return fail(VND_ERR_INVALID, "num_channels %d out of range", C);
    if (!tap_offsets) return fail(VND_ERR_INVALID, "null tap_offsets");
    if (tap_offsets[0] != 0) return fail(VND_ERR_INVALID, "tap_offsets[0] must be 0");
    for (int c = 0; c < C; ++c)
        if (tap_offsets[c + 1] < tap_offsets[c]) return fail(VND_ERR_INVALID, "tap_offsets not monotone");
    const int32_t total = tap_offsets[C];
    if (total > 0 && (!tap_index || !tap_weight)) return fail(VND_ERR_INVALID, "null tap arrays");
    int32_t max_index = 0;
    for (int32_t k = 0; k < total; ++k) {
        if (tap_index[k] < 0) return fail(VND_ERR_INVALID, "negative tap index at %d", k);
        if (tap_index[k] > (1 << 30)) return fail(VND_ERR_UNSUPPORTED, "tap index %d at %d is beyond 2^30 frames", tap_index[k], k);
        max_index = std::max(max_index, tap_index[k]);
    }
    const bool has_seg = seg_offsets != nullptr;
    int32_t total_segs = 0;
    if (has_seg) {
        if (!seg_end || !seg_gain) return fail(VND_ERR_INVALID, "segment table incomplete");
        if (seg_offsets[0] != 0) return fail(VND_ERR_INVALID, "seg_offsets[0] must be 0");
        for (int c = 0; c < C; ++c) {
            if (seg_offsets[c + 1] < seg_offsets[c]) return fail(VND_ERR_INVALID, "seg_offsets not monotone");
            int32_t prev = tap_offsets[c];
            for (int32_t s = seg_offsets[c]; s < seg_offsets[c + 1]; ++s) {
                if (seg_end[s] < prev || seg_end[s] > tap_offsets[c + 1])
                    return fail(VND_ERR_INVALID, "segment %d of channel %d out of order", s, c);
                prev = seg_end[s];
            }
            // taps after the last segment end would be dropped silently: refuse
            if (prev != tap_offsets[c + 1] && tap_offsets[c + 1] != tap_offsets[c])
                return fail(VND_ERR_INVALID, "segments of channel %d do not cover its taps", c);
        }
        total_segs = seg_offsets[C];
    }
    vnd_taps *t = new (std::nothrow) vnd_taps;
    if (!t) return fail(VND_ERR_NOMEM, "out of host memory");
    t->ctx = ctx; t->C = C; t->total = total; t->max_index = max_index;
    t->unit_weights = true;
    for (int32_t k = 0; k < total; ++k) {
        t->unit_weights &= (tap_weight[k] == 1.0f || tap_weight[k] == -1.0f);
        t->nonfinite |= !std::isfinite(tap_weight[k]);
    }
    // the LDS kernels address taps by 32-bit byte offsets; a halo this long never fits LDS anyway
    t->lds_images = max_index < (1 << 24);
    t->apply_gain = apply_gain ? 1 : 0; t->has_seg = has_seg; t->total_segs = total_segs;
    t->tap_off.assign(tap_offsets, tap_offsets + C + 1);
    if (total) { t->idx.assign(tap_index, tap_index + total); t->w.assign(tap_weight, tap_weight + total); }
    if (has_seg) {
        t->seg_off.assign(seg_offsets, seg_offsets + C + 1);
        t->seg_end.assign(seg_end, seg_end + total_segs);
        t->seg_gain.assign(seg_gain, seg_gain + total_segs);
    }
    if (chan_flags) { t->flags.assign(chan_flags, chan_flags + C); t->has_flags = true; }

    std::vector<Tap> packed((size_t)total);
    for (int32_t k = 0; k < total; ++k) { packed[k].idx = tap_index[k]; packed[k].w = tap_weight[k]; }
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = upload(&t->d_taps, packed.data(), (size_t)total);
    // fast-mode image: per channel the even-offset taps, then the odd-offset ones;
    // idx <- LDS byte offset (i & ~1) * 4, w <- weight * segment gain; 16 zero records of
    // padding so that a 16-record scalar fetch never leaves the array
    std::vector<FastTap> fast;
    std::vector<int32_t> fast_off(C + 1, 0), fast_even(C, 0);
    {
        std::vector<float> eff(tap_weight, tap_weight + total);
        if (has_seg && t->apply_gain)
            for (int c = 0; c < C; ++c) {
                int32_t k = tap_offsets[c];
                for (int32_t sgi = seg_offsets[c]; sgi < seg_offsets[c + 1]; ++sgi)
                    for (; k < seg_end[sgi]; ++k) eff[k] = tap_weight[k] * seg_gain[sgi];
            }
        for (int c = 0; c < C; ++c) {
            for (int parity = 0; parity < 2; ++parity) {
                for (int32_t k = tap_offsets[c]; k < tap_offsets[c + 1]; ++k)
                    if ((tap_index[k] & 1) == parity)
                        fast.push_back(FastTap{eff[k], t->lds_images ? (tap_index[k] & ~1) * 4 : 0});
                if (parity == 0) fast_even[c] = (int32_t)fast.size() - fast_off[c];
            }
            fast_off[c + 1] = (int32_t)fast.size();
        }
        fast.resize(fast.size() + 16, FastTap{0.0f, 0});
        // the specialised fast kernel's view of the table: channel pairs, every channel filtered
        t->spec_table.C = C;
        t->spec_table.tap_off = t->tap_off;
        t->spec_table.idx = t->idx;
        t->spec_table.w = eff;
        t->spec_table.max_index = max_index;
        bool copy_through = false;                             // a channel that is copied, not filtered: generic kernels only
        for (int c = 0; c < C && t->has_flags; ++c) copy_through |= (t->flags[c] & 1) != 0;
        t->spec_ok = t->lds_images && !t->nonfinite && !copy_through && C % 2 == 0 && C <= 64 && total > 0;
        t->spec_table.w_raw.assign(tap_weight, tap_weight + total);
        t->spec_table.has_seg = has_seg;
        t->spec_table.apply_gain = t->apply_gain != 0;
        t->spec_exact_ok = t->spec_ok;
        if (has_seg) {
            t->spec_table.seg_off = t->seg_off; t->spec_table.seg_end = t->seg_end; t->spec_table.seg_gain = t->seg_gain;
            for (int c = 0; c < C; ++c) {
                int32_t prev = tap_offsets[c];
                for (int32_t sg = seg_offsets[c]; sg < seg_offsets[c + 1]; ++sg) {
                    if (seg_end[sg] == prev) t->spec_exact_ok = false;      // an empty segment still adds +0: generic kernel
                    prev = seg_end[sg];
                }
            }
        }
        // VND_MODE_EXACT in the window form: ahead of the pair-read exact kernel on every stereo table measured once its
        // odd-offset taps became single adds (cfg2 function path 4.48 against 4.20 TB/s, class path 4.77 against 4.64; cfg3
        // 1.93 against 1.40 and 2.03 against 1.74: tools/win_exact_try.py, profiles/r03_exact_window.txt)
        t->win_exact_pays = t->spec_exact_ok && C % 2 == 0;
    }
    if (e == hipSuccess) e = upload(&t->d_taps_fast, fast.data(), fast.size());
    {   // ordered image: table order, weight first (SGPR pair layout), byte offsets, padded
        std::vector<FastTap> ord((size_t)total + 16, FastTap{0.0f, 0});
        for (int32_t k = 0; k < total; ++k) ord[k] = FastTap{tap_weight[k], t->lds_images ? tap_index[k] * 4 : 0};
        if (e == hipSuccess) e = upload(&t->d_taps_ord, ord.data(), ord.size());
    }
    if (e == hipSuccess) e = upload(&t->d_fast_off, fast_off.data(), fast_off.size());
    if (e == hipSuccess) e = upload(&t->d_fast_even, fast_even.data(), fast_even.size());
    if (e == hipSuccess) e = upload(&t->d_tap_off, t->tap_off.data(), t->tap_off.size());
    if (e == hipSuccess && has_seg) e = upload(&t->d_seg_off, t->seg_off.data(), t->seg_off.size());
    if (e == hipSuccess && has_seg) e = upload(&t->d_seg_end, t->seg_end.data(), t->seg_end.size());
    if (e == hipSuccess && has_seg) e = upload(&t->d_seg_gain, t->seg_gain.data(), t->seg_gain.size());
    if (e == hipSuccess && t->has_flags) e = upload(&t->d_flags, t->flags.data(), t->flags.size());
    if (e != hipSuccess) {
        free_taps_dev(t);
        delete t;
        return fail(VND_ERR_HIP, "uploading tap table: %s", hipGetErrorString(e));
    }
    *out = t;
    return VND_OK;
}

vnd_status vnd_taps_destroy(vnd_taps *t)
{
    if (!t) return VND_OK;
    (void)hipSetDevice(t->ctx->device);
    free_taps_dev(t);
    for (auto &kv : t->spec_modules)
        if (kv.second && kv.second->module) (void)hipModuleUnload(kv.second->module);
    delete t;
    return VND_OK;
}

vnd_status vnd_taps_info(const vnd_taps *t, int32_t *C, int32_t *total, int32_t *max_index)
{
    if (!t) return fail(VND_ERR_INVALID, "null tap table");
    if (C) *C = t->C;
    if (total) *total = t->total;
    if (max_index) *max_index = t->max_index;
    return VND_OK;
}

// image: int32 header[8] = {magic, version, C, total, total_segs, has_seg, has_flags, apply_gain}
// then tap_off[C+1], idx[total], w[total], (seg_off[C+1], seg_end[S], seg_gain[S]), (flags[C] padded to 4)
static const int32_t kMagic = 0x564e4454;  // "VNDT"

vnd_status vnd_taps_serialize(const vnd_taps *t, void *buf, int64_t capacity, int64_t *bytes)
{
    if (!t || !bytes) return fail(VND_ERR_INVALID, "null argument");
    const int64_t flag_words = t->has_flags ? (t->C + 3) / 4 : 0;
    const int64_t words = 8 + (t->C + 1) + 2 * (int64_t)t->total +
                          (t->has_seg ? (t->C + 1) + 2 * (int64_t)t->total_segs : 0) + flag_words;
    *bytes = words * 4;
    if (!buf) return VND_OK;                     // size query
    if (capacity < *bytes) return fail(VND_ERR_INVALID, "buffer too small: need %lld bytes", (long long)*bytes);
    int32_t *p = (int32_t *)buf;
    const int32_t hdr[8] = {kMagic, VND_TAPS_IMAGE_VERSION, t->C, t->total, t->total_segs,
                            t->has_seg, t->has_flags, t->apply_gain};
    memcpy(p, hdr, sizeof hdr); p += 8;
    memcpy(p, t->tap_off.data(), (t->C + 1) * 4); p += t->C + 1;
    if (t->total) { memcpy(p, t->idx.data(), t->total * 4); p += t->total;
                    memcpy(p, t->w.data(), t->total * 4); p += t->total; }
    if (t->has_seg) {
        memcpy(p, t->seg_off.data(), (t->C + 1) * 4); p += t->C + 1;
        if (t->total_segs) { memcpy(p, t->seg_end.data(), t->total_segs * 4); p += t->total_segs;
                             memcpy(p, t->seg_gain.data(), t->total_segs * 4); p += t->total_segs; }
    }
    if (t->has_flags) { memset(p, 0, flag_words * 4); memcpy(p, t->flags.data(), t->C); }
    return VND_OK;
}

vnd_status vnd_taps_deserialize(vnd_ctx *ctx, const void *buf, int64_t bytes, vnd_taps **out)
{
    if (!out) return fail(VND_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!buf || bytes < 32) return fail(VND_ERR_INVALID, "tap image too short");
    const int32_t *p = (const int32_t *)buf;
    if (p[0] != kMagic || p[1] != VND_TAPS_IMAGE_VERSION) return fail(VND_ERR_INVALID, "not a tap image of this format version");
    const int32_t C = p[2], total = p[3], segs = p[4], has_seg = p[5], has_flags = p[6], gain = p[7];
    if (C <= 0 || total < 0 || segs < 0) return fail(VND_ERR_INVALID, "corrupt tap image header");
    const int64_t flag_words = has_flags ? (C + 3) / 4 : 0;
    const int64_t words = 8 + (C + 1) + 2 * (int64_t)total + (has_seg ? (C + 1) + 2 * (int64_t)segs : 0) + flag_words;
    if (bytes < words * 4) return fail(VND_ERR_INVALID, "tap image truncated");
    const int32_t *tap_off = p + 8;
    const int32_t *idx = tap_off + C + 1;
    const float *w = (const float *)(idx + total);
    const int32_t *q = (const int32_t *)(w + total);
    const int32_t *seg_off = nullptr, *seg_end = nullptr;
    const float *seg_gain = nullptr;
    if (has_seg) { seg_off = q; seg_end = seg_off + C + 1; seg_gain = (const float *)(seg_end + segs); q = (const int32_t *)(seg_gain + segs); }
    const uint8_t *flags = has_flags ? (const uint8_t *)q : nullptr;
    if (tap_off[C] != total || (has_seg && seg_off[C] != segs)) return fail(VND_ERR_INVALID, "corrupt tap image");
    return vnd_taps_create(ctx, C, tap_off, idx, w, seg_off, seg_end, seg_gain, flags, gain, out);
}

static vnd_status ensure_scratch(vnd_ctx *ctx, size_t elems)
{
    if (elems <= ctx->scratch_elems) return VND_OK;
    if (ctx->scratch_x) (void)hipFree(ctx->scratch_x);
    if (ctx->scratch_y) (void)hipFree(ctx->scratch_y);
    ctx->scratch_x = ctx->scratch_y = nullptr;
    ctx->scratch_elems = 0;
    HIP_TRY(hipMalloc((void **)&ctx->scratch_x, elems * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&ctx->scratch_y, elems * sizeof(float)));
    ctx->scratch_elems = elems;
    return VND_OK;
}

// groups of streams the host entry points pipeline a batch in: one below 16 MB of traffic, then about
// 32 MB each, at most 16
static int host_chunks(int64_t batch, size_t bytes)
{
    if (batch < 2 || bytes < ((size_t)16 << 20)) return 1;
    const size_t want = (bytes + ((size_t)32 << 20) - 1) / ((size_t)32 << 20);
    return (int)std::min<int64_t>(std::min<int64_t>(batch, 16), (int64_t)std::max<size_t>(want, 2));
}

static vnd_status ensure_work(vnd_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->work_bytes) return VND_OK;
    if (ctx->work) (void)hipFree(ctx->work);
    ctx->work = nullptr;
    ctx->work_bytes = 0;
    HIP_TRY(hipMalloc((void **)&ctx->work, bytes));
    ctx->work_bytes = bytes;
    return VND_OK;
}

// The *_dev entry points launch on the context's device whatever the caller's current device is,
// and leave the caller's current device as they found it.
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int dev)
    {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess) prev = cur;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// ---- sharding helpers for hosts that do not go through torch.distributed -------
vnd_status vnd_shard_range(int64_t total, int32_t world_size, int32_t rank, int64_t *first, int64_t *count)
{
    if (!first || !count) return fail(VND_ERR_INVALID, "null out pointer");
    if (total < 0 || world_size <= 0 || rank < 0 || rank >= world_size)
        return fail(VND_ERR_INVALID, "bad shard query: %lld streams, rank %d of %d", (long long)total, rank, world_size);
    const int64_t base = total / world_size, extra = total % world_size;
    *count = base + (rank < extra ? 1 : 0);
    *first = rank * base + std::min<int64_t>(rank, extra);
    return VND_OK;
}

// RCCL is loaded on first use: a host that never shards needs no librccl
namespace {
struct RcclApi {
    int (*broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*error_string)(int) = nullptr;
    bool tried = false;
};
RcclApi *rccl_api()
{
    static RcclApi api;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (!api.tried) {
        api.tried = true;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            api.broadcast = (decltype(api.broadcast))dlsym(h, "ncclBroadcast");
            api.error_string = (decltype(api.error_string))dlsym(h, "ncclGetErrorString");
        }
    }
    return &api;
}
constexpr int kNcclUint8 = 1;         // rccl.h: ncclDataType_t
}  // namespace

vnd_status vnd_taps_broadcast_rccl(vnd_ctx *ctx, vnd_taps **taps, int32_t root, int32_t rank, void *rccl_comm,
                                   void *stream_)
{
    if (!ctx || !taps || !rccl_comm) return fail(VND_ERR_INVALID, "null context, table slot or communicator");
    if (rank == root && !*taps) return fail(VND_ERR_INVALID, "the root rank has no table to send");
    RcclApi &api = *rccl_api();
    if (!api.broadcast) return fail(VND_ERR_UNSUPPORTED, "librccl.so could not be loaded");
    DeviceScope on(ctx->device);
    hipStream_t stream = (hipStream_t)stream_;
    auto rccl_try = [&](int rc, const char *what) -> vnd_status {
        if (rc == 0) return VND_OK;
        return fail(VND_ERR_HIP, "%s: %s", what, api.error_string ? api.error_string(rc) : "RCCL error");
    };
    // two broadcasts: the image's length, then the image (32 B header + 8 B per tap)
    int64_t bytes = 0;
    std::vector<char> image;
    if (rank == root) {
        vnd_status st = vnd_taps_serialize(*taps, nullptr, 0, &bytes);
        if (st != VND_OK) return st;
        image.resize((size_t)bytes);
        st = vnd_taps_serialize(*taps, image.data(), bytes, &bytes);
        if (st != VND_OK) return st;
    }
    int64_t *d_len = nullptr;
    HIP_TRY(hipMalloc((void **)&d_len, sizeof(int64_t)));
    char *d_body = nullptr;
    vnd_status st = VND_OK;
    do {
        if (rank == root && hipMemcpyAsync(d_len, &bytes, sizeof bytes, hipMemcpyHostToDevice, stream) != hipSuccess) { st = fail(VND_ERR_HIP, "upload of the image length failed"); break; }
        if ((st = rccl_try(api.broadcast(d_len, d_len, sizeof(int64_t), kNcclUint8, root, rccl_comm, stream), "ncclBroadcast(length)")) != VND_OK) break;
        if (hipMemcpyAsync(&bytes, d_len, sizeof bytes, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { st = fail(VND_ERR_HIP, "download of the image length failed"); break; }
        if (bytes < 32 || bytes > ((int64_t)1 << 31)) { st = fail(VND_ERR_INVALID, "implausible tap image length %lld", (long long)bytes); break; }
        if (hipMalloc((void **)&d_body, (size_t)bytes) != hipSuccess) { st = fail(VND_ERR_NOMEM, "no device memory for the tap image"); break; }
        if (rank == root && hipMemcpyAsync(d_body, image.data(), (size_t)bytes, hipMemcpyHostToDevice, stream) != hipSuccess) { st = fail(VND_ERR_HIP, "upload of the tap image failed"); break; }
        if ((st = rccl_try(api.broadcast(d_body, d_body, (size_t)bytes, kNcclUint8, root, rccl_comm, stream), "ncclBroadcast(image)")) != VND_OK) break;
        if (rank != root) {
            image.resize((size_t)bytes);
            if (hipMemcpyAsync(image.data(), d_body, (size_t)bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) { st = fail(VND_ERR_HIP, "download of the tap image failed"); break; }
        }
        if (hipStreamSynchronize(stream) != hipSuccess) { st = fail(VND_ERR_HIP, "stream synchronisation failed"); break; }
        if (rank != root) {
            // what arrived must be a tap image of exactly the announced length before anything is built from it
            // (a communicator whose ranks disagree on the root, or a torn transfer, shows up here, loudly)
            const int32_t *hd = (const int32_t *)image.data();
            if (hd[0] != kMagic || hd[1] != VND_TAPS_IMAGE_VERSION) { st = fail(VND_ERR_INVALID, "rank %d received %lld bytes that are not a tap image (magic %08x, version %d)", rank, (long long)bytes, (unsigned)hd[0], hd[1]); break; }
            const int64_t words = 8 + ((int64_t)hd[2] + 1) + 2 * (int64_t)hd[3] + (hd[5] ? ((int64_t)hd[2] + 1) + 2 * (int64_t)hd[4] : 0) + (hd[6] ? ((int64_t)hd[2] + 3) / 4 : 0);
            if (hd[2] <= 0 || hd[3] < 0 || hd[4] < 0 || words * 4 != bytes) { st = fail(VND_ERR_INVALID, "rank %d: the tap image's header (%d channels, %d taps, %d segments) does not match its %lld bytes", rank, hd[2], hd[3], hd[4], (long long)bytes); break; }
            st = vnd_taps_deserialize(ctx, image.data(), bytes, taps);
        }
    } while (false);
    if (d_body) (void)hipFree(d_body);
    (void)hipFree(d_len);
    return st;
}


static bool overlaps(const float *x, int64_t x_elems, const float *y, int64_t y_elems)
{
    return (x < y + y_elems) && (y < x + x_elems);
}

// x: [batch][n][Cx], y: [batch][n][C]
static vnd_status convolve_dev(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch,
                               int64_t n, int32_t Cx, int32_t C, int32_t mode, void *stream)
{
    vnd_status st = check_shape(ctx, t, batch, n, C, mode, Cx);
    if (st != VND_OK) return st;
    if (batch == 0 || n == 0) return VND_OK;
    if (!x || !y) return fail(VND_ERR_INVALID, "null signal pointer");
    if (overlaps(x, batch * n * Cx, y, batch * n * C)) return fail(VND_ERR_INVALID, "x and y overlap");
    DeviceScope on(ctx->device);
    return launch(ctx, t, x, y, batch, n, C, mode, (hipStream_t)stream, nullptr, Cx);
}

// Few long streams (the reference's own use is one file at a time, tests/test_example.py:19-49) are cut in TIME:
// piece k = frames [f_k, f_k+1) of a stream.  Output frame n reads input frames n .. n + max_index
// (decorrelation.py:656-658), so the launch of piece k runs over [f_k, f_k+1 + max_index) - the tail it computes
// from an input that ends too early is overwritten by the launch of piece k + 1, on the same HIP stream - and needs
// the upload of the piece that holds frame f_k+1 + max_index.  Uploads run on one HIP stream, kernels and downloads
// on the other: the (CPU-staged) upload of piece k + 2 beside the kernel of piece k + 1 and the download of piece k.
// Every kernel of this library computes an output frame the same way wherever it lies in a launch, so the result
// is the unchunked call's, bit for bit in VND_MODE_EXACT.
static int host_time_pieces(int64_t batch, int64_t n, size_t bytes, bool pinned)
{
    // Measured (tools/host_pieces_try.py, profiles/r03_host_pieces.txt): every extra copy call costs ~50 us of fixed time on
    // this platform, so one 10 s signal (3.84 MB each way, 0.20 ms in one piece) only loses - 0.25 ms in 2 pieces, 0.36 in
    // 6 - and a pageable 60 s one too (its upload is staged by the CPU, call by call); a PAGE-LOCKED 60 s stream gains 5 %
    // with 4 pieces (0.85 vs 0.90 ms).  So: page-locked input of 16 MB and more per stream; VND_HOST_TIME_PIECES forces.
    const char *e_off = getenv("VND_HOST_TIME_CHUNKS"), *e_forced = getenv("VND_HOST_TIME_PIECES");     // (a host call is ms-scale)
    const bool off = e_off && e_off[0] == '0';
    const int forced = e_forced ? atoi(e_forced) : 0;
    if (off || batch > 4 || n < 8 * 4096) return 1;
    if (forced > 0) return (int)std::min<int64_t>(forced, n / 4096);
    const size_t per_stream = bytes / (size_t)batch / 2;
    if (per_stream < ((size_t)16 << 20) || !pinned) return 1;
    return (int)std::min<int64_t>(4, n / 4096);
}

static vnd_status host_time_pipeline(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch, int64_t n,
                                     int32_t Cx, int32_t C, int pieces,
                                     const std::function<vnd_status(const float *, float *, int64_t, hipStream_t)> &launch_piece)
{
    const int64_t total = batch * pieces;
    while ((int64_t)ctx->up_events.size() < total) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->up_events.push_back(ev);
    }
    // piece boundaries on 4096-frame marks: every piece starts 16-byte aligned whatever the channel count
    auto first_frame = [&](int k) { return k >= pieces ? n : ((n * k / pieces) / 4096) * 4096; };
    const int64_t halo = t->max_index;
    vnd_status st = VND_OK;
    hipError_t e = hipSuccess;
    int64_t uploaded = 0;                                        // flat pieces handed to the upload stream so far
    auto upload_through = [&](int64_t flat) {
        for (; uploaded <= flat && e == hipSuccess; ++uploaded) {
            const int64_t b = uploaded / pieces;
            const int k = (int)(uploaded % pieces);
            const int64_t f0 = first_frame(k), f1 = first_frame(k + 1);
            const size_t xo = ((size_t)b * n + f0) * Cx;
            if (f1 > f0) e = hipMemcpyAsync(ctx->scratch_x + xo, x + xo, (size_t)(f1 - f0) * Cx * sizeof(float), hipMemcpyHostToDevice, ctx->stream2);
            if (e == hipSuccess) e = hipEventRecord(ctx->up_events[uploaded], ctx->stream2);
        }
    };
    for (int64_t flat = 0; flat < total && st == VND_OK && e == hipSuccess; ++flat) {
        const int64_t b = flat / pieces;
        const int k = (int)(flat % pieces);
        const int64_t f0 = first_frame(k), f1 = first_frame(k + 1);
        if (f1 == f0) continue;
        const int64_t reach = std::min(n, f1 + halo);            // the launch reads input frames [f0, reach)
        int last = k;
        while (last + 1 < pieces && first_frame(last + 1) < reach) ++last;
        upload_through(b * pieces + last);
        if (e != hipSuccess) break;
        e = hipStreamWaitEvent(ctx->stream, ctx->up_events[b * pieces + last], 0);
        if (e != hipSuccess) break;
        const size_t xo = ((size_t)b * n + f0) * Cx, yo = ((size_t)b * n + f0) * C;
        st = launch_piece(ctx->scratch_x + xo, ctx->scratch_y + yo, reach - f0, ctx->stream);
        if (st != VND_OK) break;
        e = hipMemcpyAsync(y + yo, ctx->scratch_y + yo, (size_t)(f1 - f0) * C * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    }
    // whatever happened, nothing of this call is in flight when it returns: the caller's arrays and the
    // context's staging buffers are free again
    const hipError_t s1 = hipStreamSynchronize(ctx->stream), s2 = hipStreamSynchronize(ctx->stream2);
    if (st != VND_OK) return st;
    if (e == hipSuccess) e = s1 != hipSuccess ? s1 : s2;
    if (e != hipSuccess) return fail(VND_ERR_HIP, "time-chunked host pipeline failed: %s", hipGetErrorString(e));
    return VND_OK;
}

// A page-locked host buffer (hipHostMalloc: vnd_host_alloc, torch's pin_memory; hipHostRegister) is mapped into the
// device's address space: *dev = the address a kernel reaches it at, if all of [p, p + bytes) is such memory.
static bool host_mapped(const void *p, size_t bytes, void **dev)
{
    if (!p || bytes == 0) return false;
    hipPointerAttribute_t first{}, last{};
    const bool ok = hipPointerGetAttributes(&first, p) == hipSuccess &&
                    hipPointerGetAttributes(&last, (const char *)p + bytes - 1) == hipSuccess;
    (void)hipGetLastError();                                      // (an ordinary pageable pointer reports an error: not ours)
    if (!ok || first.type != hipMemoryTypeHost || last.type != hipMemoryTypeHost || !first.devicePointer || !last.devicePointer)
        return false;
    if ((const char *)last.devicePointer - (const char *)first.devicePointer != (ptrdiff_t)(bytes - 1)) return false;
    *dev = first.devicePointer;
    return true;
}

vnd_status vnd_host_buffers_mapped(const void *x, int64_t x_bytes, const void *y, int64_t y_bytes, int32_t *mapped)
{
    if (!mapped || x_bytes < 0 || y_bytes < 0) return fail(VND_ERR_INVALID, "bad arguments");
    void *xd = nullptr, *yd = nullptr;
    *mapped = host_mapped(x, (size_t)x_bytes, &xd) && host_mapped(y, (size_t)y_bytes, &yd) ? 1 : 0;
    return VND_OK;
}

static vnd_status convolve_host(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch,
                                int64_t n, int32_t Cx, int32_t C, int32_t mode)
{
    vnd_status st = check_shape(ctx, t, batch, n, C, mode, Cx);
    if (st != VND_OK) return st;
    if (batch == 0 || n == 0) return VND_OK;
    if (!x || !y) return fail(VND_ERR_INVALID, "null signal pointer");
    HostLock lock(ctx->host_mutex);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t in_elems = (size_t)batch * n * Cx, out_elems = (size_t)batch * n * C;
    // Page-locked buffers on BOTH sides: the kernel works on them in place - its loads and stores cross PCIe inside the
    // launch, both directions at once, with no staging copy before or after (one 10 s stereo signal 0.147 against 0.185 ms,
    // 1024 x 1 s 9.95 against 14.1 ms: tools/zero_copy_try.py).  Every frame is read once plus the halo at span seams, and
    // written once: the bytes over PCIe are the staged path's.  VND_HOST_DIRECT=0 keeps the staged path.
    // (measured and dropped, same tool: a mapped input read in place with a staged download per group - 15.1 ms for the
    //  1024 streams; a staged upload with every group written in place - 13.7 ms with page-locked, 9.8-10.1 with pageable
    //  input against the staged pipeline's 8.8: a pageable upload is staged by the CPU, beside the SDMA download.)
    static int direct_slot = INT32_MIN;
    const bool direct = host_env_once("VND_HOST_DIRECT", 1, &direct_slot) != 0;
    void *xd = nullptr, *yd = nullptr;
    const bool apart = !overlaps(x, (int64_t)in_elems, y, (int64_t)out_elems);
    const bool x_mapped = direct && apart && host_mapped(x, in_elems * sizeof(float), &xd);
    const bool y_mapped = direct && apart && host_mapped(y, out_elems * sizeof(float), &yd);
    if (x_mapped && y_mapped) {
        st = launch(ctx, t, (const float *)xd, (float *)yd, batch, n, C, mode, ctx->stream, nullptr, Cx);
        const hipError_t e = hipStreamSynchronize(ctx->stream);
        if (st != VND_OK) return st;
        if (e != hipSuccess) return fail(VND_ERR_HIP, "host call on mapped buffers failed: %s", hipGetErrorString(e));
        return VND_OK;
    }
    st = ensure_scratch(ctx, out_elems);
    if (st != VND_OK) return st;
    // A batch is cut into groups of whole streams that alternate between two HIP streams: the upload of
    // one group runs beside the kernel and the download of the one before (PCIe is full duplex, and a
    // download into pinned memory - vnd_host_alloc - does not hold the host thread).
    const int chunks = host_chunks(batch, (in_elems + out_elems) * sizeof(float));
    if (chunks == 1) {
        hipPointerAttribute_t attr{};
        const bool pinned = hipPointerGetAttributes(&attr, x) == hipSuccess && attr.type == hipMemoryTypeHost;
        (void)hipGetLastError();                                  // (an ordinary pageable pointer reports an error: not ours)
        const int pieces = host_time_pieces(batch, n, (in_elems + out_elems) * sizeof(float), pinned);
        if (pieces > 1)
            return host_time_pipeline(ctx, t, x, y, batch, n, Cx, C, pieces, [&](const float *xp, float *yp, int64_t frames, hipStream_t s) {
                return launch(ctx, t, xp, yp, 1, frames, C, mode, s, nullptr, Cx);
            });
    }
    hipError_t e = hipSuccess;
    for (int c = 0; c < chunks && st == VND_OK && e == hipSuccess; ++c) {
        const int64_t b0 = batch * c / chunks, b1 = batch * (c + 1) / chunks;
        if (b1 == b0) continue;
        hipStream_t s = (c & 1) ? ctx->stream2 : ctx->stream;
        const size_t xo = (size_t)b0 * n * Cx, yo = (size_t)b0 * n * C;
        // One stream or a small batch in ONE group, and the result in mapped memory (the Python layer's page-locked pool):
        // the kernel writes it in place - no download behind the kernel (a pageable 10 s stereo signal 0.166 against 0.188 ms).
        // Larger batches keep the staged download: group k's beside the upload and the kernel of group k + 1.
        const bool in_place = y_mapped && chunks == 1;
        e = hipMemcpyAsync(ctx->scratch_x + xo, x + xo, (size_t)(b1 - b0) * n * Cx * sizeof(float), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) break;
        st = launch(ctx, t, ctx->scratch_x + xo, in_place ? (float *)yd + yo : ctx->scratch_y + yo, b1 - b0, n, C, mode, s, nullptr, Cx);
        if (st != VND_OK) break;
        if (!in_place) e = hipMemcpyAsync(y + yo, ctx->scratch_y + yo, (size_t)(b1 - b0) * n * C * sizeof(float), hipMemcpyDeviceToHost, s);
    }
    // on any failure too: copies and kernels of the earlier groups may still be in flight, and the caller is about
    // to recycle its (pinned) result block, the next call this context's staging buffers
    const hipError_t s1 = hipStreamSynchronize(ctx->stream), s2 = hipStreamSynchronize(ctx->stream2);
    if (st != VND_OK) return st;
    if (e == hipSuccess) e = s1 != hipSuccess ? s1 : s2;
    if (e != hipSuccess) return fail(VND_ERR_HIP, "host pipeline failed: %s", hipGetErrorString(e));
    return VND_OK;
}

vnd_status vnd_convolve_f32_dev(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch,
                                int64_t n, int32_t C, int32_t mode, void *stream)
{
    return convolve_dev(ctx, t, x, y, batch, n, C, C, mode, stream);
}

vnd_status vnd_convolve_f32_host(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch,
                                 int64_t n, int32_t C, int32_t mode)
{
    return convolve_host(ctx, t, x, y, batch, n, C, C, mode);
}

vnd_status vnd_convolve_fanout_f32_dev(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch,
                                       int64_t n, int32_t in_channels, int32_t mode, void *stream)
{
    if (!t) return fail(VND_ERR_INVALID, "null context or tap table");
    if (in_channels <= 0) return fail(VND_ERR_INVALID, "in_channels must be positive");
    return convolve_dev(ctx, t, x, y, batch, n, in_channels, t->C, mode, stream);
}

vnd_status vnd_convolve_fanout_f32_host(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch,
                                        int64_t n, int32_t in_channels, int32_t mode)
{
    if (!t) return fail(VND_ERR_INVALID, "null context or tap table");
    if (in_channels <= 0) return fail(VND_ERR_INVALID, "in_channels must be positive");
    return convolve_host(ctx, t, x, y, batch, n, in_channels, t->C, mode);
}

vnd_status vnd_time_convolve_f32_dev(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch,
                                     int64_t n, int32_t C, int32_t mode, int32_t n_buffers, int64_t stride,
                                     int32_t iters, void *stream_, float *avg_ms)
{
    vnd_status st = check_shape(ctx, t, batch, n, C, mode);
    if (st != VND_OK) return st;
    if (!avg_ms || iters <= 0 || n_buffers <= 0) return fail(VND_ERR_INVALID, "bad timing arguments");
    DeviceScope on(ctx->device);
    hipStream_t stream = (hipStream_t)stream_;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t he = hipEventCreate(&e0);
    if (he == hipSuccess) he = hipEventCreate(&e1);
    if (he == hipSuccess) he = hipEventRecord(e0, stream);
    for (int i = 0; he == hipSuccess && st == VND_OK && i < iters; ++i) {
        const int64_t off = (int64_t)(i % n_buffers) * stride;
        st = launch(ctx, t, x + off, y + off, batch, n, C, mode, stream);
    }
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventRecord(e1, stream);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st != VND_OK) return st;
    if (he != hipSuccess) return fail(VND_ERR_HIP, "timing: %s", hipGetErrorString(he));
    *avg_ms = ms / iters;
    return VND_OK;
}

// The streaming ceiling of the box, for bench.py: a plain copy with the per-table kernels' access shape (16 bytes per
// lane, non-temporal loads and stores) - the best of the shapes tools/micro/copy_ceiling.hip tries (5.9 TB/s on 7.9 GB
// each way, where hipMemcpyAsync reaches 5.1).  Not on the data path.
typedef float copy_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(const copy_v4f *x, copy_v4f *y, long long quads)
{
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long base = (long long)blockIdx.x * 256 * 4 + threadIdx.x; base < quads; base += stride) {
        copy_v4f a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (base + k * 256 < quads) a[k] = __builtin_nontemporal_load(x + base + k * 256);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (base + k * 256 < quads) __builtin_nontemporal_store(a[k], y + base + k * 256);
    }
}

vnd_status vnd_time_copy_f32_dev(vnd_ctx *ctx, const float *x, float *y, int64_t elems, int32_t iters, void *stream_, float *avg_ms)
{
    if (!ctx || !x || !y || !avg_ms || iters <= 0 || elems <= 0 || (elems & 3)) return fail(VND_ERR_INVALID, "bad copy timing arguments");
    if (((uintptr_t)x | (uintptr_t)y) & 15) return fail(VND_ERR_INVALID, "the copy wants 16-byte aligned buffers");
    DeviceScope on(ctx->device);
    hipStream_t stream = (hipStream_t)stream_;
    const long long quads = elems / 4;
    const unsigned grid = (unsigned)std::min<long long>(65536, (quads + 1023) / 1024);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t he = hipEventCreate(&e0);
    if (he == hipSuccess) he = hipEventCreate(&e1);
    if (he == hipSuccess) he = hipEventRecord(e0, stream);
    for (int i = 0; he == hipSuccess && i < iters; ++i)
        hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(256), 0, stream, (const copy_v4f *)x, (copy_v4f *)y, quads);
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventRecord(e1, stream);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    if (he == hipSuccess) he = hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (he != hipSuccess) return fail(VND_ERR_HIP, "copy timing: %s", hipGetErrorString(he));
    *avg_ms = ms / iters;
    return VND_OK;
}

vnd_status vnd_host_alloc(int64_t bytes, void **ptr)
{
    if (!ptr || bytes <= 0) return fail(VND_ERR_INVALID, "bad host allocation request");
    *ptr = nullptr;
    hipError_t e = hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(VND_ERR_NOMEM, "hipHostMalloc(%lld): %s", (long long)bytes, hipGetErrorString(e)); }
    return VND_OK;
}

vnd_status vnd_host_free(void *ptr)
{
    if (ptr && hipHostFree(ptr) != hipSuccess) { (void)hipGetLastError(); return fail(VND_ERR_HIP, "hipHostFree failed"); }
    return VND_OK;
}

vnd_status vnd_spec_kernel_source(int32_t C, const int32_t *tap_offsets, const int32_t *tap_index,
                                  const float *tap_weight, int32_t mode, char *text, int64_t capacity, int64_t *bytes)
{
    if (mode != VND_MODE_FAST && mode != VND_MODE_EXACT)
        return fail(VND_ERR_INVALID, "the specialised kernel exists for VND_MODE_FAST and VND_MODE_EXACT");
    if (!bytes) return fail(VND_ERR_INVALID, "null bytes pointer");
    if (C <= 0 || C % 2 != 0 || C > 64 || !tap_offsets || tap_offsets[0] != 0)
        return fail(VND_ERR_INVALID, "the specialised kernel takes an even channel count (2..64) and a CSR tap table");
    SpecTable t;
    t.C = C;
    t.tap_off.assign(tap_offsets, tap_offsets + C + 1);
    const int32_t total = tap_offsets[C];
    if (total <= 0 || !tap_index || !tap_weight) return fail(VND_ERR_INVALID, "empty tap table");
    for (int32_t k = 0; k < total; ++k) {
        if (tap_index[k] < 0 || tap_index[k] >= (1 << 24) || !std::isfinite(tap_weight[k]))
            return fail(VND_ERR_UNSUPPORTED, "tap %d is outside the specialised kernel's scope", k);
        t.max_index = std::max(t.max_index, tap_index[k]);
    }
    t.idx.assign(tap_index, tap_index + total);
    t.w.assign(tap_weight, tap_weight + total);
    t.w_raw = t.w;
    SpecConfig cfg;
    if (!spec_pick_config(t, 160 * 1024, 0, 0, &cfg, false, false, mode == VND_MODE_EXACT)) return fail(VND_ERR_UNSUPPORTED, "halo does not fit the LDS ring");
    cfg.exact = mode == VND_MODE_EXACT ? 1 : 0;
    const std::string src = spec_prologue(t, cfg) + kSpecKernelSource;
    *bytes = (int64_t)src.size() + 1;
    if (!text) return VND_OK;                    // size query
    if (capacity < *bytes) return fail(VND_ERR_INVALID, "buffer too small: need %lld bytes", (long long)*bytes);
    memcpy(text, src.c_str(), src.size() + 1);
    return VND_OK;
}

vnd_status vnd_window_kernel_source(int32_t C, const int32_t *tap_offsets, const int32_t *tap_index,
                                    const float *tap_weight, const int32_t *seg_offsets, const int32_t *seg_end,
                                    const float *seg_gain, int32_t apply_gain, int32_t mode, int32_t frames_per_lane,
                                    int32_t threads, char *text, int64_t capacity, int64_t *bytes,
                                    int64_t *lds_bytes_per_tile, int64_t *fmas_per_tile)
{
    if (mode != VND_MODE_FAST && mode != VND_MODE_EXACT)
        return fail(VND_ERR_INVALID, "the specialised kernel exists for VND_MODE_FAST and VND_MODE_EXACT");
    if (!bytes) return fail(VND_ERR_INVALID, "null bytes pointer");
    if (C < 2 || (C & 1) || C > 64 || !tap_offsets || tap_offsets[0] != 0)
        return fail(VND_ERR_INVALID, "the window kernel takes a CSR tap table of whole channel pairs");
    SpecTable t;
    t.C = C;
    t.tap_off.assign(tap_offsets, tap_offsets + C + 1);
    const int32_t total = tap_offsets[C];
    if (total <= 0 || !tap_index || !tap_weight) return fail(VND_ERR_INVALID, "empty tap table");
    for (int32_t k = 0; k < total; ++k) {
        if (tap_index[k] < 0 || tap_index[k] >= (1 << 24) || !std::isfinite(tap_weight[k]))
            return fail(VND_ERR_UNSUPPORTED, "tap %d is outside the specialised kernel's scope", k);
        t.max_index = std::max(t.max_index, tap_index[k]);
    }
    t.idx.assign(tap_index, tap_index + total);
    t.w.assign(tap_weight, tap_weight + total);
    t.w_raw = t.w;
    if (seg_offsets) {
        if (!seg_end || !seg_gain || seg_offsets[0] != 0) return fail(VND_ERR_INVALID, "segment arrays incomplete");
        t.has_seg = true;
        t.apply_gain = apply_gain != 0;
        t.seg_off.assign(seg_offsets, seg_offsets + C + 1);
        t.seg_end.assign(seg_end, seg_end + seg_offsets[C]);
        t.seg_gain.assign(seg_gain, seg_gain + seg_offsets[C]);
        for (int c = 0; c < C; ++c) {
            int32_t prev = tap_offsets[c];
            for (int32_t sg = seg_offsets[c]; sg < seg_offsets[c + 1]; ++sg) {
                if (seg_end[sg] <= prev || seg_end[sg] > tap_offsets[c + 1]) return fail(VND_ERR_UNSUPPORTED, "empty or misplaced segment");
                prev = seg_end[sg];
                if (apply_gain)
                    for (int32_t k = (sg == seg_offsets[c] ? tap_offsets[c] : seg_end[sg - 1]); k < seg_end[sg]; ++k) t.w[k] = tap_weight[k] * seg_gain[sg];
            }
            if (prev != tap_offsets[c + 1]) return fail(VND_ERR_UNSUPPORTED, "segments do not cover the channel's taps");
        }
    }
    WinGeom g;
    // (tables of 4k channels: the quad / octet form, as the launches take it - VND_WIN_QUAD=0: channel pairs)
    bool quad = C % 8 == 0 && spec_env("VND_WIN_QUAD", 1) != 0 && spec_env("VND_WIN_OCTET", 1) != 0 &&
                win_geometry(t, frames_per_lane, threads, spec_env("VND_WIN_G", 8), false, 160 * 1024, &g, 2);
    quad = quad || (C % 4 == 0 && spec_env("VND_WIN_QUAD", 1) != 0 &&
                    win_geometry(t, frames_per_lane, threads, spec_env("VND_WIN_G", 8), false, 160 * 1024, &g, 1));
    const bool split = !quad && C == 2 && spec_env("VND_WIN_SPLIT", 0) != 0 &&
                       win_geometry(t, frames_per_lane, threads, spec_env("VND_WIN_G", 8), false, 160 * 1024, &g, 0, true);
    // (VND_WIN_SOURCE_FANOUT=1: the source of a mono input's fan-out launch through a stereo table - VW_BC)
    const bool bc = C == 2 && spec_env("VND_WIN_SOURCE_FANOUT", 0) != 0;
    if (bc && !win_geometry(t, frames_per_lane, threads, spec_env("VND_WIN_G", 8), true, 160 * 1024, &g, 0, split))
        return fail(VND_ERR_UNSUPPORTED, "this window geometry does not fit the LDS");
    if (!bc && !quad && !split && !win_geometry(t, frames_per_lane, threads, spec_env("VND_WIN_G", 8), false, 160 * 1024, &g))
        return fail(VND_ERR_UNSUPPORTED, "this window geometry does not fit the LDS");
    SpecConfig cfg;
    cfg.nt = threads; cfg.win = frames_per_lane; cfg.win_g = g.G; cfg.win_lds = (int)g.lds_bytes(); cfg.win_q = g.quad; cfg.win_s = g.split;
    cfg.la = spec_env("VND_SPEC_LA", (split && frames_per_lane >= 64) ? (mode == VND_MODE_EXACT ? 3 : 2) : (frames_per_lane >= 32 ? 4 : 6));      // (as win_pick_config)
    cfg.win_xpose = spec_env("VND_WIN_XPOSE_PAIRS", 1) != 0 ? 1 : 0;
    cfg.exact = mode == VND_MODE_EXACT ? 1 : 0;
    cfg.bc = bc ? 1 : 0;
    if (lds_bytes_per_tile || fmas_per_tile) {
        size_t lb = 0, fm = 0;
        if (cfg.exact) win_traffic_exact(t, frames_per_lane, &lb, &fm);
        else win_traffic(t, frames_per_lane, &lb, &fm);
        if (lds_bytes_per_tile) *lds_bytes_per_tile = (int64_t)lb;
        if (fmas_per_tile) *fmas_per_tile = (int64_t)fm;
    }
    const std::string src = win_source(t, g, cfg);
    *bytes = (int64_t)src.size() + 1;
    if (!text) return VND_OK;                    // size query
    if (capacity < *bytes) return fail(VND_ERR_INVALID, "buffer too small: need %lld bytes", (long long)*bytes);
    memcpy(text, src.c_str(), src.size() + 1);
    return VND_OK;
}

vnd_status vnd_code_object_private_bytes(const void *code, int64_t bytes, const char *kernel, int64_t *private_bytes)
{
    if (!code || bytes <= 0 || !kernel || !private_bytes) return fail(VND_ERR_INVALID, "bad arguments");
    const std::vector<char> image((const char *)code, (const char *)code + bytes);
    *private_bytes = spec_private_bytes(image, kernel);
    return VND_OK;
}

vnd_status vnd_set_variant(vnd_ctx *ctx, int32_t variant)
{
    if (!ctx) return fail(VND_ERR_INVALID, "null context");
    ctx->variant = variant;
    ctx->variant_nofuse = (variant >= 0 && ((variant >> 24) & 1)) ? 1 : 0;   // bit 24: unfused epilogue
    return VND_OK;
}

static vnd_status describe(vnd_ctx *ctx, const vnd_taps *t, int64_t batch, int64_t n, int32_t Cx, int32_t C,
                           int32_t mode, char *text, int32_t len)
{
    vnd_status st = check_shape(ctx, t, batch, n, C, mode, Cx);
    if (st != VND_OK) return st;
    if (!text || len <= 0) return fail(VND_ERR_INVALID, "null text buffer");
    // the pointers only decide alignment: describe the launch of 256-byte-aligned buffers (hipMalloc's)
    for (int attempt = 0; attempt < 8; ++attempt) {
        const SpecPlan sp = make_spec_plan(ctx, t, nullptr, nullptr, batch, n, C, Cx, mode, nullptr);
        if (!sp.use) break;
        DeviceScope on(ctx->device);
        SpecModule *m = spec_module(ctx, t, sp.cfg, !sp.eager);
        if (m && m->failed && sp.cfg.win && attempt < 7) {         // as launch(): plan again without that geometry
            if (getenv("VND_SPEC_VERBOSE")) fprintf(stderr, "vnd: window form (frames_per_lane=%d threads=%d) unavailable: %s\n", sp.cfg.win, sp.cfg.nt, m->log.c_str());
            continue;
        }
        if (m && !m->failed) {
            if (sp.cfg.win) {
                char split[96];
                if (sp.chunk_tiles > 0) snprintf(split, sizeof split, "a chunk of %d tiles per CU as %d + %d, %d chunks", sp.chunk_tiles, sp.chunk_len0, sp.chunk_tiles - sp.chunk_len0, sp.chunks_per_stream);
                else snprintf(split, sizeof split, "%d spans x %d tiles", sp.spans, sp.tiles_per_span);
                snprintf(text, (size_t)len,
                         "conv_spec%s_window (hipRTC, per table) frames_per_lane=%d tile=%d reads_ahead=%d "
                         "nt_stores=%d mode=%d lds=%zuB workgroups=%u (%u units: %s per stream) threads=%d store_phase=%s",
                         sp.cfg.exact ? "_exact" : "", sp.cfg.win, sp.cfg.tile(), sp.cfg.la, sp.cfg.nt_stores, mode,
                         sp.cfg.lds_bytes(), sp.nblocks, sp.units, split, sp.cfg.nt,
                         sp.cfg.win_s ? "planar waves=split-by-channel" : sp.cfg.win_q == 2 ? "planar pieces=channel-octets waves=split-by-channel" : (sp.cfg.win_q ? "planar pieces=channel-quads waves=split-by-channel" : (sp.cfg.win_xpose ? "frame-pairs" : "planar")));
                return VND_OK;
            }
            snprintf(text, (size_t)len,
                     "conv_spec%s (hipRTC, per table) pairs_per_lane=%d tile=%d ring_slots=%d prefetch=%d reads_ahead=%d "
                     "nt_stores=%d mode=%d lds=%zuB workgroups=%u (%u units: %d spans x %d tiles per stream) threads=%d",
                     sp.cfg.exact ? "_exact" : "", sp.cfg.rr, sp.cfg.tile(), sp.cfg.pp, sp.cfg.dd, sp.cfg.la, sp.cfg.nt_stores, mode,
                     sp.cfg.lds_bytes(), sp.nblocks, sp.units, sp.spans, sp.tiles_per_span, sp.cfg.nt);
            return VND_OK;
        }
        if (m && getenv("VND_SPEC_VERBOSE")) fprintf(stderr, "vnd: specialised kernel unavailable: %s\n", m->log.c_str());
        break;
    }
    const Plan p = make_plan(ctx, t, batch, n, C, mode, Cx);
    if (p.direct)
        snprintf(text, (size_t)len, "conv_direct mode=%d blocks=%u threads=%d", mode, p.nblocks, kDirectThreads);
    else
        snprintf(text, (size_t)len,
                 "%s%s cg=%d pairs_per_lane=%d tile=%d halo=%d mode=%d lds=%zuB workgroups=%u threads=%d",
                 mode == VND_MODE_FAST ? "conv_fast" : "conv_ordered", p.bc ? "_fanout" : "", p.cg, p.r,
                 2 * p.nt * p.r, p.W - 2 * p.nt * p.r, mode, p.lds_bytes, p.nblocks, p.nt);
    return VND_OK;
}

vnd_status vnd_describe_launch(vnd_ctx *ctx, const vnd_taps *t, int64_t batch, int64_t n, int32_t C,
                               int32_t mode, char *text, int32_t len)
{
    return describe(ctx, t, batch, n, C, C, mode, text, len);
}

vnd_status vnd_describe_fanout_launch(vnd_ctx *ctx, const vnd_taps *t, int64_t batch, int64_t n,
                                      int32_t in_channels, int32_t mode, char *text, int32_t len)
{
    if (!t) return fail(VND_ERR_INVALID, "null context or tap table");
    if (in_channels <= 0) return fail(VND_ERR_INVALID, "in_channels must be positive");
    return describe(ctx, t, batch, n, in_channels, t->C, mode, text, len);
}

vnd_status vnd_prepare_launch(vnd_ctx *ctx, const vnd_taps *t, int64_t batch, int64_t n, int32_t in_channels, int32_t mode)
{
    if (!ctx || !t) return fail(VND_ERR_INVALID, "null context or tap table");
    if (in_channels <= 0) return fail(VND_ERR_INVALID, "in_channels must be positive");
    vnd_status st = check_shape(ctx, t, batch, n, t->C, mode, in_channels);
    if (st != VND_OK) return st;
    if (batch == 0 || n == 0) return VND_OK;
    DeviceScope on(ctx->device);
    for (int attempt = 0; attempt < 8; ++attempt) {               // (a window geometry that does not build is skipped: plan again)
        const SpecPlan sp = make_spec_plan(ctx, t, nullptr, nullptr, batch, n, t->C, in_channels, mode, nullptr);
        if (!sp.use) return VND_OK;
        SpecModule *m = spec_module(ctx, t, sp.cfg, false);
        if (m && !m->failed) return VND_OK;
        if (!sp.cfg.win) break;
    }
    return VND_OK;                                                 // the generic kernels take such launches
}

vnd_status vnd_debug_read_stamps(vnd_ctx *ctx, const vnd_taps *t, int64_t batch, int64_t n, int32_t in_channels, int32_t mode,
                                 uint64_t *stamps, int64_t capacity, int64_t *count)
{
    if (!ctx || !t || !count || capacity < 0 || (capacity > 0 && !stamps)) return fail(VND_ERR_INVALID, "bad arguments");
    *count = 0;
    vnd_status st = check_shape(ctx, t, batch, n, t->C, mode, in_channels);
    if (st != VND_OK) return st;
    DeviceScope on(ctx->device);
    const SpecPlan sp = make_spec_plan(ctx, t, nullptr, nullptr, batch, n, t->C, in_channels, mode, nullptr);
    if (!sp.use || !sp.cfg.win) return VND_OK;
    SpecModule *m = spec_module(ctx, t, sp.cfg, true);
    if (!m || m->failed || !m->module) return VND_OK;
    hipDeviceptr_t at = nullptr;
    size_t bytes = 0;
    if (hipModuleGetGlobal(&at, &bytes, m->module, "vw_stamps") != hipSuccess) { (void)hipGetLastError(); return VND_OK; }
    *count = (int64_t)(bytes / sizeof(uint64_t));
    const size_t take = std::min<size_t>(bytes, (size_t)capacity * sizeof(uint64_t));
    HIP_TRY(hipDeviceSynchronize());
    if (take) HIP_TRY(hipMemcpy(stamps, at, take, hipMemcpyDeviceToHost));
    return VND_OK;
}

static int64_t epi_chunks(int64_t n) { return (n + kEpiChunk - 1) / kEpiChunk; }

// rows of partial sums per stream: pass-1 chunks, or - fused - one row per tile (>= 512 frames each)
static int64_t epi_rows_max(int64_t n) { return std::max<int64_t>(epi_chunks(n), (n + 511) / 512 + 1); }

static int64_t par_blocks(int64_t n) { return std::max<int64_t>((n + kParFrames - 1) / kParFrames, 1); }

static int64_t pw_chunks(int64_t n) { return std::max<int64_t>((n + kPwChunk - 1) / kPwChunk, 1); }

vnd_status vnd_decorrelate_workspace_bytes(int64_t batch, int64_t n, int32_t C, int64_t *bytes)
{
    if (!bytes || batch < 0 || n < 0 || C <= 0) return fail(VND_ERR_INVALID, "bad workspace query");
    *bytes = batch * epi_rows_max(n) * 2 * C * (int64_t)sizeof(double) + batch * C * (int64_t)sizeof(float) + 16;
    // the parallel exact sums of a stereo table: per stream and chain, a float64 sum and a record per block
    if (C == 2) *bytes += 32 + batch * 4 * (par_blocks(n) * (int64_t)(sizeof(double) + sizeof(ParRec) + sizeof(ParGrp)) + (int64_t)sizeof(float));
    // the pairwise sums of a single-channel table: one float per (stream, array, 8192-sample chunk)
    if (C == 1) *bytes += 32 + batch * 2 * pw_chunks(n) * (int64_t)sizeof(float);
    return VND_OK;
}

static vnd_status decorrelate_dev(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch,
                                  int64_t n, int32_t Cx, int32_t C, int32_t mode, int32_t ms_encode,
                                  int32_t use_width, double width, int32_t normalize, float eps, void *workspace,
                                  int64_t workspace_bytes, void *stream_)
{
    vnd_status st = check_shape(ctx, t, batch, n, C, mode, Cx);
    if (st != VND_OK) return st;
    if (batch == 0 || n == 0) return VND_OK;
    if (!x || !y) return fail(VND_ERR_INVALID, "null signal pointer");
    if (overlaps(x, batch * n * Cx, y, batch * n * C)) return fail(VND_ERR_INVALID, "x and y overlap");
    if ((ms_encode || use_width) && C != 2)
        return fail(VND_ERR_INVALID, "side-channel encode and stereo width need 2 channels, got %d", C);
    int64_t need = 0;
    vnd_decorrelate_workspace_bytes(batch, n, C, &need);
    if (normalize && (!workspace || workspace_bytes < need))
        return fail(VND_ERR_INVALID, "workspace too small: need %lld bytes", (long long)need);
    if (batch > VND_MAX_STREAMS) return fail(VND_ERR_UNSUPPORTED, "more than %d streams per call: split the batch", VND_MAX_STREAMS);
    DeviceScope on(ctx->device);
    hipStream_t stream = (hipStream_t)stream_;
    const bool any = ms_encode || use_width || normalize;

    EArgs e{};
    e.x = x; e.y = y; e.partials = (double *)workspace; e.n = n; e.C = C; e.Cx = Cx;
    e.scales = (float *)((double *)workspace + batch * epi_rows_max(n) * 2 * C);
    e.ms_encode = ms_encode ? 1 : 0; e.use_width = use_width ? 1 : 0;
    e.w_mid = (float)(1.0 - width); e.w_side = (float)width;   // float32(python float), as NumPy's in-place multiply
    e.normalize = normalize ? 1 : 0; e.eps = eps;
    e.wide = (((uintptr_t)y & 15) == 0 && ((uintptr_t)x & (Cx == 1 ? 7 : 15)) == 0 && (batch == 1 || n % 2 == 0) && spec_env("VND_EPI_WIDE", 1) != 0) ? 1 : 0;
    const dim3 grid((unsigned)epi_chunks(n), (unsigned)batch);

    // Fused form: the fast kernel applies the pointwise steps and writes one row of sums per tile.
    const Plan p = make_plan(ctx, t, batch, n, C, mode, Cx);
    // normalize == VND_NORMALIZE_RMS_REFERENCE_ORDER: the sums of squares in NumPy's own (sequential
    // float32) order in every mode, so that the scale differs from the reference's only through y
    // frames per staged block of the sums kernel: as many as the 2C rows of squares leave room for
    const int seq_frames = C == 2 ? kSeqFramesStereo
                         : ((size_t)2 * C * kSeqFrames * sizeof(float) <= (size_t)ctx->lds_limit ? kSeqFrames : kSeqFramesWide);
    const bool seq_ok = normalize && C >= 2 && 2 * C <= 64 &&
                        (size_t)2 * C * seq_frames * sizeof(float) <= (size_t)ctx->lds_limit;
    // a single-channel table: NumPy sums that array pairwise (rms_pairwise_kernel); the flow is the same
    const bool pair_ok = normalize && C == 1 && Cx == 1;
    const bool want_seq = (seq_ok || pair_ok) && (mode == VND_MODE_EXACT || normalize == VND_NORMALIZE_RMS_REFERENCE_ORDER);
    const bool fused = any && mode == VND_MODE_FAST && ctx->variant_nofuse == 0 && fast_epi_kernel(p) != nullptr &&
                       (!(ms_encode || use_width) || p.cg == 2) && !(want_seq && !(ms_encode || use_width));
    // stereo: the reference-order sums parallel over the stream's 2048-frame blocks (vnd_epilogue.hpp, rms_par_*).  They start from
    // per-block sums of squares (predictions of the running sum's binade) - which the window kernel's store phase leaves on its way
    // (x still in the ring, the finished y in registers: EpiFuse::blk_sum) where that kernel runs; rms_par_sum_kernel reads both
    // arrays for them otherwise.  Which form, by batch (tools/rms_batch_rate.py, 10 s signals, ms per stage: per-stream / block-parallel):
    //   up to 64 streams the one-workgroup-per-stream kernel leaves most CUs dark (16: 0.49 / 0.17);
    //   65 .. 255: it still fills less than every CU once (128: 0.80 / 0.82, and 0.66 once the block sums come from the convolution);
    //   256 and more: it fills the chip by itself and reads the data once instead of twice (1024: 4.36 / 6.76).
    // variant bit 19 keeps the per-stream kernel, bit 17 forces the block-parallel form (A/B runs).
    const bool par_ok = want_seq && C == 2 && par_blocks(n) <= kParMaxBlocks && !(ctx->variant >= 0 && ((ctx->variant >> 19) & 1));
    const bool par_forced = ctx->variant >= 0 && ((ctx->variant >> 17) & 1);
    RArgs r{};
    int conv_path = 0;                                     // EpiFuse::path of the convolution launch
    if (par_ok) {
        r.x = x; r.y = y; r.n = n; r.Cx = Cx; r.nblocks = (int32_t)par_blocks(n);
        char *extra = (char *)((float *)((double *)workspace + batch * epi_rows_max(n) * 2 * C) + batch * C);
        extra += (16 - ((uintptr_t)extra & 15)) & 15;
        r.blk_sum = (double *)extra;
        r.rec = (ParRec *)(r.blk_sum + batch * 4 * (int64_t)r.nblocks);
        r.grp = (ParGrp *)(r.rec + batch * 4 * (int64_t)r.nblocks);
        r.first = (float *)(r.grp + batch * 4 * (int64_t)r.nblocks);
        r.partials = (double *)workspace;
        r.prefixed = r.nblocks > kParPrefixBlocks ? 1 : 0;
        r.wide = e.wide;
    }
    const bool want_blk = par_ok && (batch < 256 || par_forced) && spec_env("VND_EPI_BLOCK_SUMS", 1) != 0;
    bool sums_pending = false;                             // the sequential sums still have to run
    if (fused) {
        // with reference-order sums the fused kernel only applies the pointwise steps
        EpiFuse f{(double *)workspace, e.ms_encode, e.use_width, want_seq ? 0 : e.normalize, e.w_mid, e.w_side};
        f.path = &conv_path;
        if (want_blk && want_seq) { f.blk_sum = r.blk_sum; f.nblocks = r.nblocks; }
        // the fully fused stage: the window kernel writes one row of sums per 2048-frame block where it runs (the generic fast
        // kernel one per tile), and one streaming pass scales
        else if (!want_seq && e.normalize && C == 2 && spec_env("VND_EPI_BLOCK_SUMS", 1) != 0) { f.blk_sum = (double *)workspace; f.nblocks = (int)par_blocks(n); f.rows_major = 1; }
        st = launch(ctx, t, x, y, batch, n, C, mode, stream, &f, Cx);
        if (st != VND_OK) return st;
        e.rows = p.tiles;
        if (!want_seq && e.normalize && conv_path == 1) e.rows = (int32_t)par_blocks(n);
        if (!want_seq && e.normalize && conv_path == 2) {
            // (the pair-read per-table kernel took the launch: pointwise steps done, no sums - one more pass for them)
            e.ms_encode = e.use_width = 0;
            e.rows = (int32_t)epi_chunks(n);
            hipLaunchKernelGGL(epilogue_pointwise_kernel, grid, dim3(kEpiThreads), 0, stream, e);
        }
        sums_pending = want_seq;
    } else {
        // table-order modes: the pointwise steps ride in the ordered kernel's store phase when the
        // plan has both channels in one workgroup; the sums of squares follow as their own pass
        const bool pointwise = ms_encode || use_width;
        const bool in_kernel = pointwise && mode != VND_MODE_FAST && ctx->variant_nofuse == 0 &&
                               ordered_epi_kernel(p, arithmetic_of(t, mode)) != nullptr;
        if (in_kernel) {
            EpiFuse f{nullptr, e.ms_encode, e.use_width, 0, e.w_mid, e.w_side};
            f.path = &conv_path;
            if (want_blk) { f.blk_sum = r.blk_sum; f.nblocks = r.nblocks; }
            st = launch(ctx, t, x, y, batch, n, C, mode, stream, &f, Cx);
            e.ms_encode = e.use_width = 0;                 // done
        } else {
            st = launch(ctx, t, x, y, batch, n, C, mode, stream, nullptr, Cx);
        }
        if (st != VND_OK || !any) return st;
        // reference-order sums (always in VND_MODE_EXACT, C >= 2: the bit-identical stage); C == 1 is
        // summed pairwise by NumPy and keeps the float64 sums.
        const bool seq = want_seq;
        e.rows = seq ? 1 : (int32_t)epi_chunks(n);
        if (seq) e.normalize = 0;                          // pointwise pass without its partial sums
        if (e.ms_encode || e.use_width || (normalize && !seq))
            hipLaunchKernelGGL(epilogue_pointwise_kernel, grid, dim3(kEpiThreads), 0, stream, e);
        sums_pending = seq;
    }
    const bool blk_done = conv_path == 1 && want_blk;
    const bool par_sums = sums_pending && par_ok && (batch <= 64 || par_forced || blk_done);
    if (par_sums) {
        e.rows = 1;
        e.exact_rms = 1;
        e.normalize = 1;
        const dim3 pgrid((unsigned)r.nblocks, (unsigned)batch), tgrid((unsigned)r.nblocks, (unsigned)batch);     // tally: blocks 1.., plus block 0's chain
        const dim3 sgrid((unsigned)(batch * 4));
        if (Cx == 1) {
            if (!blk_done) hipLaunchKernelGGL(rms_par_sum_kernel<true>, pgrid, dim3(kParThreads), 0, stream, r);
            if (r.prefixed) hipLaunchKernelGGL(rms_par_prefix_kernel, dim3((unsigned)(batch * 4)), dim3(kParThreads), 0, stream, r);
            hipLaunchKernelGGL(rms_par_tally_kernel<true>, tgrid, dim3(kParThreads), 0, stream, r);
            hipLaunchKernelGGL(rms_par_stitch_kernel<true>, sgrid, dim3(64), 0, stream, r);
        } else {
            if (!blk_done) hipLaunchKernelGGL(rms_par_sum_kernel<false>, pgrid, dim3(kParThreads), 0, stream, r);
            if (r.prefixed) hipLaunchKernelGGL(rms_par_prefix_kernel, dim3((unsigned)(batch * 4)), dim3(kParThreads), 0, stream, r);
            hipLaunchKernelGGL(rms_par_tally_kernel<false>, tgrid, dim3(kParThreads), 0, stream, r);
            hipLaunchKernelGGL(rms_par_stitch_kernel<false>, sgrid, dim3(64), 0, stream, r);
        }
    } else if (sums_pending && C == 1) {
        e.rows = 1;
        e.exact_rms = 1;
        e.normalize = 1;
        PwArgs q{};
        q.x = x; q.y = y; q.n = n; q.nchunks = (int32_t)pw_chunks(n);
        char *extra = (char *)(e.scales + batch * C);
        extra += (16 - ((uintptr_t)extra & 15)) & 15;
        q.chunk_sums = (float *)extra;
        q.partials = e.partials;
        hipLaunchKernelGGL(rms_pairwise_kernel, dim3((unsigned)q.nchunks, (unsigned)batch), dim3(2 * kPwThreads), 0, stream, q);
        hipLaunchKernelGGL(rms_pairwise_fold_kernel, dim3((unsigned)batch), dim3(64), 0, stream, q);
    } else if (sums_pending) {
        e.rows = 1;
        e.exact_rms = 1;
        e.normalize = 1;
        const size_t lds = (size_t)2 * C * seq_frames * sizeof(float);
        const int waves = C == 2 ? 4 : std::min(2 * C, kSeqMaxWaves);
        auto k = C == 2 ? (Cx == 1 ? epilogue_rms_seq_kernel<true, true> : epilogue_rms_seq_kernel<true, false>)
                        : (seq_frames == kSeqFrames ? epilogue_rms_seq_kernel<false, false, kSeqFrames>
                                                    : epilogue_rms_seq_kernel<false, false, kSeqFramesWide>);
        if (lds > 65536) HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      ctx->lds_limit));
        hipLaunchKernelGGL(k, dim3((unsigned)batch), dim3(64 * waves), lds, stream, e);
    }
    if (normalize) {
        hipLaunchKernelGGL(epilogue_reduce_kernel, dim3((unsigned)batch), dim3(kEpiThreads), 0, stream, e);
        hipLaunchKernelGGL(epilogue_scale_kernel, grid, dim3(kEpiThreads), 0, stream, e);
    }
    HIP_TRY(hipGetLastError());
    return VND_OK;
}

static vnd_status decorrelate_host(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch,
                                   int64_t n, int32_t Cx, int32_t C, int32_t mode, int32_t ms_encode,
                                   int32_t use_width, double width, int32_t normalize, float eps)
{
    vnd_status st = check_shape(ctx, t, batch, n, C, mode, Cx);
    if (st != VND_OK) return st;
    if (batch == 0 || n == 0) return VND_OK;
    if (!x || !y) return fail(VND_ERR_INVALID, "null signal pointer");
    HostLock lock(ctx->host_mutex);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t in_elems = (size_t)batch * n * Cx, out_elems = (size_t)batch * n * C;
    st = ensure_scratch(ctx, out_elems);
    if (st != VND_OK) return st;
    int64_t ws = 0;
    vnd_decorrelate_workspace_bytes(batch, n, C, &ws);
    st = ensure_work(ctx, (size_t)ws);
    if (st != VND_OK) return st;
    const int chunks = host_chunks(batch, (in_elems + out_elems) * sizeof(float));
    if (chunks > 1) {
        // one workspace per pipeline lane: the two lanes' epilogues run side by side
        const int64_t per = (batch + chunks - 1) / chunks;
        vnd_decorrelate_workspace_bytes(per, n, C, &ws);
        ws = (ws + 255) & ~(int64_t)255;
        st = ensure_work(ctx, (size_t)ws * 2);
        if (st != VND_OK) return st;
    }
    hipError_t e = hipSuccess;
    for (int c = 0; c < chunks && st == VND_OK && e == hipSuccess; ++c) {
        const int64_t b0 = batch * c / chunks, b1 = batch * (c + 1) / chunks;
        if (b1 == b0) continue;
        hipStream_t s = (c & 1) ? ctx->stream2 : ctx->stream;
        const size_t xo = (size_t)b0 * n * Cx, yo = (size_t)b0 * n * C;
        e = hipMemcpyAsync(ctx->scratch_x + xo, x + xo, (size_t)(b1 - b0) * n * Cx * sizeof(float), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) break;
        st = decorrelate_dev(ctx, t, ctx->scratch_x + xo, ctx->scratch_y + yo, b1 - b0, n, Cx, C, mode, ms_encode, use_width,
                             width, normalize, eps, ctx->work + (size_t)(c & 1) * (size_t)ws, ws, s);
        if (st != VND_OK) break;
        e = hipMemcpyAsync(y + yo, ctx->scratch_y + yo, (size_t)(b1 - b0) * n * C * sizeof(float), hipMemcpyDeviceToHost, s);
    }
    // on any failure too: nothing of this call stays in flight behind its return (see convolve_host)
    const hipError_t s1 = hipStreamSynchronize(ctx->stream), s2 = hipStreamSynchronize(ctx->stream2);
    if (st != VND_OK) return st;
    if (e == hipSuccess) e = s1 != hipSuccess ? s1 : s2;
    if (e != hipSuccess) return fail(VND_ERR_HIP, "host pipeline failed: %s", hipGetErrorString(e));
    return VND_OK;
}

vnd_status vnd_decorrelate_f32_dev(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch,
                                   int64_t n, int32_t C, int32_t mode, int32_t ms_encode, int32_t use_width,
                                   double width, int32_t normalize, float eps, void *workspace,
                                   int64_t workspace_bytes, void *stream)
{
    return decorrelate_dev(ctx, t, x, y, batch, n, C, C, mode, ms_encode, use_width, width, normalize, eps,
                           workspace, workspace_bytes, stream);
}

vnd_status vnd_decorrelate_f32_host(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y, int64_t batch,
                                    int64_t n, int32_t C, int32_t mode, int32_t ms_encode, int32_t use_width,
                                    double width, int32_t normalize, float eps)
{
    return decorrelate_host(ctx, t, x, y, batch, n, C, C, mode, ms_encode, use_width, width, normalize, eps);
}

vnd_status vnd_decorrelate_fanout_f32_dev(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y,
                                          int64_t batch, int64_t n, int32_t in_channels, int32_t mode,
                                          int32_t ms_encode, int32_t use_width, double width, int32_t normalize,
                                          float eps, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!t) return fail(VND_ERR_INVALID, "null context or tap table");
    if (in_channels <= 0) return fail(VND_ERR_INVALID, "in_channels must be positive");
    return decorrelate_dev(ctx, t, x, y, batch, n, in_channels, t->C, mode, ms_encode, use_width, width, normalize,
                           eps, workspace, workspace_bytes, stream);
}

vnd_status vnd_decorrelate_fanout_f32_host(vnd_ctx *ctx, const vnd_taps *t, const float *x, float *y,
                                           int64_t batch, int64_t n, int32_t in_channels, int32_t mode,
                                           int32_t ms_encode, int32_t use_width, double width, int32_t normalize,
                                           float eps)
{
    if (!t) return fail(VND_ERR_INVALID, "null context or tap table");
    if (in_channels <= 0) return fail(VND_ERR_INVALID, "in_channels must be positive");
    return decorrelate_host(ctx, t, x, y, batch, n, in_channels, t->C, mode, ms_encode, use_width, width,
                            normalize, eps);
}

vnd_status vnd_convolve_promote_host(vnd_ctx *ctx, int32_t C, const int32_t *tap_offsets, const int32_t *tap_index,
                                     const double *tap_weight, const void *x, int32_t x_is_f64, float *y,
                                     int64_t batch, int64_t n)
{
    if (!ctx) return fail(VND_ERR_INVALID, "null context");
    if (C <= 0 || batch < 0 || n < 0) return fail(VND_ERR_INVALID, "bad channel, batch or frame count");
    if (!tap_offsets || tap_offsets[0] != 0) return fail(VND_ERR_INVALID, "bad tap_offsets");
    for (int c = 0; c < C; ++c)
        if (tap_offsets[c + 1] < tap_offsets[c]) return fail(VND_ERR_INVALID, "tap_offsets not monotone");
    const int32_t taps = tap_offsets[C];
    if (taps > 0 && (!tap_index || !tap_weight)) return fail(VND_ERR_INVALID, "null tap arrays");
    for (int32_t k = 0; k < taps; ++k) {
        if (tap_index[k] < 0) return fail(VND_ERR_INVALID, "negative tap index at %d", k);
        if (tap_index[k] > (1 << 30)) return fail(VND_ERR_UNSUPPORTED, "tap index %d at %d is beyond 2^30 frames", tap_index[k], k);
    }
    const int64_t total = batch * n * C;
    if (total == 0) return VND_OK;
    if (!x || !y) return fail(VND_ERR_INVALID, "null signal pointer");
    HostLock lock(ctx->host_mutex);
    HIP_TRY(hipSetDevice(ctx->device));
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t xb = (size_t)total * (x_is_f64 ? 8 : 4), yb = (size_t)total * 4;
    const size_t wb = (size_t)taps * 8, ob = (size_t)(C + 1) * 4, ib = (size_t)taps * 4;
    vnd_status st = ensure_work(ctx, up16(xb) + up16(yb) + up16(wb) + up16(ob) + up16(ib) + 16);
    if (st != VND_OK) return st;
    char *p = ctx->work;                                   // hipMalloc'ed: 256-byte aligned
    PArgs a{};
    a.x = p;
    a.y = (float *)(p + up16(xb));
    a.w = (const double *)((const char *)a.y + up16(yb));
    a.tap_off = (const int32_t *)((const char *)a.w + up16(wb));
    a.idx = (const int32_t *)((const char *)a.tap_off + up16(ob));
    a.n = n; a.total = total; a.C = C; a.x_is_f64 = x_is_f64 ? 1 : 0;
    HIP_TRY(hipMemcpyAsync(p, x, xb, hipMemcpyHostToDevice, ctx->stream));
    if (taps) {
        HIP_TRY(hipMemcpyAsync((void *)a.w, tap_weight, (size_t)taps * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync((void *)a.idx, tap_index, (size_t)taps * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(hipMemcpyAsync((void *)a.tap_off, tap_offsets, (size_t)(C + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    const int64_t blocks = std::min<int64_t>((total + kDirectThreads - 1) / kDirectThreads, (int64_t)cus * 32);
    hipLaunchKernelGGL(conv_promote_kernel, dim3((unsigned)blocks), dim3(kDirectThreads), 0, ctx->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(y, a.y, yb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VND_OK;
}

// ------------------------------------------------------------------------------
// candidate scan
// ------------------------------------------------------------------------------
static int64_t mom_chunks(int64_t n) { return std::max<int64_t>((n + kMomFrames - 1) / kMomFrames, 1); }

vnd_status vnd_polar_moments_workspace_bytes(int64_t n, int32_t n_pairs, int64_t *bytes)
{
    if (!bytes || n < 0 || n_pairs <= 0) return fail(VND_ERR_INVALID, "bad workspace query");
    *bytes = mom_chunks(n) * n_pairs * kMoments * (int64_t)sizeof(double);
    return VND_OK;
}

vnd_status vnd_polar_moments_f32_dev(vnd_ctx *ctx, const float *y, int64_t n, int32_t n_pairs, double *moments,
                                     void *workspace, int64_t workspace_bytes, void *stream_)
{
    if (!ctx) return fail(VND_ERR_INVALID, "null context");
    if (n < 0 || n_pairs <= 0) return fail(VND_ERR_INVALID, "bad frame or pair count");
    if (!moments || (n > 0 && !y)) return fail(VND_ERR_INVALID, "null pointer");
    int64_t need = 0;
    vnd_polar_moments_workspace_bytes(n, n_pairs, &need);
    if (!workspace || workspace_bytes < need)
        return fail(VND_ERR_INVALID, "workspace too small: need %lld bytes", (long long)need);
    const int64_t chunks = mom_chunks(n);
    if (chunks > 0x7fffffffLL || n_pairs > 65535 * kMomThreads)
        return fail(VND_ERR_UNSUPPORTED, "scan too large; split it");
    DeviceScope on(ctx->device);
    hipStream_t stream = (hipStream_t)stream_;
    MArgs a{};
    a.y = y; a.partials = (double *)workspace; a.moments = moments; a.n = n; a.F = n_pairs; a.chunks = (int32_t)chunks;
    if (n_pairs >= 64) {
        const dim3 grid((unsigned)chunks, (unsigned)((n_pairs + kMomThreads - 1) / kMomThreads));
        hipLaunchKernelGGL(moments_by_candidate_kernel, grid, dim3(kMomThreads), 0, stream, a);
    } else {
        hipLaunchKernelGGL(moments_by_frame_kernel, dim3((unsigned)chunks, (unsigned)n_pairs), dim3(kMomThreads), 0,
                           stream, a);
    }
    hipLaunchKernelGGL(moments_reduce_kernel, dim3((unsigned)n_pairs), dim3(kMomThreads), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return VND_OK;
}

vnd_status vnd_scan_bank_f32_host(vnd_ctx *ctx, const vnd_taps *t, const float *x, int64_t n, int32_t in_channels,
                                  int32_t mode, double *moments)
{
    if (!ctx || !t) return fail(VND_ERR_INVALID, "null context or tap table");
    if (t->C % 2 != 0) return fail(VND_ERR_INVALID, "a scan needs stereo pairs: the bank has %d channels", t->C);
    if (in_channels != 1 && in_channels != 2)
        return fail(VND_ERR_INVALID, "a scan takes a mono or stereo signal, got %d channels", in_channels);
    vnd_status st = check_shape(ctx, t, 1, n, t->C, mode, in_channels);
    if (st != VND_OK) return st;
    if (!moments || (n > 0 && !x)) return fail(VND_ERR_INVALID, "null pointer");
    const int32_t pairs = t->C / 2;
    HostLock lock(ctx->host_mutex);
    HIP_TRY(hipSetDevice(ctx->device));
    // Fused form: the convolution kernel's store phase reduces each tile to the eight moments per
    // candidate (KArgs.sink_partials) - the [n][2F] output, 1.6 GB there and back for 400 candidates
    // of a 5.7 s signal, is never written.  Needs the two-channels-per-workgroup epilogue instantiation.
    if (n > 0 && !(ctx->variant >= 0 && ((ctx->variant >> 16) & 0x100))) {
        const Plan p = make_plan(ctx, t, 1, n, t->C, mode, in_channels);
        kern_t k = p.direct ? nullptr
                            : (mode == VND_MODE_FAST ? fast_epi_kernel(p) : ordered_epi_kernel(p, arithmetic_of(t, mode)));
        if (k != nullptr && p.cg == 2) {
            const size_t part_bytes = (size_t)p.tiles * pairs * kMoments * sizeof(double);
            const size_t out_bytes = (size_t)pairs * kMoments * sizeof(double);
            st = ensure_scratch(ctx, (size_t)n * in_channels);
            if (st != VND_OK) return st;
            st = ensure_work(ctx, part_bytes + out_bytes);
            if (st != VND_OK) return st;
            HIP_TRY(hipMemcpyAsync(ctx->scratch_x, x, (size_t)n * in_channels * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
            EpiFuse f{nullptr, 0, 0, 0, 0.0f, 0.0f, (double *)ctx->work};
            st = launch(ctx, t, ctx->scratch_x, ctx->scratch_y, 1, n, t->C, mode, ctx->stream, &f, in_channels);
            if (st != VND_OK) return st;
            MArgs m{};
            m.partials = (double *)ctx->work; m.moments = (double *)(ctx->work + part_bytes); m.n = n; m.F = pairs; m.chunks = p.tiles;
            hipLaunchKernelGGL(moments_reduce_kernel, dim3((unsigned)pairs), dim3(kMomThreads), 0, ctx->stream, m);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(moments, ctx->work + part_bytes, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            return VND_OK;
        }
    }
    st = ensure_scratch(ctx, (size_t)std::max<int64_t>(n, 1) * t->C);
    if (st != VND_OK) return st;
    int64_t ws = 0;
    vnd_polar_moments_workspace_bytes(n, pairs, &ws);
    const size_t out_bytes = (size_t)pairs * kMoments * sizeof(double);
    st = ensure_work(ctx, (size_t)ws + out_bytes);
    if (st != VND_OK) return st;
    char *work = ctx->work;
    if (n > 0)
        HIP_TRY(hipMemcpyAsync(ctx->scratch_x, x, (size_t)n * in_channels * sizeof(float), hipMemcpyHostToDevice,
                               ctx->stream));
    st = launch(ctx, t, ctx->scratch_x, ctx->scratch_y, 1, n, t->C, mode, ctx->stream, nullptr, in_channels);
    if (st != VND_OK) return st;
    st = vnd_polar_moments_f32_dev(ctx, ctx->scratch_y, n, pairs, (double *)(work + ws), work, ws, ctx->stream);
    if (st != VND_OK) return st;
    HIP_TRY(hipMemcpyAsync(moments, work + ws, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VND_OK;
}

// ------------------------------------------------------------------------------
// HaasEffect on the device
// ------------------------------------------------------------------------------
static vnd_status haas_check(const vnd_ctx *ctx, int64_t batch, int64_t n, int32_t in_channels, int32_t delay,
                             int32_t delayed_channel)
{
    if (!ctx) return fail(VND_ERR_INVALID, "null context");
    if (batch < 0 || n < 0 || delay < 0) return fail(VND_ERR_INVALID, "negative batch, frame count or delay");
    if (in_channels != 1 && in_channels != 2)
        return fail(VND_ERR_INVALID, "HaasEffect takes a mono or stereo signal, got %d channels", in_channels);
    if (delayed_channel != 0 && delayed_channel != 1)
        return fail(VND_ERR_INVALID, "delayed_channel must be 0 or 1, got %d", delayed_channel);
    if (batch > VND_MAX_STREAMS) return fail(VND_ERR_UNSUPPORTED, "more than %d streams per call: split the batch", VND_MAX_STREAMS);
    return VND_OK;
}

vnd_status vnd_haas_f64_dev(vnd_ctx *ctx, const float *x, double *y, int64_t batch, int64_t n, int32_t in_channels,
                            int32_t delay, int32_t delayed_channel, int32_t ms_mode, int32_t use_width,
                            double width, void *stream)
{
    vnd_status st = haas_check(ctx, batch, n, in_channels, delay, delayed_channel);
    if (st != VND_OK) return st;
    const int64_t total = n + delay;
    if (batch == 0 || total == 0) return VND_OK;
    if (!y || (n > 0 && !x)) return fail(VND_ERR_INVALID, "null signal pointer");
    DeviceScope on(ctx->device);
    HArgs a{};
    a.x = x; a.y = y; a.n = n; a.Cx = in_channels; a.delay = delay; a.delayed_channel = delayed_channel;
    a.ms = ms_mode ? 1 : 0; a.use_width = use_width ? 1 : 0; a.w_mid = 1.0 - width; a.w_side = width;
    const dim3 grid((unsigned)((total + kHaasThreads - 1) / kHaasThreads), (unsigned)batch);
    hipLaunchKernelGGL(haas_kernel, grid, dim3(kHaasThreads), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return VND_OK;
}

vnd_status vnd_haas_f64_host(vnd_ctx *ctx, const float *x, double *y, int64_t batch, int64_t n, int32_t in_channels,
                             int32_t delay, int32_t delayed_channel, int32_t ms_mode, int32_t use_width,
                             double width)
{
    vnd_status st = haas_check(ctx, batch, n, in_channels, delay, delayed_channel);
    if (st != VND_OK) return st;
    const int64_t total = n + delay;
    if (batch == 0 || total == 0) return VND_OK;
    if (!y || (n > 0 && !x)) return fail(VND_ERR_INVALID, "null signal pointer");
    HostLock lock(ctx->host_mutex);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t in_bytes = (size_t)batch * n * in_channels * sizeof(float);
    const size_t out_bytes = (size_t)batch * total * 2 * sizeof(double);
    st = ensure_work(ctx, out_bytes + std::max<size_t>(in_bytes, 16));
    if (st != VND_OK) return st;
    char *buf = ctx->work;
    if (in_bytes) HIP_TRY(hipMemcpyAsync(buf + out_bytes, x, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    st = vnd_haas_f64_dev(ctx, (const float *)(buf + out_bytes), (double *)buf, batch, n, in_channels, delay,
                          delayed_channel, ms_mode, use_width, width, ctx->stream);
    if (st != VND_OK) return st;
    HIP_TRY(hipMemcpyAsync(y, buf, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VND_OK;
}

#ifdef VND_STAMPS
// diagnostic builds only (not declared in vnd_amd.h)
int vnd_debug_read_stamps(unsigned long long *dst, int count)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(vnd::g_stamps), (size_t)count * sizeof(unsigned long long));
}
#endif

}  // extern "C"
