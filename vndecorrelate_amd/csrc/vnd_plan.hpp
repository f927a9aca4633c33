// vnd_plan.hpp - launch geometry: which kernel takes a launch and how its work is cut (generic kernels: make_plan; per-table kernels: make_spec_plan, their modules, launch).
// (one translation unit: included by vnd_amd.hip after vnd_objects.hpp; everything static here is private to the library)
#pragma once

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

// VND_TUNING=1 sessions: a tuning variable the library read that is not in kTuningNames (vnd_spec.hpp) is a bug in the library;
// the entry point that planned the launch reports it as VND_ERR_INVALID - it never ends the host process
static vnd_status tuning_status()
{
    const char *name = spec_unregistered_name().load();
    if (name == nullptr) return VND_OK;
    return fail(VND_ERR_INVALID, "tuning variable %s is read by the library but not registered in kTuningNames", name);
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
    int bal_total = 0;              // > 0: the BALANCED cut - every workgroup a contiguous range of the pool's streams x tiles_total tiles
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
    // 4k-channel tables (quad / octet form): the store phase leaves one row of 2 C sums of squares per (tile, wave of a channel) in
    // blk_sum (rows_major); *rows = rows per stream it writes.  spec_only: launch nothing if the per-table kernel does not take it
    // (*path stays 0: the caller goes on with its unfused passes)
    int *rows = nullptr;
    int rows_max = 0;            // rows per stream the caller's workspace has room for: a plan that would write more is refused BEFORE anything runs
    bool spec_only = false;
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
    const bool pointwise2 = epi != nullptr && (!epi->normalize || epi->blk_sum != nullptr) && epi->sink == nullptr && C == 2;
    // ... and on 4k channels the normaliser's sums alone (LR mode: no pointwise step exists there), in the quad / octet form's store phase
    // (fast mode: rows of sums for the fused stage; exact mode: the per-block predictions of the block-parallel NumPy-order sums)
    const bool sums_q = epi != nullptr && epi->sink == nullptr && C % 4 == 0 && Cx == C && !epi->ms_encode && !epi->use_width &&
                        epi->blk_sum != nullptr && epi->rows != nullptr &&
                        ((mode == VND_MODE_FAST && epi->normalize && epi->rows_major) || (mode == VND_MODE_EXACT && !epi->rows_major));
    const bool pointwise = pointwise2 || sums_q;
    // fan-out: a mono input through a stereo table is in scope (one LDS plane, VS_BC); wider fan-outs are not
    const bool bc = Cx == 1 && C == 2;
    if ((epi != nullptr && !pointwise) || (Cx != C && !bc)) { p.why = "fused epilogue or fan-out launch"; return p; }
    if (!(mode == VND_MODE_EXACT ? t->spec_exact_ok : t->spec_ok)) { p.why = "table outside the specialised kernel's scope"; return p; }
    // VND_MODE_EXACT specialises by default as well: with the shifted plane copies (odd offsets as aligned pairs) the per-table
    // kernel is ahead of the generic ordered one by 24 % on a function-path table, 37 % on a class-path one and 23-50 % on a mono
    // input fanned out (cfg2 pool; tools/closed/exact_geometry_try.py, tools/closed/fanout_spec_try.py).  VND_SPEC_EXACT=0 keeps the generic kernel.
    if (mode == VND_MODE_EXACT && !(v >= 0 && ((v >> 15) & 1))) {
        static const bool off = [] { const char *e = getenv("VND_SPEC_EXACT"); return e && e[0] == '0'; }();
        if (off) { p.why = "exact mode specialisation switched off"; return p; }
    }
    const bool force = v >= 0 && ((v >> 23) & 1);
    // the fast mode's window forms in the reference's class-path association (adds inside a segment, the gain ratio once per segment:
    // vnd_win.hpp, win_adds_ok) wherever the table has few distinct |w| - every generated table has; VND_WIN_ADDS=0: one FMA per tap
    const int adds_now = (mode == VND_MODE_FAST && spec_env("VND_WIN_ADDS", 1) != 0 && win_adds_ok(t->spec_table)) ? 1 : 0;
    if (spec_disabled_by_env() || (v >= 0 && ((v >> 25) & 1))) { p.why = "disabled"; return p; }
    // access shape: 16 bytes per frame pair (stereo) or 8 per frame, from every stream's first sample
    const uintptr_t align = C == 2 ? 16 : 8, align_x = bc ? 8 : align;
    if (((uintptr_t)y & (align - 1)) || ((uintptr_t)x & (align_x - 1))) { p.why = "unaligned base"; return p; }
    if (batch > 1 && (((uint64_t)n * C * 4) % align != 0 || ((uint64_t)n * Cx * 4) % align_x != 0)) { p.why = "unaligned streams"; return p; }
    // (before any geometry is searched: below about two million frames per channel pair the generic kernels - many small workgroups -
    //  stay ahead, and a tiny launch should not pay for a plan it will discard)
    if (!force && batch * (C / 2) * n < 2000000) { p.why = "too little work for persistent workgroups"; return p; }
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
    //  tools/closed/fanout_win_try.py)
    // (wider signals - a workgroup per channel PAIR, VW_C - keep the pair-read kernel unless variant bits 5-7 or VND_WIN_WIDE=1
    //  ask for the window form: there a workgroup moves 8 bytes of every frame, the memory pipeline's time per useful byte
    //  is 2-4x a stereo signal's and the window form's few waves per CU do not hide it - cfg5 0.54 ms against 0.45, while
    //  the same tables on planar channel pairs run 0.33 against 0.39: tools/closed/c8_win_try.py, profiles/r03_cfg5_request_floor.txt)
    const int win_wide_env = spec_env("VND_WIN_WIDE", 0);
    const bool win_c = C == 2 || (C % 2 == 0 && (win_wide_env != 0 || vw >= 2));
    // signals of 4k channels: the window form on channel QUADS / OCTETS (VW_Q, vw_span_qc: a workgroup moves 16 / 32 bytes of every
    // frame, a wave per channel) - VND_WIN_QUAD=0 keeps the pair-read kernel (or, with VND_WIN_WIDE=1 / variant bits 5-7, the
    // window form on channel pairs)
    // (4k + 2 channels - 6, 10, ... - ride the quad form too: k quads and one more from channel C - 4, overlapping in one pair)
    const bool win_quad = (C % 4 == 0 || (C % 4 == 2 && C >= 6 && epi == nullptr)) && Cx == C && (!pointwise || sums_q) && spec_env("VND_WIN_QUAD", 1) != 0;
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
        c.adds = c0.win ? adds_now : 0;
        std::lock_guard<std::mutex> g(const_cast<vnd_taps *>(t)->spec_mutex);
        auto it = t->spec_modules.find(c);
        return it != t->spec_modules.end() && !it->second->building && it->second->failed;
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
    //    frames (rejected builds fall back to the plain form) (tools/closed/win_split_try.py, profiles/r03_split_waves.txt)
    // a mono input fanned out, fast mode: the plain form with ONE read stream for both output channels (win_taps_function_merged:
    // the two channels' taps lie almost alike, their windows' union is little more than one channel's - 1.48 B of LDS per FMA)
    // ... and in the exact mode for function-path tables (one ascending pass per channel: win_taps_function_exact_merged - the reads and
    // the products f32(x * |w|) shared by both channels, every sum the same operation as in a pass per channel); VND_WIN_EXACT_MERGED=0
    // keeps the split form with the input staged into both plane sets
    const bool merged_exact = bc && mode == VND_MODE_EXACT && win_exact && spec_env("VND_WIN_EXACT_MERGED", 1) != 0 && win_exact_merged_ok(t->spec_table);
    if (!picked && win_mode_ok && bc && (mode == VND_MODE_FAST || merged_exact) && vw == 0)
        picked = win_pick_config(t->spec_table, (size_t)ctx->lds_limit, win_m, attempt == 1, true, &p.cfg, rejected);
    const int split_env = spec_env("VND_WIN_SPLIT", 1);
    // (a mono input fanned out rides the same form: its one channel staged into both plane sets, VW_BC - cfg1's shape 0.177 -> 0.15 ms
    //  for 128 x 10 s against the pair-read form, tools/closed/fanout_win_try.py; VND_WIN_SPLIT_FANOUT=0 keeps that)
    const bool split_scope = C == 2 && (Cx == 2 || (bc && spec_env("VND_WIN_SPLIT_FANOUT", 1) != 0)) && !pointwise;
    //    In the FAST mode (E and P: 128 accumulator registers) the 64-frame split form needs its refill loaded late (VW_LATE: 15
    //    of a wave's 16 accesses per tile at the start of the store phase that consumes them, not a tile ahead) and the per-access
    //    constants kept out of the tile loop's registers: cfg3 +3-4.5 % (0.457 -> 0.437 ms), cfg2 +3.8 % (0.195 -> 0.188 ms,
    //    tools/split64_fast_probe.py); a table whose build spills all the same falls back to the plain 32-frame form
    const bool exact_now = mode == VND_MODE_EXACT;
    //    Class-path tables in the exact mode (a segment sum and an output sum per pair: the fast mode's register count) take it with
    //    the fast mode's late refill since round 6 (VND_WIN_SPLIT_CLASS=0: the plain 32-frame form)
    if (!picked && win_mode_ok && split_scope && split_env == 1 && vw == 0 &&
        ((exact_now && (!t->spec_table.has_seg || spec_env("VND_WIN_SPLIT_CLASS", 1) != 0)) || mode == VND_MODE_FAST))
        picked = win_pick_config(t->spec_table, (size_t)ctx->lds_limit, 64, attempt == 1, bc, &p.cfg, rejected, 0, true, exact_now);
    if (!picked && win_mode_ok && split_scope && split_env == 2)
        picked = win_pick_config(t->spec_table, (size_t)ctx->lds_limit, win_m, attempt == 1, bc, &p.cfg, rejected, 0, true, exact_now);
    // (a mono input under the decorrelate stage's pointwise steps, exact mode - VelvetNoise.decorrelate of mono signals,
    //  decorrelation.py:428-440 - rides the plain form too since round 6: both channels' passes read the one plane set, the store
    //  phase has the mono frame for the side-channel encode and leaves the block sums of the exact RMS; until then the pair-read form
    //  plus a pass of its own for those sums: 128 x 10 s 0.916 -> 0.655 ms, bit-identical; pools of more than 320 streams, whose sums
    //  run per stream and take no block sums, stay 2 % ahead in the pair-read form and keep it: profiles/r06_f1_mono.txt)
    if (!picked && win_mode_ok && win_c && (!bc || vw >= 2 || (pointwise2 && epi->blk_sum != nullptr && mode == VND_MODE_EXACT && spec_env("VND_WIN_FANOUT_EPI", 1) != 0)))
        picked = win_pick_config(t->spec_table, (size_t)ctx->lds_limit, win_m, attempt == 1, bc, &p.cfg, rejected);
    if (sums_q && !(picked && p.cfg.win_q)) { p.why = "the sums of a 4k-channel table ride in the quad / octet form only"; return p; }
    if (!picked && !spec_pick_config(t->spec_table, (size_t)ctx->lds_limit, rr_hint, dd_hint, &p.cfg, attempt == 1 || C != 2, bc, mode == VND_MODE_EXACT)) { p.why = "halo does not fit the ring"; return p; }
    const int64_t T = p.cfg.tile();
    const int64_t tiles_total = (n + T - 1) / T;
    const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
    // resident workgroups per CU: LDS-bound, within the 32 waves a CU holds
    const int64_t per_cu = p.cfg.win ? p.cfg.win_per_cu
                                     : std::min<int64_t>(std::min<int64_t>(16, 2048 / p.cfg.nt), (int64_t)(160 * 1024) / (int64_t)p.cfg.lds_bytes());
    const int64_t resident = (int64_t)cus * std::max<int64_t>(per_cu, 1);
    const int64_t units = batch * (p.cfg.win_q ? (C + 4 * p.cfg.win_q - 1) / (4 * p.cfg.win_q) : C / 2);      // (stream, channel pair) - or channel quad / octet
    // a workgroup needs a span long enough to amortise filling its ring: 8 tiles when there are 16 and more per resident
    // slot; with less, shorter spans (down to 2 tiles) so that the chip still fills - a lone 60 s stream then runs 1.1x
    // (fast) to 1.75x (exact, 128 taps) faster than through the generic kernels, tools/closed/single_stream_try.py - and below
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
    // the fused fast stage of a 4k-channel table: the store phase writes a row of sums per (tile, wave of a channel) into the caller's
    // workspace - a geometry with more rows than that has room for (several waves per channel on a short signal) is not taken: the
    // caller's unfused passes run instead (spec_only), nothing is written past the rows
    if (sums_q && epi->rows_major && epi->rows_max > 0 &&
        tiles_total * std::max<int64_t>(1, p.cfg.nt / 64 / (4 * std::max(1, p.cfg.win_q))) > (int64_t)epi->rows_max) {
        p.use = false;
        p.why = "the store phase's rows of sums do not fit the caller's workspace";
        return p;
    }
    p.cfg.nt_stores = nt_stores_of(p.cfg);
    p.cfg.exact = mode == VND_MODE_EXACT ? 1 : 0;
    p.cfg.epi = pointwise ? 1 : 0;
    p.cfg.bc = bc ? 1 : 0;
    p.cfg.adds = p.cfg.win ? adds_now : 0;
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
            p.stagger_ticks = std::max(0, spec_env("VND_WIN_STAGGER_TICKS", 300));      // 3 us: about the first workgroup's ring fill - the later one loads while that one computes (tools/ablate/RUNS.md: run_r4b, run_r4c.sh)
            p.units = (uint32_t)(2 * cus);
            p.nblocks = p.units;
        }
    }
    // ---- spans that fill the one round unevenly: the BALANCED cut ------------------------------------------------------
    // 192 ten-second streams are 960 spans of 12 tiles on 512 resident workgroups: most walk two, the launch takes 24 tile periods for
    // 22.1 tiles of work per workgroup (0.608 of 8 TB/s where 128 and 256 streams run 0.66-0.67, profiles/r06_pool_sweep.txt).  Instead
    // every workgroup takes a contiguous range of the pool's tiles, the same number (+- 1); a range that crosses into the next stream
    // starts a new ring there.  Taken when a two-line cost model (a ring fill = 0.55 tile periods) puts it 3 % ahead of the spans.
    p.bal_total = 0;
    if (p.cfg.win && C == 2 && p.chunk_tiles == 0 && !(v >= 0 && ((v >> 28) & 7)) && !(v >= 0 && ((v >> 20) & 7)) && spec_env("VND_WIN_BALANCE", 1) != 0) {
        const int64_t total = batch * tiles_total;
        const int64_t wgs = std::min<int64_t>(resident, total / 4);                      // (at least 4 tiles per workgroup)
        if (wgs >= 1 && total < 0x7fffffffLL) {
            const int64_t per_wg = (total + wgs - 1) / wgs;
            const int64_t uni_units = (units * spans + p.nblocks - 1) / p.nblocks;         // units the busiest workgroup walks
            const double cost_spans = (double)uni_units * ((double)per_span + 0.55);
            const double cost_bal = (double)per_wg + 0.55 * (1.0 + (double)per_wg / (double)tiles_total);
            if (per_wg <= max_tiles && (spec_env("VND_WIN_BALANCE", 1) == 2 || cost_bal < 0.97 * cost_spans)) {      // (a range stays under 2 GiB of descriptor offsets)
                p.bal_total = (int)total;
                p.nblocks = (uint32_t)wgs; p.units = (uint32_t)wgs;
                p.tiles_per_span = (int)per_wg; p.spans = 0;                           // (what the description and the pacing rule read)
            }
        }
    }
    p.use = true;
    // (window form: 8192-frame tiles down to 3 per span - 256 one-second streams 42.6 us with them, 45.7 with 4096-frame
    //  tiles; at 2 per span - 128 such streams - the smaller tiles win, 26.3 against 28.1 us: tools/closed/shard_try.py)
    if (per_span >= (p.cfg.win ? 3 : 12) || rr_hint > 0 || p.chunk_tiles > 0) break;
    }
    return p;
}

// compiled on first use, once per (table, geometry); a failed build is remembered and the generic
// kernel takes over (the reason stays readable through vnd_describe_launch)
// cache_only: a launch too small to be worth a 1.5-5 s build takes the per-table kernel only when its code object is
// already there - in this table's map or in the disk cache (looked up once) - and the generic kernel otherwise
// spec_compile builds strings and vectors: an exception in it (std::bad_alloc) must neither cross the C ABI nor leave the entry
// `building` for ever (every later launch of that geometry would wait on spec_built) - the entry is published as failed instead
static void spec_compile_guarded(const SpecTable &table, const SpecConfig &cfg, int device, int lds_limit, SpecModule *m, bool cache_only)
{
    try {
        spec_compile(table, cfg, device, lds_limit, m, cache_only);
    } catch (const std::exception &e) {
        m->failed = true; m->pending = false;
        try { m->log = std::string("exception while building: ") + e.what(); } catch (...) {}
    } catch (...) {
        m->failed = true; m->pending = false;
    }
}

static SpecModule *spec_module(vnd_ctx *ctx, const vnd_taps *t_, const SpecConfig &cfg, bool cache_only = false)
{
    // The table's mutex guards the MAP, not the 1.5-5 s of hipRTC: the thread that finds no entry inserts one marked `building`,
    // compiles outside the lock and publishes the result; meanwhile a launch that would not have built anyway (cache_only: a small
    // launch on the host's hot thread while another thread runs vnd_prepare_launch) sees "not there yet" and takes the generic
    // kernel, and one that needs this very kernel waits for the builder instead of compiling it a second time.
    vnd_taps *t = const_cast<vnd_taps *>(t_);
    std::unique_lock<std::mutex> g(t->spec_mutex);
    for (;;) {
        auto it = t->spec_modules.find(cfg);
        if (it == t->spec_modules.end()) {
            std::unique_ptr<SpecModule> fresh(new SpecModule);
            fresh->building = true;
            SpecModule *m = fresh.get();
            t->spec_modules.emplace(cfg, std::move(fresh));
            g.unlock();
            spec_compile_guarded(t->spec_table, cfg, ctx->device, ctx->lds_limit, m, cache_only);
            g.lock();
            m->building = false;
            t->spec_built.notify_all();
            return m->pending ? nullptr : m;
        }
        SpecModule *m = it->second.get();
        if (m->building) {
            if (cache_only) return nullptr;
            t->spec_built.wait(g, [m] { return !m->building; });
            continue;
        }
        if (m->pending && !cache_only) {                                  // looked for in the disk cache only, before: build it now
            m->building = true;
            g.unlock();
            spec_compile_guarded(t->spec_table, cfg, ctx->device, ctx->lds_limit, m, false);
            g.lock();
            m->building = false;
            t->spec_built.notify_all();
        }
        return m->pending ? nullptr : m;
    }
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
    a.chunk_tiles = p.chunk_tiles; a.chunk_len0 = p.chunk_len0; a.chunks_per_stream = p.chunks_per_stream;
    a.cus_per_xcd = std::max(1, (ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256) / 8);      // (pacing needs it without a chunk plan too)
    // pacing: one full round of workgroups, two per CU (their co-residency lasts the whole launch), plain stereo forms
    if (p.cfg.win && !p.cfg.win_q && p.chunk_tiles == 0 && p.cfg.win_per_cu == 2 && spec_env("VND_WIN_PACE", 1) != 0 &&
        p.nblocks > (uint32_t)(8 * a.cus_per_xcd) && p.nblocks <= (uint32_t)(2 * 8 * a.cus_per_xcd) && p.units >= p.nblocks &&
        // (long launches only: with a few tiles per workgroup the bias it corrects has no time to build up, and handing the later
        //  workgroup the priority costs - cfg4's N = 4 shard, 3 tiles each: 42.8 -> 48.8 us; cfg3's 17 tiles: +4.7 %)
        (int64_t)p.units * p.tiles_per_span >= (int64_t)p.nblocks * spec_env("VND_WIN_PACE_MIN_TILES", 16)) {
        a.pace = ctx->pace;                                  // made and zeroed by vnd_ctx_create: a *_dev launch only enqueues (null: no pacing)
    }
    a.stagger_ticks = p.stagger_ticks; a.chunk_prio = 1; a.bal_total = p.bal_total;
    if (epi != nullptr && p.cfg.epi) {
        a.epi_ms_encode = epi->ms_encode; a.epi_use_width = epi->use_width; a.epi_w_mid = epi->w_mid; a.epi_w_side = epi->w_side;
        // a wave of the plain 32-frame form owns one 2048-frame block of the sums (kParFrames)
        if (epi->blk_sum != nullptr && p.cfg.win == 32 && !p.cfg.win_s && !p.cfg.win_q && p.cfg.tile() % kParFrames == 0) {
            a.epi_blk_sum = epi->blk_sum; a.epi_nblocks = epi->nblocks; a.epi_rows_major = epi->rows_major;
        }
        // quads / octets: a row per (tile, wave of a channel) - the plan's tiles x the waves a channel has
        if (epi->blk_sum != nullptr && p.cfg.win_q && epi->rows_major && epi->rows != nullptr) {
            a.epi_blk_sum = epi->blk_sum; a.epi_rows_major = 1;
            a.epi_nblocks = p.tiles_total * std::max(1, p.cfg.nt / 64 / (4 * p.cfg.win_q));
            *epi->rows = a.epi_nblocks;
        }
        // ... or, exact mode, the block sums of rms_par_*: only when a channel wave's tile is exactly one of their 2048-frame blocks
        if (epi->blk_sum != nullptr && p.cfg.win_q && !epi->rows_major && epi->rows != nullptr &&
            p.cfg.tile() == kParFrames && p.cfg.nt / 64 == 4 * p.cfg.win_q && p.tiles_total == epi->nblocks) {
            a.epi_blk_sum = epi->blk_sum; a.epi_rows_major = 0; a.epi_nblocks = epi->nblocks;
            *epi->rows = a.epi_nblocks;
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
        if (vnd_status ts = tuning_status(); ts != VND_OK) return ts;
        if (!sp.use) break;
        bool launched = false, built = true;
        vnd_status st = launch_spec(ctx, t, sp, x, y, n, stream, &launched, epi, &built);
        if (st != VND_OK || launched) return st;
        if (!sp.cfg.win || built) break;                          // (a failed window build: plan again, that geometry is skipped now)
    }
    if (epi != nullptr && epi->spec_only) return VND_OK;          // (*epi->path is 0: nothing was launched)
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
