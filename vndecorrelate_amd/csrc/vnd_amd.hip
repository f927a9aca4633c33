// vnd_amd.hip - C ABI (include/vnd_amd.h) over the gfx950 kernels: the one translation unit of libvnd_amd.so.
// Host code only decides launch geometry and orchestrates; all arithmetic lives in the kernels (vnd_kernels.hpp, vnd_win_kernel.inc,
// vnd_spec_kernel.inc, vnd_epilogue.hpp, vnd_moments.hpp, vnd_haas.hpp).  Parts:
//   vnd_objects.hpp  context, tap table, error channel            vnd_plan.hpp   which kernel, how the work is cut, launch
//   vnd_host.hpp     *_host entry points (pipelined staging)       vnd_stage.hpp  decorrelate stage, promoted operands, scan, Haas
//   vnd_rccl.hpp     shard ranges, the tap table over RCCL         vnd_hooks.hpp  measurement / tuning / diagnosis hooks
#include "vnd_objects.hpp"
#include "vnd_plan.hpp"

// ------------------------------------------------------------------------------
// ABI: library, device, tap tables, launch introspection
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
    DeviceScope on(device);                       // (the caller's current device is the caller's: restored on every way out)
    if (!on.ok || hipGetDeviceProperties(&c->prop, device) != hipSuccess) {
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
    // the window kernel's pacing slots (16 KB): here, so that no launch allocates, touches the null stream or breaks a capture.
    // Zeroed on the context's own stream and waited for; a failure only switches pacing off.
    if (hipMalloc((void **)&c->pace, 2048 * 2 * sizeof(unsigned)) == hipSuccess) {
        if (hipMemsetAsync(c->pace, 0, 2048 * 2 * sizeof(unsigned), c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) {
            (void)hipFree(c->pace);
            c->pace = nullptr;
        }
    } else {
        c->pace = nullptr;
    }
    (void)hipGetLastError();
    *out = c;
    return VND_OK;
}

vnd_status vnd_ctx_destroy(vnd_ctx *c)
{
    if (!c) return VND_OK;
    DeviceScope on(c->device);
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
    DeviceScope on(ctx->device);
    hipError_t e = on.ok ? hipSuccess : hipErrorInvalidDevice;
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
        // 1.93 against 1.40 and 2.03 against 1.74: tools/closed/win_exact_try.py, profiles/r03_exact_window.txt)
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
    DeviceScope on(t->ctx->device);
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

}  // extern "C"

#include "vnd_host.hpp"
#include "vnd_rccl.hpp"

extern "C" {

static vnd_status describe(vnd_ctx *ctx, const vnd_taps *t, int64_t batch, int64_t n, int32_t Cx, int32_t C,
                           int32_t mode, char *text, int32_t len)
{
    vnd_status st = check_shape(ctx, t, batch, n, C, mode, Cx);
    if (st != VND_OK) return st;
    if (!text || len <= 0) return fail(VND_ERR_INVALID, "null text buffer");
    // the pointers only decide alignment: describe the launch of 256-byte-aligned buffers (hipMalloc's)
    for (int attempt = 0; attempt < 8; ++attempt) {
        const SpecPlan sp = make_spec_plan(ctx, t, nullptr, nullptr, batch, n, C, Cx, mode, nullptr);
        if (vnd_status ts = tuning_status(); ts != VND_OK) return ts;
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
                else if (sp.bal_total > 0) snprintf(split, sizeof split, "balanced ranges of %d tiles, %d in the pool", sp.tiles_per_span, sp.bal_total);
                else snprintf(split, sizeof split, "%d spans x %d tiles", sp.spans, sp.tiles_per_span);
                snprintf(text, (size_t)len,
                         "conv_spec%s_window (hipRTC, per table) frames_per_lane=%d tile=%d reads_ahead=%d "
                         "nt_stores=%d mode=%d lds=%zuB workgroups=%u (%u units: %s per stream) threads=%d store_phase=%s%s",
                         sp.cfg.exact ? "_exact" : "", sp.cfg.win, sp.cfg.tile(), sp.cfg.la, sp.cfg.nt_stores, mode,
                         sp.cfg.lds_bytes(), sp.nblocks, sp.units, split, sp.cfg.nt,
                         sp.cfg.win_s ? "planar waves=split-by-channel" : sp.cfg.win_q == 2 ? "planar pieces=channel-octets waves=split-by-channel" : (sp.cfg.win_q ? "planar pieces=channel-quads waves=split-by-channel" : (sp.cfg.win_xpose ? "frame-pairs" : "planar")),
                         sp.cfg.adds ? " taps=adds-per-segment" : "");
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
        if (vnd_status ts = tuning_status(); ts != VND_OK) return ts;
        if (!sp.use) return VND_OK;
        SpecModule *m = spec_module(ctx, t, sp.cfg, false);
        if (m && !m->failed) return VND_OK;
        if (!sp.cfg.win) break;
    }
    return VND_OK;                                                 // the generic kernels take such launches
}

}  // extern "C"

#include "vnd_stage.hpp"
#include "vnd_hooks.hpp"
