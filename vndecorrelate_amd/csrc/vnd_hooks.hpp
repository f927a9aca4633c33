// vnd_hooks.hpp - measurement, tuning and diagnosis hooks (include/vnd_amd_internal.h): timing loops, the streaming-copy ceiling, the generated kernel sources, the variant override, phase stamps.
// (one translation unit: included by vnd_amd.hip after vnd_objects.hpp; everything static here is private to the library)
#pragma once

extern "C" {

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
    quad = quad || ((C % 4 == 0 || (C % 4 == 2 && C >= 6)) && spec_env("VND_WIN_QUAD", 1) != 0 &&
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
    cfg.adds = (!cfg.exact && spec_env("VND_WIN_ADDS", 1) != 0 && win_adds_ok(t)) ? 1 : 0;      // (as make_spec_plan)
    // (VND_WIN_SOURCE_EPI=1: with the decorrelate stage's steps in the store phase - stereo: pointwise steps and block sums; quads /
    //  octets, fast mode: the normaliser's sums)
    cfg.epi = (spec_env("VND_WIN_SOURCE_EPI", 0) != 0 && !bc && !split && (C == 2 || quad)) ? 1 : 0;
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

vnd_status vnd_tuning_read(const char *name, int32_t fallback, int32_t *value)
{
    if (!name || !value) return fail(VND_ERR_INVALID, "null name or value pointer");
    // (the caller's pointer never reaches the process-wide record of spec_env: that one holds pointers into the library's own
    //  literals only, and another thread's launch may be reading it - or have just written its own finding - right now)
    const char *known = nullptr;
    for (const char *k : kTuningNames) if (strcmp(k, name) == 0) known = k;
    if (!known) { *value = fallback; return fail(VND_ERR_INVALID, "'%.64s' is not a registered tuning variable (kTuningNames, csrc/vnd_spec.hpp)", name); }
    *value = spec_env(known, fallback);
    return VND_OK;
}

vnd_status vnd_set_variant(vnd_ctx *ctx, int32_t variant)
{
    if (!ctx) return fail(VND_ERR_INVALID, "null context");
    ctx->variant = variant;
    ctx->variant_nofuse = (variant >= 0 && ((variant >> 24) & 1)) ? 1 : 0;   // bit 24: unfused epilogue
    return VND_OK;
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

}  // extern "C"
