// vnd_host.hpp - the *_host entry points: staging buffers, the pipelined host paths (groups of streams, time pieces, page-locked buffers in place), page-locked allocation.
// (one translation unit: included by vnd_amd.hip after vnd_objects.hpp; everything static here is private to the library)
#pragma once

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

extern "C" {

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
    // Measured (tools/closed/host_pieces_try.py, profiles/r03_host_pieces.txt): every extra copy call costs ~50 us of fixed time on
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
    // The range is walked REGISTRATION BY REGISTRATION: every probe must be page-locked host memory whose device address continues
    // the first one's, and the next probe is the first byte past the extent (hipMemGetAddressRange) of the registration the last
    // one fell in - so a hole of any size between two hipHostRegister ranges is stepped ON, not over, and such a range never
    // reaches a kernel as one device pointer.  Where the runtime does not report an extent for this kind of memory the walk
    // falls back to fixed 2 MiB steps (plus the last byte) and declines ranges that would need more than 4096 of them.
    void *base = nullptr;
    size_t probes = 0;
    for (size_t off = 0;;) {
        hipPointerAttribute_t at{};
        const bool ok = hipPointerGetAttributes(&at, (const char *)p + off) == hipSuccess;
        (void)hipGetLastError();                                  // (an ordinary pageable pointer reports an error: not ours)
        if (!ok || at.type != hipMemoryTypeHost || !at.devicePointer) return false;
        if (off == 0) base = at.devicePointer;
        else if ((const char *)at.devicePointer - (const char *)base != (ptrdiff_t)off) return false;
        if (off == bytes - 1) break;
        size_t next = off + ((size_t)2 << 20);
        hipDeviceptr_t ext_base = nullptr;
        size_t ext_bytes = 0;
        if (hipMemGetAddressRange(&ext_base, &ext_bytes, (hipDeviceptr_t)at.devicePointer) == hipSuccess && ext_bytes > 0 &&
            (const char *)ext_base <= (const char *)at.devicePointer &&
            (const char *)at.devicePointer < (const char *)ext_base + ext_bytes) {
            next = off + (size_t)((const char *)ext_base + ext_bytes - (const char *)at.devicePointer);   // first byte past this registration
        } else {
            (void)hipGetLastError();
            if (++probes > 4096) return false;
        }
        off = std::min(next, bytes - 1);
    }
    *dev = base;
    return true;
}

static bool host_direct_enabled()
{
    static int slot = INT32_MIN;
    return host_env_once("VND_HOST_DIRECT", 1, &slot) != 0;
}

vnd_status vnd_host_buffers_mapped(const void *x, int64_t x_bytes, const void *y, int64_t y_bytes, int32_t *mapped)
{
    if (!mapped || x_bytes < 0 || y_bytes < 0) return fail(VND_ERR_INVALID, "bad arguments");
    // exactly the conditions of convolve_host's in-place path: the switch, two buffers that do not overlap, both mapped
    void *xd = nullptr, *yd = nullptr;
    const bool apart = !((const char *)x < (const char *)y + y_bytes && (const char *)y < (const char *)x + x_bytes);
    *mapped = host_direct_enabled() && apart && host_mapped(x, (size_t)x_bytes, &xd) && host_mapped(y, (size_t)y_bytes, &yd) ? 1 : 0;
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
    DeviceScope on(ctx->device);
    if (!on.ok) return fail(VND_ERR_HIP, "cannot select device %d", ctx->device);
    const size_t in_elems = (size_t)batch * n * Cx, out_elems = (size_t)batch * n * C;
    // Page-locked buffers on BOTH sides: the kernel works on them in place - its loads and stores cross PCIe inside the
    // launch, both directions at once, with no staging copy before or after (one 10 s stereo signal 0.147 against 0.185 ms,
    // 1024 x 1 s 9.95 against 14.1 ms: tools/closed/zero_copy_try.py).  Every frame is read once plus the halo at span seams, and
    // written once: the bytes over PCIe are the staged path's.  VND_HOST_DIRECT=0 keeps the staged path.
    // (measured and dropped, same tool: a mapped input read in place with a staged download per group - 15.1 ms for the
    //  1024 streams; a staged upload with every group written in place - 13.7 ms with page-locked, 9.8-10.1 with pageable
    //  input against the staged pipeline's 8.8: a pageable upload is staged by the CPU, beside the SDMA download.)
    const bool direct = host_direct_enabled();
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

}  // extern "C"
