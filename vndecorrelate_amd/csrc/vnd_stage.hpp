// vnd_stage.hpp - the rows either side of the convolution: the decorrelate stage (pointwise steps, exact / fused RMS normaliser), promoted operand types, the optimiser's candidate scan, HaasEffect.
// (one translation unit: included by vnd_amd.hip after vnd_objects.hpp; everything static here is private to the library)
#pragma once

extern "C" {

static int64_t epi_chunks(int64_t n) { return (n + kEpiChunk - 1) / kEpiChunk; }

// rows of partial sums per stream: pass-1 chunks, or - fused - one row per tile (>= 512 frames each)
static int64_t epi_rows_max(int64_t n) { return std::max<int64_t>(epi_chunks(n), (n + 511) / 512 + 1); }

static int64_t par_blocks(int64_t n) { return std::max<int64_t>((n + kParFrames - 1) / kParFrames, 1); }

static int64_t pw_chunks(int64_t n) { return std::max<int64_t>((n + kPwChunk - 1) / kPwChunk, 1); }

vnd_status vnd_decorrelate_workspace_bytes(int64_t batch, int64_t n, int32_t C, int64_t *bytes)
{
    if (!bytes || batch < 0 || n < 0 || C <= 0) return fail(VND_ERR_INVALID, "bad workspace query");
    *bytes = batch * epi_rows_max(n) * 2 * C * (int64_t)sizeof(double) + batch * C * (int64_t)sizeof(float) + 16;
    // the parallel exact sums (stereo tables, and channel pair by channel pair for wider ones): per stream, pair and chain, a float64
    // sum and a record per block
    if (C % 2 == 0) *bytes += 32 + batch * (C / 2) * 4 * (par_blocks(n) * (int64_t)(sizeof(double) + sizeof(ParRec) + sizeof(ParGrp)) + (int64_t)sizeof(float));
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
    // (wide == 2, the block-parallel sums' staging: a wave's 16-byte accesses on consecutive bytes instead of 32 consecutive bytes per
    //  lane as two accesses - interleaved A/B at 128 / 64 / 32 streams: stereo +1 ... 3 %, mono +3 ... 10 %, the quads of a wider signal
    //  441 -> 348 us (tools/stage_coalesced_ab.py).  The per-stream kernel of pools of 256 streams and more keeps its own mapping:
    //  it measured slower with this one - see epilogue_rms_seq_kernel)
    const int wide_par = e.wide ? (spec_env("VND_EPI_PAR_COALESCED", 1) != 0 ? 2 : 1) : 0;
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
    //   more than 320: it fills the chip by itself and reads the data once instead of twice (1024: 4.43 / 5.0; the switch was at 256 until round 6).
    // variant bit 19 keeps the per-stream kernel, bit 17 forces the block-parallel form (A/B runs).
    // Wider signals (round 5): the same kernels channel pair by channel pair - a "stream" of theirs is one pair of a stream (RArgs::pairs),
    // 8 bytes of every frame.  The per-stream kernel takes 16 workgroups for cfg5's pool of 16 signals (8.7 ms for the stage); the
    // block-parallel form 16 x 4 pairs x 469 blocks.
    const int pairs = C / 2;
    const bool par_ok = want_seq && C % 2 == 0 && (C == 2 || Cx == C) && par_blocks(n) <= kParMaxBlocks && !(ctx->variant >= 0 && ((ctx->variant >> 19) & 1));
    const bool par_forced = ctx->variant >= 0 && ((ctx->variant >> 17) & 1);
    RArgs r{};
    int conv_path = 0;                                     // EpiFuse::path of the convolution launch
    if (par_ok) {
        r.x = x; r.y = y; r.n = n; r.Cx = Cx; r.nblocks = (int32_t)par_blocks(n); r.C = C; r.pairs = pairs;
        char *extra = (char *)((float *)((double *)workspace + batch * epi_rows_max(n) * 2 * C) + batch * C);
        extra += (16 - ((uintptr_t)extra & 15)) & 15;
        r.blk_sum = (double *)extra;
        r.rec = (ParRec *)(r.blk_sum + batch * pairs * 4 * (int64_t)r.nblocks);
        r.grp = (ParGrp *)(r.rec + batch * pairs * 4 * (int64_t)r.nblocks);
        r.first = (float *)(r.grp + batch * pairs * 4 * (int64_t)r.nblocks);
        r.partials = (double *)workspace;
        r.prefixed = r.nblocks > kParPrefixBlocks ? 1 : 0;
        r.wide = wide_par;
    }
    // (round 6, with the tally's staging on consecutive bytes: 256 streams 1.27-1.37 -> 1.22-1.25 ms block-parallel; from 384 on the per-stream
    //  kernel leads, 1.70 against 1.80-1.84 - tools/f1_threshold_try.py, profiles/r06_f1_threshold.txt)
    const bool want_blk = par_ok && C == 2 && (batch <= 320 || par_forced) && spec_env("VND_EPI_BLOCK_SUMS", 1) != 0;
    bool sums_pending = false;                             // the sequential sums still have to run
    // 4k channels, fast mode, the normaliser alone (LR mode - cfg5 through the class API, decorrelation.py:433-440): the quad / octet
    // kernel's store phase leaves the sums of squares on its way (x still in the ring, y in registers), one streaming pass scales:
    // 16 bytes per sample instead of 24.  Where that kernel does not take the launch nothing has run and the passes below do.
    bool q_done = false, q_blk_done = false;
    if (mode == VND_MODE_FAST && normalize && !want_seq && !ms_encode && !use_width && C % 4 == 0 && Cx == C &&
        ctx->variant_nofuse == 0 && spec_env("VND_EPI_BLOCK_SUMS", 1) != 0) {
        int rows_q = 0;
        EpiFuse f{(double *)workspace, 0, 0, e.normalize, e.w_mid, e.w_side};
        f.path = &conv_path; f.blk_sum = (double *)workspace; f.rows_major = 1; f.rows = &rows_q; f.rows_max = (int)std::min<int64_t>(epi_rows_max(n), INT32_MAX); f.spec_only = true;
        st = launch(ctx, t, x, y, batch, n, C, mode, stream, &f, Cx);
        if (st != VND_OK) return st;
        if (conv_path == 1 && rows_q > 0 && rows_q <= epi_rows_max(n)) { e.rows = rows_q; q_done = true; }
        else if (conv_path != 0) return fail(VND_ERR_HIP, "the quad / octet kernel ran without its sums");
    }
    if (q_done) {
        // (nothing more before the reduce and scale passes)
    } else if (fused) {
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
        // (LR mode with the normaliser alone - no pointwise step - takes the same launch for the block sums its store phase leaves: the
        //  sums' own pass over x and y is a fifth of the stage at 128 streams)
        const bool sums_only = !pointwise && want_blk && normalize && spec_env("VND_EPI_SUMS_ONLY", 1) != 0;
        const bool in_kernel = (pointwise || sums_only) && mode != VND_MODE_FAST && ctx->variant_nofuse == 0 &&
                               ordered_epi_kernel(p, arithmetic_of(t, mode)) != nullptr;
        if (in_kernel) {
            EpiFuse f{nullptr, e.ms_encode, e.use_width, 0, e.w_mid, e.w_side};
            f.path = &conv_path;
            if (want_blk) { f.blk_sum = r.blk_sum; f.nblocks = r.nblocks; }
            st = launch(ctx, t, x, y, batch, n, C, mode, stream, &f, Cx);
            e.ms_encode = e.use_width = 0;                 // done
        } else {
            // 4k channels, exact mode: the quad / octet kernel's store phase leaves the per-block sums of squares the block-parallel
            // NumPy-order sums start from (as the stereo window form does) - when that kernel takes the launch
            bool launched_q = false;
            if (par_ok && C > 2 && C % 4 == 0 && Cx == C && normalize && mode == VND_MODE_EXACT && batch < 256 &&
                ctx->variant_nofuse == 0 && spec_env("VND_EPI_BLOCK_SUMS", 1) != 0) {
                int rows_q = 0;
                EpiFuse f{nullptr, 0, 0, 0, e.w_mid, e.w_side};
                f.path = &conv_path; f.blk_sum = r.blk_sum; f.nblocks = r.nblocks; f.rows = &rows_q; f.spec_only = true;
                st = launch(ctx, t, x, y, batch, n, C, mode, stream, &f, Cx);
                if (st != VND_OK) return st;
                launched_q = conv_path != 0;
                q_blk_done = conv_path == 1 && rows_q == r.nblocks;
            }
            if (!launched_q) st = launch(ctx, t, x, y, batch, n, C, mode, stream, nullptr, Cx);
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
    const bool blk_done = (conv_path == 1 && want_blk) || q_blk_done;
    // (wider signals: the per-stream kernel fills the chip from 256 streams on, as for stereo; below that the pairs' blocks do)
    const bool par_sums = sums_pending && par_ok && (batch <= 64 || par_forced || blk_done || (C > 2 && batch < 256));
    if (par_sums) {
        e.rows = 1;
        e.exact_rms = 1;
        e.normalize = 1;
        // (wider signals: the pairs of a block side by side in one XCD's queue - par_unit)
        // 4k channels with 16-byte-aligned streams: a channel QUAD per workgroup (512 threads, whole 16-byte accesses)
        const int pw = (pairs > 1 && pairs % 2 == 0 && r.wide) ? 2 : 1;
        const unsigned gx = pairs == 1 ? (unsigned)r.nblocks : (unsigned)(((r.nblocks + 7) / 8) * 8 * (pairs / pw));
        const dim3 pgrid(gx, (unsigned)batch), tgrid(gx, (unsigned)batch);     // tally: blocks 1.., plus block 0's chain
        const dim3 sgrid((unsigned)(batch * pairs * 4));
        if (Cx == 1) {
            if (!blk_done) hipLaunchKernelGGL(rms_par_sum_kernel<true>, pgrid, dim3(kParThreads), 0, stream, r);
            if (r.prefixed) hipLaunchKernelGGL(rms_par_prefix_kernel, dim3((unsigned)(batch * 4)), dim3(kParThreads), 0, stream, r);
            hipLaunchKernelGGL(rms_par_tally_kernel<true>, tgrid, dim3(kParThreads), 0, stream, r);
            hipLaunchKernelGGL(rms_par_stitch_kernel<true>, sgrid, dim3(64), 0, stream, r);
        } else if (pw == 2) {
            if (!blk_done) hipLaunchKernelGGL((rms_par_sum_kernel<false, 2>), pgrid, dim3(2 * kParThreads), 0, stream, r);
            if (r.prefixed) hipLaunchKernelGGL(rms_par_prefix_kernel, dim3((unsigned)(batch * pairs * 4)), dim3(kParThreads), 0, stream, r);
            hipLaunchKernelGGL((rms_par_tally_kernel<false, 2>), tgrid, dim3(2 * kParThreads), 0, stream, r);
            hipLaunchKernelGGL(rms_par_stitch_kernel<false>, sgrid, dim3(64), 0, stream, r);
        } else {
            if (!blk_done) hipLaunchKernelGGL(rms_par_sum_kernel<false>, pgrid, dim3(kParThreads), 0, stream, r);
            if (r.prefixed) hipLaunchKernelGGL(rms_par_prefix_kernel, dim3((unsigned)(batch * pairs * 4)), dim3(kParThreads), 0, stream, r);
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
        // stereo, fewer streams than two per CU: a workgroup per ARRAY of a stream (x's chains, y's chains) - twice the loads in flight
        const int cus = ctx->prop.multiProcessorCount > 0 ? ctx->prop.multiProcessorCount : 256;
        e.seq_split = (C == 2 && batch < 2 * (int64_t)cus && spec_env("VND_EPI_SEQ_SPLIT", 1) != 0) ? 1 : 0;
        hipLaunchKernelGGL(k, dim3((unsigned)(e.seq_split ? 2 * batch : batch)), dim3(64 * waves), lds, stream, e);
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
    DeviceScope on(ctx->device);
    if (!on.ok) return fail(VND_ERR_HIP, "cannot select device %d", ctx->device);
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
    DeviceScope on(ctx->device);
    if (!on.ok) return fail(VND_ERR_HIP, "cannot select device %d", ctx->device);
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
    DeviceScope on(ctx->device);
    if (!on.ok) return fail(VND_ERR_HIP, "cannot select device %d", ctx->device);
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
    DeviceScope on(ctx->device);
    if (!on.ok) return fail(VND_ERR_HIP, "cannot select device %d", ctx->device);
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

}  // extern "C"
