/* vnd_amd.h - C ABI of the MI355X (gfx950) velvet-noise decorrelator.
 *
 * The reference (ckonst/VNDecorrelate v1.1.0) is pure Python and has no FFI:
 * its boundary for this path is the Python call surface.  Each entry point
 * below names the reference interface it stands behind (file:line relative to
 * the reference checkout).  The Python host layer (vndecorrelate_amd/) keeps
 * the reference's names, arguments and exceptions and calls these through
 * ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - plain C types only; no C++ exceptions cross the ABI; every call returns a
 *    vnd_status (0 = ok) and vnd_last_error() gives the text for this thread;
 *  - signals are frame-interleaved, C-contiguous float32: (n_frames, C), or
 *    (batch, n_frames, C) for independent streams - the reference's layout
 *    (decorrelation.py:647), so the host never transposes;
 *  - "*_dev" functions take DEVICE pointers (hipMalloc / torch) and a
 *    hipStream_t passed as void*; they enqueue and return (no sync).  "*_host"
 *    functions take host pointers, copy H2D/D2H around the same kernels and
 *    return after the result is in `y` - the reference's synchronous semantics;
 *  - the caller owns every buffer it passes; handles own device memory;
 *  - threading: the reference's functions are pure and re-entrant
 *    (decorrelation.py:630-660), and so are these.  A context owns one stream and
 *    one set of staging buffers for its "*_host" entry points, which therefore hold
 *    a per-context mutex from entry to return: concurrent host calls on one
 *    context are serialised, never interleaved (use one context per thread to
 *    overlap them).  "*_dev" calls only enqueue on the caller's stream and may run
 *    concurrently.  A tap table is immutable and may be shared by threads; it must
 *    outlive every call that uses it and must live on the context's device.
 *    vnd_set_variant (tuning, vnd_amd_internal.h) is not synchronised.
 */
#ifndef VND_AMD_H
#define VND_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VND_ABI_VERSION 2          /* 2: measurement / tuning hooks moved to vnd_amd_internal.h */
#define VND_TAPS_IMAGE_VERSION 1   /* format of vnd_taps_serialize images (unchanged since ABI 1) */

typedef enum vnd_status {
    VND_OK = 0,
    VND_ERR_INVALID = 1,      /* bad argument (host layer maps to ValueError)      */
    VND_ERR_NO_DEVICE = 2,    /* no gfx950 device / HIP runtime failed to start   */
    VND_ERR_HIP = 3,          /* a HIP call failed; see vnd_last_error()          */
    VND_ERR_UNSUPPORTED = 4,  /* shape outside what the kernels cover             */
    VND_ERR_NOMEM = 5
} vnd_status;

/* Arithmetic of the tap sum (argument `mode` of the convolve calls). */
typedef enum vnd_mode {
    VND_MODE_EXACT = 0,  /* acc = f32(acc + f32(x*w)), taps in table order: bit-identical
                            to the reference's NumPy paths (decorrelation.py:656-658, :405-414) */
    VND_MODE_FMA = 1,    /* acc = fma(x, w, acc) in table order: one rounding per tap       */
    VND_MODE_FAST = 2    /* the throughput mode: free summation order - per output two chains (even / odd
                            offsets), far taps first; the per-table kernels add and subtract inside a run of
                            equal |w| and apply the gain once per run, the reference's own class-path
                            association (decorrelation.py:402-414).  Distance from the reference (= from
                            VND_MODE_EXACT), as a fraction of the output peak, MEASURED on uniform random
                            input (profiles/r06_k128_unseeded.json): 30 taps 3.1-4.3e-7, 64 taps 5.1-6.9e-7
                            - below 1e-6 with a margin of 1.5x and more; 128 taps 6.1-9.0e-7 of a pool's
                            peak over 64 unseeded pools of 69 M frames (none above 1e-6; 1.05e-6 of one
                            stream's own peak once).  That distance is the reference's own float32
                            rounding noise - no summation order lands closer - so it is not bounded by
                            construction and grows with the tap count: VND_MODE_EXACT is the guaranteed mode */
} vnd_mode;

/* One per (process, device).  Every call - vnd_ctx_create and the *_host entry points included -
 * runs on the context's device and restores the caller's current HIP device before it returns.
 * Device pointers and streams passed in must belong to the context's device.            */
typedef struct vnd_ctx vnd_ctx;
typedef struct vnd_taps vnd_taps;   /* device-resident tap table, immutable            */

/* ---- library / device ---------------------------------------------------- */
int vnd_abi_version(void);
const char *vnd_last_error(void);
vnd_status vnd_device_count(int32_t *count);
vnd_status vnd_ctx_create(int32_t device, vnd_ctx **ctx);
vnd_status vnd_ctx_destroy(vnd_ctx *ctx);
/* name (<= len-1 chars), compute units, bytes of HBM, LDS bytes per workgroup */
vnd_status vnd_ctx_info(const vnd_ctx *ctx, char *name, int32_t len, int32_t *compute_units,
                        int64_t *hbm_bytes, int32_t *lds_bytes);

/* ---- tap tables ------------------------------------------------------------
 * CSR over output channels: taps of channel c are [tap_offsets[c], tap_offsets[c+1])
 * of tap_index / tap_weight, consumed IN TABLE ORDER.
 *
 * Function-path table (replaces the np.where scan of decorrelation.py:651-654):
 *   seg_offsets = seg_end = seg_gain = NULL; indices ascending.
 * Class-path table (replaces _ParallelVelvetNoise, decorrelation.py:240-323,
 * consumed by VelvetNoise.convolve :402-414): per channel a CSR of segments,
 * seg_end[s] = exclusive end (absolute position in tap_index) of segment s,
 * seg_gain[s] = segment_envelope[s]; weights are -1 (negatives first) / +1;
 * apply_gain = 0 reproduces the skipped multiply of the identity envelope (:411).
 * chan_flags[c] & 1 marks an unfiltered channel that is copied through (:399-400);
 * may be NULL.  All indices must be >= 0 (and <= 2^30).  Arrays are HOST pointers, copied.
 * A table with a non-finite weight keeps the reference's semantics (a term whose tap
 * reaches past the end of the stream drops, decorrelation.py:656-658, instead of
 * becoming 0 * inf) by running the index-testing gather kernel instead of the LDS ones. */
vnd_status vnd_taps_create(vnd_ctx *ctx, int32_t num_channels,
                           const int32_t *tap_offsets, const int32_t *tap_index,
                           const float *tap_weight,
                           const int32_t *seg_offsets, const int32_t *seg_end,
                           const float *seg_gain, const uint8_t *chan_flags,
                           int32_t apply_gain, vnd_taps **taps);
vnd_status vnd_taps_destroy(vnd_taps *taps);
vnd_status vnd_taps_info(const vnd_taps *taps, int32_t *num_channels, int32_t *total_taps,
                         int32_t *max_index);
/* Packed image of a table for transport between ranks (RCCL broadcast of the
 * shared impulse tables): serialise on rank 0, broadcast bytes, rebuild.      */
vnd_status vnd_taps_serialize(const vnd_taps *taps, void *buf, int64_t capacity, int64_t *bytes);
vnd_status vnd_taps_deserialize(vnd_ctx *ctx, const void *buf, int64_t bytes, vnd_taps **taps);

/* Sharding helpers for hosts that bring their own process group (the many-stream mode of
 * decorrelation.py:649-658's loop over independent streams):
 * vnd_shard_range: rank `rank` of `world_size` owns streams [first, first + count) of `total` -
 *   contiguous blocks, the remainder one each to the lowest ranks (what distributed.shard_range cuts).
 * vnd_taps_broadcast_rccl: the one exchange of the path - the shared impulse table goes from `root` to
 *   every rank of an RCCL communicator the HOST created (ncclComm_t, passed as void*; one rank per GPU,
 *   the context's device being the communicator's).  On `root` *taps is the table to send and is left
 *   alone; on the other ranks *taps receives a new table the caller destroys.  Collective: every rank
 *   calls it, with the same root.  Synchronous (returns after the stream has drained).  librccl.so is
 *   loaded at first use: VND_ERR_UNSUPPORTED when it is not there.                                    */
vnd_status vnd_shard_range(int64_t total, int32_t world_size, int32_t rank, int64_t *first, int64_t *count);
vnd_status vnd_taps_broadcast_rccl(vnd_ctx *ctx, vnd_taps **taps, int32_t root, int32_t rank, void *rccl_comm,
                                   void *hip_stream);

/* ---- the hot path ------------------------------------------------------------
 * y[b,n,c] = sum_k w[c,k] * x[b, n + i[c,k], c]   (terms with n+i >= n_frames drop)
 * Replaces convolve_velvet_noise (decorrelation.py:630-660) and
 * VelvetNoise.convolve (decorrelation.py:393-415).  x and y must not overlap.
 * n_channels must equal the table's num_channels.  batch >= 0, n_frames >= 0.  */
vnd_status vnd_convolve_f32_dev(vnd_ctx *ctx, const vnd_taps *taps, const float *x_dev,
                                float *y_dev, int64_t batch, int64_t n_frames,
                                int32_t n_channels, int32_t mode, void *hip_stream);
vnd_status vnd_convolve_f32_host(vnd_ctx *ctx, const vnd_taps *taps, const float *x,
                                 float *y, int64_t batch, int64_t n_frames,
                                 int32_t n_channels, int32_t mode);

/* The same call for the operand types NumPy promotes: a float64 signal, or any signal with
 * a float64 filter (VelvetNoise.FIR, decorrelation.py:454-472) - there the reference forms
 * each product in float64 and adds it to the float32 output in float64, rounding at every
 * tap (decorrelation.py:656-658): acc = f32(f64(acc) + f64(x) * f64(w)).  Bit-identical to
 * that; function-path table given inline (host arrays, float64 weights), host pointers,
 * x float32 or float64 (x_is_f64), y float32.                                        */
vnd_status vnd_convolve_promote_host(vnd_ctx *ctx, int32_t num_channels, const int32_t *tap_offsets,
                                     const int32_t *tap_index, const double *tap_weight, const void *x,
                                     int32_t x_is_f64, float *y, int64_t batch, int64_t n_frames);

/* ---- the full stage: convolution + decorrelate epilogue on the device ----------
 * Replaces VelvetNoise.decorrelate after its float32 cast / mono->stereo
 * (decorrelation.py:431-440): convolve, then in place on y
 *   ms_encode  encode_signal_to_side_channel(x, y)      utils/dsp.py:40-63   (2 channels)
 *   use_width  apply_stereo_width(y, width)             utils/dsp.py:21-37   (2 channels)
 *   normalize  rms_normalize(x, y), DUAL_MONO           utils/dsp.py:87-109  (eps = 1e-10 upstream)
 * The pointwise steps are bit-identical to NumPy.  The normaliser's sums of squares:
 *   VND_MODE_EXACT, or normalize = VND_NORMALIZE_RMS_REFERENCE_ORDER in any mode:
 *     NumPy's own float32 summation order, reproduced bit for bit - the sequential row-by-row
 *     recurrence of an (n, C >= 2) array, the pairwise sums (8192-element chunks) of an (n, 1)
 *     one; in exact mode the whole stage is bit-identical to the reference;
 *   otherwise (normalize = VND_NORMALIZE_RMS): exactly rounded float64 sums, fused into the
 *     fast kernel - the fastest form, ~1e-4 relative from NumPy's RMS on long signals.
 * `workspace` is device memory of >= vnd_decorrelate_workspace_bytes().          */
#define VND_MAX_STREAMS 65535   /* streams (batch) per decorrelate / Haas call: split larger batches */
#define VND_NORMALIZE_OFF 0
#define VND_NORMALIZE_RMS 1
#define VND_NORMALIZE_RMS_REFERENCE_ORDER 2
vnd_status vnd_decorrelate_workspace_bytes(int64_t batch, int64_t n_frames, int32_t n_channels,
                                           int64_t *bytes);
vnd_status vnd_decorrelate_f32_dev(vnd_ctx *ctx, const vnd_taps *taps, const float *x_dev, float *y_dev,
                                   int64_t batch, int64_t n_frames, int32_t n_channels, int32_t mode,
                                   int32_t ms_encode, int32_t use_width, double width, int32_t normalize,
                                   float eps, void *workspace_dev, int64_t workspace_bytes,
                                   void *hip_stream);
vnd_status vnd_decorrelate_f32_host(vnd_ctx *ctx, const vnd_taps *taps, const float *x, float *y,
                                    int64_t batch, int64_t n_frames, int32_t n_channels, int32_t mode,
                                    int32_t ms_encode, int32_t use_width, double width, int32_t normalize,
                                    float eps);

/* ---- fan-out: fewer input channels than table channels ------------------------
 * x is [batch][n_frames][in_channels], y is [batch][n_frames][C] with C the table's
 * num_channels (a multiple of in_channels); output channel c reads input channel
 * c % in_channels:
 *   y[b,n,c] = sum_k w[c,k] * x[b, n + i[c,k], c % in_channels]
 * in_channels = 1, C = 2: VelvetNoise.decorrelate on a mono signal without the
 * mono_to_stereo copy (decorrelation.py:431-432, utils/dsp.py:112-115) - the input is
 * staged once per tile and read by both channels' taps.
 * in_channels = 2, C = 2F: one stereo signal through a bank of F filter pairs in a
 * single launch - the candidate scan of optimization.py:107-117 (grid_scan), with
 * the F tables concatenated channel-wise; y[..., 2f:2f+2] is filter f's output.
 * Results are those of vnd_convolve / vnd_decorrelate on the replicated input, bit
 * for bit in VND_MODE_EXACT.  The decorrelate epilogue pairs output channel c with
 * input channel c % in_channels.                                                 */
vnd_status vnd_convolve_fanout_f32_dev(vnd_ctx *ctx, const vnd_taps *taps, const float *x_dev,
                                       float *y_dev, int64_t batch, int64_t n_frames,
                                       int32_t in_channels, int32_t mode, void *hip_stream);
vnd_status vnd_convolve_fanout_f32_host(vnd_ctx *ctx, const vnd_taps *taps, const float *x,
                                        float *y, int64_t batch, int64_t n_frames,
                                        int32_t in_channels, int32_t mode);
vnd_status vnd_decorrelate_fanout_f32_dev(vnd_ctx *ctx, const vnd_taps *taps, const float *x_dev,
                                          float *y_dev, int64_t batch, int64_t n_frames,
                                          int32_t in_channels, int32_t mode, int32_t ms_encode,
                                          int32_t use_width, double width, int32_t normalize, float eps,
                                          void *workspace_dev, int64_t workspace_bytes, void *hip_stream);
vnd_status vnd_decorrelate_fanout_f32_host(vnd_ctx *ctx, const vnd_taps *taps, const float *x, float *y,
                                           int64_t batch, int64_t n_frames, int32_t in_channels,
                                           int32_t mode, int32_t ms_encode, int32_t use_width, double width,
                                           int32_t normalize, float eps);

/* ---- candidate scan: many filters, one signal, scores instead of signals -------
 * Replaces the loop of optimization.py:107-117 (grid_scan) over
 * symmetry_aware_objective (:46-105): the bank's F stereo pairs are convolved in one
 * fan-out launch and reduced on the device to VND_MOMENTS doubles per pair,
 *   {sum r, sum r*theta, sum r*theta^2, sum r*theta^3, max|theta|, sum L*R, sum L^2, sum R^2}
 * with r, theta the polar samples of utils/dsp.py:374-422 (MS angle, folded, radii not
 * normalised); the objective's terms are quotients of these.  float32 element maths as
 * NumPy, float64 sums: scores agree with the reference to ~1e-7 relative.
 * vnd_polar_moments_f32_dev reduces an existing device array y[n_frames][2*n_pairs];
 * vnd_scan_bank_f32_host runs the whole scan from a host signal (in_channels 1 or 2; the
 * bank table has 2*n_pairs channels) and returns moments[n_pairs][VND_MOMENTS].  The
 * candidates' convolution only (no side-channel encode / width / normaliser), which is
 * what optimize_velvet_noise builds (optimization.py:259-271: mode LR, normalizer None). */
#define VND_MOMENTS 8
vnd_status vnd_polar_moments_workspace_bytes(int64_t n_frames, int32_t n_pairs, int64_t *bytes);
vnd_status vnd_polar_moments_f32_dev(vnd_ctx *ctx, const float *y_dev, int64_t n_frames, int32_t n_pairs,
                                     double *moments_dev, void *workspace_dev, int64_t workspace_bytes,
                                     void *hip_stream);
vnd_status vnd_scan_bank_f32_host(vnd_ctx *ctx, const vnd_taps *bank, const float *x, int64_t n_frames,
                                  int32_t in_channels, int32_t mode, double *moments);

/* ---- HaasEffect on the device: the stage either side of the path in a chain ----
 * Replaces HaasEffect.decorrelate after its float32 cast (decorrelation.py:192-230):
 * x is float32 [batch][n_frames][in_channels] (1 = mono, duplicated; 2 = stereo), y is
 * FLOAT64 [batch][n_frames + delay_frames][2] as upstream (:206).  Column
 * `delayed_channel` - of the L/R pair, or of the mid/side pair when ms_mode - is delayed
 * by delay_frames = round(delay_time_seconds * fs) >= 0 with np.roll's wrap of the zero
 * tail (:220-222); use_width applies apply_stereo_width (utils/dsp.py:21-37).  float64
 * arithmetic in NumPy's operation order: bit-identical to the reference.           */
vnd_status vnd_haas_f64_dev(vnd_ctx *ctx, const float *x_dev, double *y_dev, int64_t batch,
                            int64_t n_frames, int32_t in_channels, int32_t delay_frames,
                            int32_t delayed_channel, int32_t ms_mode, int32_t use_width, double width,
                            void *hip_stream);
vnd_status vnd_haas_f64_host(vnd_ctx *ctx, const float *x, double *y, int64_t batch, int64_t n_frames,
                             int32_t in_channels, int32_t delay_frames, int32_t delayed_channel,
                             int32_t ms_mode, int32_t use_width, double width);

/* ---- host staging memory ---------------------------------------------------------
 * The reference returns a freshly allocated array from every call (out = np.zeros(...),
 * decorrelation.py:647).  A fresh pageable buffer costs a page fault per 4 KiB and a staged,
 * host-blocking download; page-locked memory takes the GPU's DMA directly.  These allocate /
 * release page-locked host memory (hipHostMalloc) for the "*_host" calls' buffers - the Python
 * layer draws its result arrays from a pool of them.  Any host pointer is accepted by the
 * "*_host" calls; pinned ones are just faster.  The "*_host" calls themselves cut a batch into
 * groups of streams on two HIP streams, so uploads, kernels and downloads overlap.        */
vnd_status vnd_host_alloc(int64_t bytes, void **ptr);
vnd_status vnd_host_free(void *ptr);
/* When BOTH buffers of vnd_convolve_f32_host / vnd_convolve_fanout_f32_host are page-locked and mapped
 * (vnd_host_alloc, hipHostMalloc, hipHostRegister), there is no staging at all: the
 * kernel reads x and writes y in place, across PCIe in both directions at once (VND_HOST_DIRECT=0: never).
 * *mapped = 1 if a call on these two buffers takes that path.  Results are the staged path's (bit for bit
 * in VND_MODE_EXACT; VND_MODE_FAST within its tolerance - the kernel may be cut into other launches). */
vnd_status vnd_host_buffers_mapped(const void *x, int64_t x_bytes, const void *y, int64_t y_bytes, int32_t *mapped);

/* ---- launch introspection -------------------------------------------------
 * (Measurement, tuning and diagnosis hooks - timing loops, the generated kernel sources, the
 *  variant override - are declared in vnd_amd_internal.h: exported, but not part of this ABI.) */
/* Describes the launch the library would make (for DESIGN.md / bench output). */
vnd_status vnd_describe_launch(vnd_ctx *ctx, const vnd_taps *taps, int64_t batch,
                               int64_t n_frames, int32_t n_channels, int32_t mode,
                               char *text, int32_t len);

vnd_status vnd_describe_fanout_launch(vnd_ctx *ctx, const vnd_taps *taps, int64_t batch,
                                      int64_t n_frames, int32_t in_channels, int32_t mode,
                                      char *text, int32_t len);

/* Builds, NOW, whatever per-table kernel a later vnd_convolve_* / vnd_decorrelate_* launch of this shape would
 * use (hipRTC, 1.5-5 s the first time a table meets a geometry; the code object is then kept with the table and
 * in the disk cache).  Launches themselves never stall for a build unless they are large (12 M frames and more
 * per channel pair): a host that will run many SMALL launches of one shape - a rank's shard of a batch, an audio
 * callback - calls this once, off its hot thread; without it such launches keep the generic kernels.
 * in_channels = channels of the input (1 for a mono input through a stereo table, else the table's).
 * Returns VND_OK also when no per-table kernel applies to the shape (the generic kernels need no preparation). */
vnd_status vnd_prepare_launch(vnd_ctx *ctx, const vnd_taps *taps, int64_t batch, int64_t n_frames,
                              int32_t in_channels, int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* VND_AMD_H */
