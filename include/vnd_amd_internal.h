/* vnd_amd_internal.h - measurement, tuning and diagnosis hooks of libvnd_amd.so.
 *
 * NOT part of the drop-in ABI (include/vnd_amd.h): bench.py, tools/ and the tests use these to time
 * launches, to read the per-table kernel sources without a device, and to force kernel variants.
 * They are exported by the same shared library, may change with any build, and a host that only
 * decorrelates never needs them.  Same conventions as vnd_amd.h (plain C, vnd_status, no torch types).
 */
#ifndef VND_AMD_INTERNAL_H
#define VND_AMD_INTERNAL_H

#include "vnd_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement helpers (used by bench.py; not on the data path) ---------- */
/* Launches the convolve `iters` times back to back on `hip_stream`, cycling
 * through `n_buffers` (x,y) pairs laid out at x_dev + i*stride_elems, and
 * returns the average kernel milliseconds between two hipEvents recorded on
 * that same stream.                                                          */
vnd_status vnd_time_convolve_f32_dev(vnd_ctx *ctx, const vnd_taps *taps, const float *x_dev,
                                     float *y_dev, int64_t batch, int64_t n_frames,
                                     int32_t n_channels, int32_t mode, int32_t n_buffers,
                                     int64_t stride_elems, int32_t iters, void *hip_stream,
                                     float *avg_ms);
/* Bytes of private (scratch) memory per lane the kernel `kernel` of a gfx950 code object (an ELF image)
 * asks for, read from its kernel descriptor: > 0 means the compiler spilled registers.  The library
 * rejects such builds of the window form of its per-table kernels (the registers ARE that kernel);
 * exposed so that the check can be tested without a device.  -1: no such kernel in the image.        */
vnd_status vnd_code_object_private_bytes(const void *code, int64_t bytes, const char *kernel,
                                         int64_t *private_bytes);
/* The box's streaming ceiling, as a companion of the 8 TB/s figure: `iters` plain copies of `elems`
 * floats (a multiple of 4; 16-byte aligned buffers) with 16-byte non-temporal accesses, the average
 * kernel milliseconds between two hipEvents on `hip_stream`.                                    */
vnd_status vnd_time_copy_f32_dev(vnd_ctx *ctx, const float *x_dev, float *y_dev, int64_t elems,
                                 int32_t iters, void *hip_stream, float *avg_ms);
/* VND_MODE_FAST and VND_MODE_EXACT compile a kernel PER TAP TABLE with hipRTC on first use
 * (offsets become LDS-read immediates, weights literals; persistent workgroups over an LDS ring;
 * `mode` picks the arithmetic: free summation order, or the reference's own association bit for
 * bit); the generic kernels take over whenever that is not possible.  This returns the HIP source the
 * library would hand to hipRTC for a function-path table - no device needed - so that it can be
 * audited or compiled offline (`hipcc --offload-arch=gfx950 -include hip/hip_runtime.h`).
 * text == NULL queries the size.  Even channel counts only (channel pairs share a workgroup). */
vnd_status vnd_spec_kernel_source(int32_t num_channels, const int32_t *tap_offsets,
                                  const int32_t *tap_index, const float *tap_weight, int32_t mode,
                                  char *text, int64_t capacity, int64_t *bytes);
/* The WINDOW form of the per-table kernel (stereo tables): a lane owns `frames_per_lane` (16 | 32 | 64)
 * consecutive output frames and reads the union of its taps' windows from LDS once.
 * Same contract as vnd_spec_kernel_source; the table as for vnd_taps_create (seg_* NULL: function path);
 * `threads` = workgroup size (multiple of 64).  When `lds_bytes_per_tile` / `fmas_per_tile` are non-NULL
 * they receive what ONE lane reads from LDS for its tap sums per tile and the (tap, output) products that
 * feeds - the kernel's figure of merit. */
vnd_status vnd_window_kernel_source(int32_t num_channels, const int32_t *tap_offsets,
                                    const int32_t *tap_index, const float *tap_weight,
                                    const int32_t *seg_offsets, const int32_t *seg_end,
                                    const float *seg_gain, int32_t apply_gain, int32_t mode,
                                    int32_t frames_per_lane, int32_t threads, char *text,
                                    int64_t capacity, int64_t *bytes, int64_t *lds_bytes_per_tile,
                                    int64_t *fmas_per_tile);
/* Kernel variant override for tuning runs: -1 = automatic choice. */
vnd_status vnd_set_variant(vnd_ctx *ctx, int32_t variant);
/* Reads a tuning variable the way the launch planner does (VND_TUNING=1 sessions only; otherwise *value = fallback).  A name that
 * is not in the library's registry (kTuningNames, csrc/vnd_spec.hpp) returns VND_ERR_INVALID with the name in vnd_last_error -
 * the same report a launch gives when the LIBRARY reads such a name (until round 5 a debug-build abort()) - and leaves no trace. */
vnd_status vnd_tuning_read(const char *name, int32_t fallback, int32_t *value);
/* Diagnosis (VND_TUNING=1 with VND_WIN_STAMPS=<workgroups>): the window-form kernel that a launch of this shape
 * uses was built with phase stamps - wave 0 of its first workgroups leaves the 100 MHz wall clock at its phase
 * boundaries, 16 uint64 per workgroup ([0] start, [1] ring filled, then per tile: taps done, window dead, outputs
 * exchanged, refill published; [15] = HW_ID | XCC_ID << 32).  Copies up to `capacity` uint64 of the last launch's
 * stamps into `stamps`; *count = how many the kernel holds (0: this launch's kernel has none).        */
vnd_status vnd_debug_read_stamps(vnd_ctx *ctx, const vnd_taps *taps, int64_t batch, int64_t n_frames,
                                 int32_t in_channels, int32_t mode, uint64_t *stamps, int64_t capacity,
                                 int64_t *count);

#ifdef __cplusplus
}
#endif
#endif /* VND_AMD_INTERNAL_H */
