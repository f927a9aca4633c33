// vnd_haas.hpp - HaasEffect on the device, so that a chain of stages can
// stay in HBM: one channel - or the mid / side channel - of a stereo signal delayed by d
// frames (reference: src/vndecorrelate/decorrelation.py:192-230 with utils/dsp.py:21-37,
// :124-167).  The reference builds a zero-padded float64 (n + d, 2) array, converts to
// mid/side, np.roll's the delayed column by d (the d zero rows of the tail wrap to the
// front), converts back and applies the stereo width; every step is float64 arithmetic on
// float32 samples, restated here operation by operation, so the result is bit-identical.
// Memory-bound: 8 (or 4) bytes read and 16 written per output frame.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vnd {

constexpr int kHaasThreads = 256;

struct HArgs {
    const float *__restrict__ x;     // [batch][n][Cx]   Cx = 1 (mono, duplicated) or 2
    double *__restrict__ y;          // [batch][n + delay][2]
    int64_t n;
    int32_t Cx;
    int32_t delay;                   // >= 0
    int32_t delayed_channel;         // 0 | 1
    int32_t ms;                      // LayoutMode.MS
    int32_t use_width;
    double w_mid, w_side;            // 1.0 - width, width
};

// column c of the padded buffer before the roll, at frame k
__device__ __forceinline__ double haas_column(const HArgs &a, const float *__restrict__ xs, int c, int64_t k)
{
    if (k < 0 || k >= a.n) return 0.0;
    if (a.Cx == 1) return (double)xs[k];                      // mono_to_stereo: both columns (never re-encoded, :214)
    const double l = (double)xs[2 * k], r = (double)xs[2 * k + 1];
    if (!a.ms) return c == 0 ? l : r;
    return c == 0 ? (l + r) * 0.5 : (l - r) * 0.5;           // LR_to_MS
}

__global__ __launch_bounds__(kHaasThreads) void haas_kernel(const HArgs a)
{
    const int64_t total = a.n + a.delay;
    const int64_t k = (int64_t)blockIdx.x * kHaasThreads + threadIdx.x;
    if (k >= total) return;
    const float *__restrict__ xs = a.x + (int64_t)blockIdx.y * a.n * a.Cx;
    double *__restrict__ ys = a.y + (int64_t)blockIdx.y * total * 2;
    double v[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
        v[c] = haas_column(a, xs, c, c == a.delayed_channel ? k - a.delay : k);   // np.roll: zeros wrap in
    if (a.ms) {                                               // MS_to_LR, then the mono compensation (:226-229)
        const double l = v[0] + v[1], r = v[0] - v[1];
        v[0] = l; v[1] = r;
        if (a.Cx == 1) { v[0] = v[0] * 0.5; v[1] = v[1] * 0.5; }
    }
    if (a.use_width) {                                        // apply_stereo_width (utils/dsp.py:21-37)
        double m = (v[0] + v[1]) * 0.5, s = (v[0] - v[1]) * 0.5;
        m = m * a.w_mid;
        s = s * a.w_side;
        v[0] = m + s; v[1] = m - s;
    }
    *(double2 *)(ys + 2 * k) = make_double2(v[0], v[1]);
}

}  // namespace vnd
