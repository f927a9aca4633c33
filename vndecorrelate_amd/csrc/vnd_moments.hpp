// vnd_moments.hpp - device side of the optimiser's candidate scan.
//
// The reference scores a candidate filter by decorrelating the signal and taking
// amplitude-weighted angular moments of the result's polar samples
// (src/vndecorrelate/optimization.py:46-105 with utils/dsp.py:374-422), one candidate
// after the other (optimization.py:107-117).  Here the F candidates are convolved in ONE
// fan-out launch (y = [n][2F], stereo pair f at channels 2f, 2f+1) and this file reduces
// that array to eight numbers per candidate, so only 64 F bytes go back to the host:
//
//   0  sum r            r     = sqrt(L^2 + R^2)                        (dsp.py:414)
//   1  sum r*theta      theta = atan2(L - R, L + R), folded onto [-pi/2, pi/2]  (dsp.py:401-412)
//   2  sum r*theta^2
//   3  sum r*theta^3
//   4  max |theta|
//   5  sum L*R          for left_right_correlation (optimization.py:11-17)
//   6  sum L^2
//   7  sum R^2
//
// theta, r and the products are float32 like NumPy's (the inputs are float32 arrays); the
// sums run in float64 where NumPy adds float32 pairwise, so a score agrees with the
// reference to ~1e-7 relative, not bit for bit.  Frames past the end do not exist here
// (the array is exactly n frames), and a silent frame has r = 0 and theta = 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vnd_polar.hpp"

namespace vnd {

constexpr int kMomThreads = 256;
constexpr int kMomFrames = 512;        // frames per workgroup

struct MArgs {
    const float *__restrict__ y;       // [n][2*F]
    double *__restrict__ partials;     // [chunks][F][8]
    double *__restrict__ moments;      // [F][8]
    int64_t n;
    int32_t F;
    int32_t chunks;
};

// Wide banks (F >= 64): a lane is a candidate, so a wave reads 64 consecutive stereo pairs of
// one frame (512 contiguous bytes) and nothing is reduced across lanes.
// grid = (chunks, ceil(F / 256))
__global__ __launch_bounds__(kMomThreads) void moments_by_candidate_kernel(const MArgs a)
{
    const int f = blockIdx.y * kMomThreads + threadIdx.x;
    if (f >= a.F) return;
    const int64_t n0 = (int64_t)blockIdx.x * kMomFrames;
    const int64_t n1 = min(n0 + kMomFrames, a.n);
    const float2 *__restrict__ p = (const float2 *)a.y + n0 * a.F + f;
    PolarAcc acc;
    for (int64_t i = n0; i < n1; ++i, p += a.F) {
        const float2 v = *p;
        polar_add(acc, v.x, v.y);
    }
    polar_store(a.partials + ((int64_t)blockIdx.x * a.F + f) * kMoments, acc);
}

// Narrow banks: a workgroup takes a chunk of frames of ONE candidate, lanes along time.
// grid = (chunks, F)
__global__ __launch_bounds__(kMomThreads) void moments_by_frame_kernel(const MArgs a)
{
    __shared__ double red[kMomThreads / 64][kMoments];
    const int f = blockIdx.y;
    const int64_t n0 = (int64_t)blockIdx.x * kMomFrames;
    const int64_t n1 = min(n0 + kMomFrames, a.n);
    PolarAcc acc;
    for (int64_t i = n0 + threadIdx.x; i < n1; i += kMomThreads) {
        const float2 v = *((const float2 *)a.y + i * a.F + f);
        polar_add(acc, v.x, v.y);
    }
    double v[kMoments];
    polar_store(v, acc);
#pragma unroll
    for (int k = 0; k < kMoments; ++k) {
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) {
            const double o = __shfl_xor(v[k], sh);
            v[k] = k == 4 ? fmax(v[k], o) : v[k] + o;
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < kMoments; ++k) red[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < kMoments) {
        const int k = threadIdx.x;
        double t = red[0][k];
        for (int w = 1; w < kMomThreads / 64; ++w) t = k == 4 ? fmax(t, red[w][k]) : t + red[w][k];   // fixed order
        a.partials[((int64_t)blockIdx.x * a.F + f) * kMoments + k] = t;
    }
}

// One workgroup per candidate: threads stride over the chunks (each reads the chunk's 8 contiguous
// doubles), then a fixed-order tree across the workgroup - the same order every run.
__global__ __launch_bounds__(kMomThreads) void moments_reduce_kernel(const MArgs a)
{
    __shared__ double red[kMomThreads / 64][kMoments];
    const int f = blockIdx.x;
    double v[kMoments];
#pragma unroll
    for (int k = 0; k < kMoments; ++k) v[k] = 0.0;
    for (int c = threadIdx.x; c < a.chunks; c += kMomThreads) {
        const double *p = a.partials + ((int64_t)c * a.F + f) * kMoments;
#pragma unroll
        for (int k = 0; k < kMoments; ++k) v[k] = k == 4 ? fmax(v[k], p[k]) : v[k] + p[k];
    }
#pragma unroll
    for (int k = 0; k < kMoments; ++k) {
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) {
            const double o = __shfl_xor(v[k], sh);
            v[k] = k == 4 ? fmax(v[k], o) : v[k] + o;
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < kMoments; ++k) red[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < kMoments) {
        const int k = threadIdx.x;
        double t = red[0][k];
        for (int w = 1; w < kMomThreads / 64; ++w) t = k == 4 ? fmax(t, red[w][k]) : t + red[w][k];
        a.moments[(int64_t)f * kMoments + k] = t;
    }
}

}  // namespace vnd
