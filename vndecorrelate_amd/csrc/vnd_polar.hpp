// vnd_polar.hpp - the polar sample of one stereo frame and its moments, shared by the
// stand-alone moments kernels (vnd_moments.hpp) and the convolution kernels' moments sink (vnd_kernels.hpp).
// Element maths in float32 like NumPy's (utils/dsp.py:374-422), sums in float64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vnd {

constexpr int kMoments = 8;

struct PolarAcc {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, lr = 0.0, ll = 0.0, rr = 0.0;
    float tmax = 0.0f;
};

__device__ __forceinline__ void polar_add(PolarAcc &a, float l, float r)
{
    const float kHalfPi = 1.57079632679489661923f, kPi = 3.14159265358979323846f;   // float32(np.pi / 2), float32(np.pi)
    float th = atan2f(l - r, l + r);
    if (th < -kHalfPi) th = th + kPi;                  // np.where(t < -pi/2, t + pi, np.where(t > pi/2, t - pi, t))
    else if (th > kHalfPi) th = th - kPi;
    const float rad = sqrtf(l * l + r * r);
    const float t2 = th * th;
    a.s0 += (double)rad;
    a.s1 += (double)(rad * th);
    a.s2 += (double)(rad * t2);
    a.s3 += (double)(rad * (t2 * th));
    a.tmax = fmaxf(a.tmax, fabsf(th));
    a.lr += (double)(l * r);
    a.ll += (double)(l * l);
    a.rr += (double)(r * r);
}

__device__ __forceinline__ void polar_store(double *out, const PolarAcc &a)
{
    out[0] = a.s0; out[1] = a.s1; out[2] = a.s2; out[3] = a.s3;
    out[4] = (double)a.tmax; out[5] = a.lr; out[6] = a.ll; out[7] = a.rr;
}


}  // namespace vnd
