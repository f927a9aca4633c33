// vnd_epilogue.hpp - device side of VelvetNoise.decorrelate's epilogue:
// side-channel encode, stereo width, per-channel RMS normalisation
// (reference: src/vndecorrelate/decorrelation.py:433-440, utils/dsp.py:21-63, :87-109).
//
// Two streaming passes over y behind the convolution:
//   pass 1  pointwise steps in the reference's float32 operation order (so they
//           are bit-identical to NumPy) + per-chunk sums of x^2 and y^2 per channel;
//   reduce  one workgroup per stream adds its rows of sums in a fixed order (double
//           precision, deterministic) and forms the per-channel scales;
//   pass 2  multiplies y by the scales.
// The reference sums the squares with NumPy's float32 axis-0 reduction, which is a
// plain sequential sum (relative error ~1e-4 on 10 s of audio); the scale computed
// here is the correctly rounded one, so the normalised output agrees with the
// reference to ~2e-4 of peak and with exact arithmetic to float32 rounding.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vnd_kernels.hpp"      // raw buffer loads

namespace vnd {

constexpr int kEpiThreads = 256;
constexpr int kEpiFramesPerThread = 16;
constexpr int kEpiChunk = kEpiThreads * kEpiFramesPerThread;   // frames per workgroup

struct EArgs {
    const float *__restrict__ x;
    float *__restrict__ y;
    double *__restrict__ partials;   // [batch][rows][2*C]: sum x_c^2 (c < C) then sum y_c^2
    float *__restrict__ scales;      // [batch][C], written by the reduce kernel
    int64_t n;
    int32_t C;
    int32_t Cx;                      // input channels: output channel c pairs with input channel c % Cx
    int32_t rows;                    // rows of partial sums per stream (pass-1 chunks, or tiles when fused)
    int32_t ms_encode;               // stereo only
    int32_t use_width;               // stereo only
    float w_mid, w_side;             // float32(1 - width), float32(width)
    int32_t normalize;
    int32_t exact_rms;               // sums are the reference's sequential float32 sums (one row per stream)
    float eps;
};

__device__ __forceinline__ double block_sum(double v, double *scratch)
{
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_xor(v, sh);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kEpiThreads / 64; ++w) t += scratch[w];      // fixed order
    return t;
}

// Pass 1.  Stereo fast path handles both pointwise steps; other channel counts only
// feed the sums (the host layer rejects MS / width for non-stereo, as upstream).
__global__ __launch_bounds__(kEpiThreads) void epilogue_pointwise_kernel(const EArgs a)
{
    __shared__ double scratch[kEpiThreads / 64];
    const int64_t b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int C = a.C, Cx = a.Cx;
    const float *__restrict__ xs = a.x + b * a.n * Cx;
    float *__restrict__ ys = a.y + b * a.n * C;
    const int64_t f0 = (int64_t)chunk * kEpiChunk;
    double *out = a.partials + (b * a.rows + chunk) * 2 * C;

    if (C == 2) {
        float sx0 = 0.f, sx1 = 0.f, sy0 = 0.f, sy1 = 0.f;
#pragma unroll 4
        for (int i = 0; i < kEpiFramesPerThread; ++i) {
            const int64_t f = f0 + threadIdx.x + (int64_t)i * kEpiThreads;
            if (f >= a.n) break;
            float2 xv;
            if (Cx == 2) xv = *(const float2 *)(xs + 2 * f);
            else { xv.x = xs[f]; xv.y = xv.x; }          // mono input fanned out to both channels
            float2 yv = *(const float2 *)(ys + 2 * f);
            if (a.ms_encode) {                       // utils/dsp.py:59-63
                const float mid = xv.x + xv.y;
                const float side = (yv.x - yv.y) * 0.5f;
                yv.x = (mid + side) * 0.5f;
                yv.y = (mid - side) * 0.5f;
            }
            if (a.use_width) {                       // utils/dsp.py:34-37 with :140-144, :163-167
                float m = (yv.x + yv.y) * 0.5f;
                float s = (yv.x - yv.y) * 0.5f;
                m = m * a.w_mid;
                s = s * a.w_side;
                yv.x = m + s;
                yv.y = m - s;
            }
            if (a.ms_encode || a.use_width) *(float2 *)(ys + 2 * f) = yv;
            sx0 += xv.x * xv.x; sx1 += xv.y * xv.y;
            sy0 += yv.x * yv.x; sy1 += yv.y * yv.y;
        }
        if (a.normalize) {
            const double r0 = block_sum((double)sx0, scratch), r1 = block_sum((double)sx1, scratch);
            const double r2 = block_sum((double)sy0, scratch), r3 = block_sum((double)sy1, scratch);
            if (threadIdx.x == 0) { out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3; }
        }
        return;
    }
    if (!a.normalize) return;
    for (int c = 0; c < C; ++c) {
        float sx = 0.f, sy = 0.f;
        for (int i = 0; i < kEpiFramesPerThread; ++i) {
            const int64_t f = f0 + threadIdx.x + (int64_t)i * kEpiThreads;
            if (f >= a.n) break;
            const float xv = xs[f * Cx + c % Cx], yv = ys[f * C + c];
            sx += xv * xv;
            sy += yv * yv;
        }
        const double rx = block_sum((double)sx, scratch), ry = block_sum((double)sy, scratch);
        if (threadIdx.x == 0) { out[c] = rx; out[C + c] = ry; }
    }
}

// Between the passes: one workgroup per stream adds that stream's rows of partial sums
// (strided per thread, then a fixed-order tree: deterministic) and writes the C scales
//   sqrt(mean(x_c^2)) / sqrt(mean(y_c^2) + eps)                     (utils/dsp.py:107-109)
__global__ __launch_bounds__(kEpiThreads) void epilogue_reduce_kernel(const EArgs a)
{
    __shared__ double scratch[kEpiThreads / 64];
    const int64_t b = blockIdx.x;
    const int C = a.C;
    const double *p = a.partials + b * a.rows * 2 * C;
    for (int c = 0; c < C; ++c) {
        double sx = 0.0, sy = 0.0;
        for (int k = threadIdx.x; k < a.rows; k += kEpiThreads) { sx += p[k * 2 * C + c]; sy += p[k * 2 * C + C + c]; }
        sx = block_sum(sx, scratch);
        sy = block_sum(sy, scratch);
        if (threadIdx.x == 0) {
            // np.mean: float32 sum / n with n as float32 (exact below 2^24 frames)
            const double count = a.exact_rms ? (double)(float)a.n : (double)a.n;
            const float mean_x = (float)(sx / count), mean_y = (float)(sy / count);
            const float rms_x = (float)sqrt((double)mean_x);
            const float rms_y = (float)sqrt((double)(mean_y + a.eps));
            a.scales[b * C + c] = (float)((double)rms_x / (double)rms_y);
        }
    }
}

// ---- the reference's own sum of squares, bit for bit (VND_MODE_EXACT, C >= 2) -----------------
// np.mean(np.square(a), axis=0) on a C-contiguous float32 (n, C >= 2) array adds the rows one
// after the other in float32 - acc[c] = f32(acc[c] + f32(a[i,c]^2)) - with no pairwise
// splitting (measured against np.cumsum for n up to 2.9e6).  That recurrence
// cannot be parallelised without changing its roundings, so ONE lane walks each
// (stream, array, channel) chain while the whole wave keeps it fed: all 64 lanes load the next
// block of frames and square it into LDS (chain-major), the 2C chain lanes then add their row of
// the block, 64 squares per LDS burst.  The cost is the dependent-add latency, ~6 cycles per
// frame whatever the batch (one wave per stream, up to a thousand streams side by side):
// ~1.3 ms for 10 s of 48 kHz audio.  Writes one row of 2C sums per stream (as doubles, exact).
constexpr int kSeqFrames = 2048;     // frames per block: its adds (~7 us) cover the next block's HBM latency
constexpr int kSeqLanes = 64;

// the chain lanes' part of one block: 64 squares per LDS burst, the next burst in flight while the
// current one is added (the adds are a dependent chain, ~6.6 cycles each: nothing else to overlap)
__device__ __forceinline__ float seq_add64(const float4 (&v)[16], float acc)
{
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        acc = acc + v[u].x;
        acc = acc + v[u].y;
        acc = acc + v[u].z;
        acc = acc + v[u].w;
    }
    return acc;
}

__device__ __forceinline__ float seq_add_block(const float *row_base, float acc)
{
    static_assert(kSeqFrames % 128 == 0, "two bursts of 64 squares per iteration");
    const float4 *__restrict__ row = (const float4 *)row_base;
    float4 va[16], vb[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) va[u] = row[u];
#pragma unroll 1
    for (int i = 0; i < kSeqFrames / 4; i += 32) {
#pragma unroll
        for (int u = 0; u < 16; ++u) vb[u] = row[i + 16 + u];
        acc = seq_add64(va, acc);
        const int next = i + 32 < kSeqFrames / 4 ? i + 32 : i;       // the last prefetch re-reads: harmless
#pragma unroll
        for (int u = 0; u < 16; ++u) va[u] = row[next + u];
        acc = seq_add64(vb, acc);
    }
    return acc;
}

// Stereo: branch-free staging.  Each lane takes 4 consecutive frames per 256-frame round through
// range-checked buffer loads (frames past the end read 0 and add +0: exact) and writes one float4
// per chain; the NEXT block's loads are issued before the chain lanes start adding the current
// one.  (A first version with per-frame bounds branches spent more time staging than adding: one
// wave alone on its SIMD pays ~20 cycles per scalar branch.)
template <bool MONO>
__global__ __launch_bounds__(kSeqLanes) void epilogue_rms_seq_stereo_kernel(const EArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sq[];      // [4][kSeqFrames]
    constexpr int ROUND = 4 * kSeqLanes;                             // frames per round
    constexpr int PER = kSeqFrames / ROUND;
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const v4i rx = make_rsrc(a.x + b * a.n * (MONO ? 1 : 2), a.n * (MONO ? 4 : 8));
    const v4i ry = make_rsrc(a.y + b * a.n * 2, a.n * 8);
    v2f xr[PER][4], yr[PER][4];
    auto fetch = [&](int f0) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int fr = f0 + u * ROUND + 4 * lane + k;
                yr[u][k] = buf_load2(ry, fr * 8, 0, 0);
                if constexpr (MONO) {
                    const float v = buf_load1(rx, fr * 4, 0, 0);
                    xr[u][k] = v2f{v, v};
                } else {
                    xr[u][k] = buf_load2(rx, fr * 8, 0, 0);
                }
            }
        }
    };
    float acc = 0.0f;
    const int n = (int)a.n;                                // the host keeps 8 n below 2^31
    fetch(0);
    for (int f0 = 0; f0 < n; f0 += kSeqFrames) {
        __syncthreads();                                   // the chain lanes are done with the previous block
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            float4 *dst = (float4 *)(sq + u * ROUND + 4 * lane);
            dst[0 * kSeqFrames / 4] = make_float4(xr[u][0].x * xr[u][0].x, xr[u][1].x * xr[u][1].x,
                                                  xr[u][2].x * xr[u][2].x, xr[u][3].x * xr[u][3].x);
            dst[1 * kSeqFrames / 4] = make_float4(xr[u][0].y * xr[u][0].y, xr[u][1].y * xr[u][1].y,
                                                  xr[u][2].y * xr[u][2].y, xr[u][3].y * xr[u][3].y);
            dst[2 * kSeqFrames / 4] = make_float4(yr[u][0].x * yr[u][0].x, yr[u][1].x * yr[u][1].x,
                                                  yr[u][2].x * yr[u][2].x, yr[u][3].x * yr[u][3].x);
            dst[3 * kSeqFrames / 4] = make_float4(yr[u][0].y * yr[u][0].y, yr[u][1].y * yr[u][1].y,
                                                  yr[u][2].y * yr[u][2].y, yr[u][3].y * yr[u][3].y);
        }
        __syncthreads();
#if !(defined(VND_SEQ_ABLATE) && VND_SEQ_ABLATE == 2)
        if (f0 + kSeqFrames < n) fetch(f0 + kSeqFrames);
#endif
#if !(defined(VND_SEQ_ABLATE) && VND_SEQ_ABLATE == 1)
        if (lane < 4) acc = seq_add_block(sq + lane * kSeqFrames, acc);
#endif
    }
    if (lane < 4) a.partials[b * a.rows * 4 + lane] = (double)acc;
}

// Any channel count >= 2 (plain loop, no prefetch).
__global__ __launch_bounds__(kSeqLanes) void epilogue_rms_seq_kernel(const EArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sq[];      // [2C][kSeqFrames]
    const int lane = threadIdx.x;
    const int C = a.C, Cx = a.Cx;
    const int64_t b = blockIdx.x;
    const float *__restrict__ xs = a.x + b * a.n * Cx;
    const float *__restrict__ ys = a.y + b * a.n * C;
    const int chains = 2 * C;
    float acc = 0.0f;
    for (int64_t f0 = 0; f0 < a.n; f0 += kSeqFrames) {
        __syncthreads();
        for (int e = lane; e < kSeqFrames * C; e += kSeqLanes) {
            const int f = e / C, c = e - f * C;
            const int64_t fr = f0 + f;
            const float xv = fr < a.n ? xs[fr * Cx + c % Cx] : 0.f;
            const float yv = fr < a.n ? ys[fr * C + c] : 0.f;
            sq[c * kSeqFrames + f] = xv * xv;
            sq[(C + c) * kSeqFrames + f] = yv * yv;
        }
        __syncthreads();
        if (lane < chains) acc = seq_add_block(sq + lane * kSeqFrames, acc);
    }
    if (lane < chains) a.partials[b * a.rows * 2 * C + lane] = (double)acc;
}

// Pass 2: y[:, c] *= scale[c]
__global__ __launch_bounds__(kEpiThreads) void epilogue_scale_kernel(const EArgs a)
{
    const int64_t b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int C = a.C;
    const float *__restrict__ scale = a.scales + b * C;
    float *__restrict__ ys = a.y + b * a.n * C;
    const int64_t e0 = (int64_t)chunk * kEpiChunk * C;
    const int64_t e1 = min(e0 + (int64_t)kEpiChunk * C, a.n * C);
    if (C == 2) {
        const float s0 = scale[0], s1 = scale[1];
        if ((((uintptr_t)ys) & 15) == 0) {               // 16 B per lane: two frames
            for (int64_t e = e0 + 4 * threadIdx.x; e < e1; e += 4 * kEpiThreads) {
                if (e + 4 <= e1) {
                    float4 v = *(float4 *)(ys + e);
                    v.x = v.x * s0; v.y = v.y * s1; v.z = v.z * s0; v.w = v.w * s1;
                    *(float4 *)(ys + e) = v;
                } else {
                    float2 v = *(float2 *)(ys + e);
                    v.x = v.x * s0; v.y = v.y * s1;
                    *(float2 *)(ys + e) = v;
                }
            }
        } else {
            for (int64_t e = e0 + 2 * threadIdx.x; e < e1; e += 2 * kEpiThreads) {
                float2 v = *(float2 *)(ys + e);
                v.x = v.x * s0;
                v.y = v.y * s1;
                *(float2 *)(ys + e) = v;
            }
        }
    } else {
        for (int64_t e = e0 + threadIdx.x; e < e1; e += kEpiThreads) ys[e] = ys[e] * scale[(int)(e % C)];
    }
}

}  // namespace vnd
