// vnd_kernels.hpp - gfx950 (CDNA4, wave64) kernels of the velvet-noise tap sum.
//
//   y[b,n,c] = sum_k w[c,k] * x[b, n + i[c,k], c]        (n + i >= N drops out)
//
// What the reference does with K NumPy slice-adds per channel
// (ckonst/VNDecorrelate src/vndecorrelate/decorrelation.py:649-658 and
// :402-414) is, on the GPU, a *gather* with a forward halo of max(i) frames.
// It is HBM-streaming work (8 algorithmic bytes per output sample) whose
// on-chip cost is the K LDS reads per output, so the design is about LDS
// bandwidth, not FLOPs - no MFMA anywhere:
//
//   * one workgroup = one time tile of one stream (x CG channels);
//   * the tile + halo is read once from HBM with coalesced vector loads and
//     de-interleaved into per-channel LDS planes, so every later LDS read is a
//     unit-stride, conflict-free access;
//   * each lane owns PAIRS of consecutive frames: a tap is one ds_read_b64
//     (the widest conflict-free form: 256 B/clk/CU) feeding two accumulators.
//     Odd tap offsets would be 4-byte-misaligned b64 reads, so (DUAL) a second
//     plane shifted by one frame keeps them aligned, or (!DUAL) two aligned
//     reads straddle the pair;
//   * tap (index, weight) records are wave-uniform: a wave parks 64 of them in
//     two VGPRs and broadcasts one per step with v_readlane - no memory access
//     of any kind for the table inside the tap loop;
//   * outputs leave as 16-byte interleaved stores, 1 KiB per wave-instruction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vnd {

constexpr int kThreads = 256;   // 4 waves; several workgroups share a CU

typedef float v2f __attribute__((ext_vector_type(2)));

// One aligned ds_read_b64.  The access is volatile on purpose: hipcc otherwise
// fuses neighbouring pairs into ds_read2st64_b64 / ds_read2_b32, which move half
// the bytes per LDS cycle of plain ds_read_b64 (MI355X LDS table: 128 vs 256 B/clk).
__device__ __forceinline__ v2f lds_pair(const float *p)
{
    typedef __attribute__((address_space(3))) const volatile v2f *lds_v2f_ptr;
    return *(lds_v2f_ptr)p;
}

struct Tap {          // 8-byte record -> one s_load_dwordx2 per tap
    int32_t idx;
    float w;
};

struct KArgs {
    const float *__restrict__ x;
    float *__restrict__ y;
    const Tap *__restrict__ taps;
    const int32_t *__restrict__ tap_off;    // [C+1]
    const int32_t *__restrict__ seg_off;    // [C+1] or nullptr (function-path table)
    const int32_t *__restrict__ seg_end;    // exclusive ends, absolute tap positions
    const float *__restrict__ seg_gain;
    const uint8_t *__restrict__ chan_flags; // bit0: pass-through, or nullptr
    int64_t n;                 // frames per stream
    int32_t C;                 // interleaved channels
    int32_t groups;            // C / CG
    int32_t tiles;             // tiles per stream
    int32_t W;                 // floats per LDS plane (tile + halo, even)
    int32_t apply_gain;
    uint32_t nblocks;
};

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Map
// them so that one XCD walks CONSECUTIVE logical tiles: a tile's halo is the
// next tile's head, and channel groups of one tile share cache lines, so both
// re-reads hit that XCD's L2 instead of going back to the fabric.  Bijective
// for any grid size.  Placement only changes speed, never results.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nwg)
{
    const uint32_t q = nwg >> 3, r = nwg & 7u, xcd = b & 7u;
    const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (b >> 3);
}

template <int MODE>
__device__ __forceinline__ float tap_op(float acc, float v, float w)
{
    if constexpr (MODE == 0) {
        float p = v * w;          // file is built with -ffp-contract=off: two roundings,
        return acc + p;           // exactly NumPy's  out += x * w
    } else {
        return __builtin_fmaf(v, w, acc);
    }
}

// CG   channels handled per workgroup (C % CG == 0)
// R    frame pairs per lane (tile = 2 * kThreads * R frames)
// MODE 0 exact (mul, add)  1 fma
// DUAL second LDS plane shifted by one frame for odd tap offsets
template <int CG, int R, int MODE, bool DUAL>
__global__ __launch_bounds__(kThreads) void conv_lds_kernel(const KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = 2 * kThreads * R;
    const int tid = threadIdx.x;
    const int W = a.W;

    uint32_t lid = xcd_remap(blockIdx.x, a.nblocks);
    const int g = lid % (uint32_t)a.groups;
    lid /= (uint32_t)a.groups;
    const int tile = lid % (uint32_t)a.tiles;
    const int64_t b = lid / (uint32_t)a.tiles;

    const int C = a.C;
    const int c0 = g * CG;
    const int64_t t0 = (int64_t)tile * T;
    const float *__restrict__ xs = a.x + b * a.n * C;
    float *__restrict__ ys = a.y + b * a.n * C;

    float *planeA = lds;                       // [CG][W]   dword m = x[t0 + m]
    float *planeB = lds + (DUAL ? CG * W : 0); // [CG][W]   dword m = x[t0 + m + 1]
    // (plane c of B sits CG*W floats after plane c of A: one base serves both)

    // ---- stage tile + halo: coalesced HBM read, de-interleave into planes ----
    {
        const int64_t remain = a.n - t0;                   // frames available from t0
        const int valid = remain < W ? (int)remain : W;
        const bool vec_ok = (CG == 1) ||
            ((((uintptr_t)(xs + c0)) & (CG * 4 - 1)) == 0 && (C % CG) == 0);
        for (int f = tid; f < W; f += kThreads) {
            float v[CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) v[c] = 0.0f;
            if (f < valid) {
                const float *src = xs + (t0 + f) * C + c0;
                if constexpr (CG == 2) {
                    if (vec_ok) { float2 t = *(const float2 *)src; v[0] = t.x; v[1] = t.y; }
                    else { v[0] = src[0]; v[1] = src[1]; }
                } else if constexpr (CG == 4) {
                    if (vec_ok) { float4 t = *(const float4 *)src; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
                    else { v[0] = src[0]; v[1] = src[1]; v[2] = src[2]; v[3] = src[3]; }
                } else {
#pragma unroll
                    for (int c = 0; c < CG; ++c) v[c] = src[c];
                }
            }
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                planeA[c * W + f] = v[c];
                if constexpr (DUAL) { if (f > 0) planeB[c * W + f - 1] = v[c]; }
            }
        }
        if constexpr (DUAL) {
            if (tid < CG) planeB[tid * W + W - 1] = 0.0f;   // never read; keep it defined
        }
    }
    __syncthreads();

    // ---- tap sum: every lane owns R pairs of consecutive frames ----------------
    // Tap records are wave-uniform.  Each wave keeps a chunk of 64 of them in two
    // VGPRs (lane l <-> tap l) and broadcasts one per iteration with v_readlane:
    // no scalar-memory round trip sits inside the tap loop.
    float2 out[CG][R];
    const int lane = tid & 63;
    const int lane_base = 2 * tid;
    const bool has_seg = a.seg_off != nullptr;

#pragma unroll
    for (int c = 0; c < CG; ++c) {
        const int ch = c0 + c;
        const float *pa = planeA + c * W + lane_base;
        const bool pass = a.chan_flags != nullptr && (a.chan_flags[ch] & 1);
        if (pass) {
#pragma unroll
            for (int j = 0; j < R; ++j) out[c][j] = *(const float2 *)(pa + 2 * kThreads * j);
            continue;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) out[c][j] = make_float2(0.0f, 0.0f);

        int k = a.tap_off[ch];
        const int k_last = a.tap_off[ch + 1];
        const int s_begin = has_seg ? a.seg_off[ch] : 0;
        const int nseg = has_seg ? a.seg_off[ch + 1] - s_begin : 1;
        int chunk = k;                      // first tap held in the VGPR chunk
        int tv_off = 0;
        float tv_w = 0.0f;
        auto load_chunk = [&](int base) {
            const int kk = base + lane;
            Tap t;
            t.idx = 0; t.w = 0.0f;
            if (kk < k_last) t = a.taps[kk];
            // DUAL: odd offsets read the shifted plane at an even position
            tv_off = (DUAL && (t.idx & 1)) ? CG * W + t.idx - 1 : t.idx;
            tv_w = t.w;
        };
        load_chunk(chunk);
        for (int s = 0; s < nseg; ++s) {
            const int kend = has_seg ? a.seg_end[s_begin + s] : k_last;
            float2 sb[R];
#pragma unroll
            for (int j = 0; j < R; ++j) sb[j] = make_float2(0.0f, 0.0f);
            for (; k < kend; ++k) {
                if (k - chunk >= 64) { chunk += 64; load_chunk(chunk); }
                const int off = __builtin_amdgcn_readlane(tv_off, k - chunk);
                const float w = __builtin_bit_cast(float,
                    __builtin_amdgcn_readlane(__builtin_bit_cast(int, tv_w), k - chunk));
                if (DUAL || (off & 1) == 0) {
                    const float *p = pa + off;
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        const v2f v = lds_pair(p + 2 * kThreads * j);
                        sb[j].x = tap_op<MODE>(sb[j].x, v.x, w);
                        sb[j].y = tap_op<MODE>(sb[j].y, v.y, w);
                    }
                } else {
                    const float *p = pa + (off - 1);     // two aligned pairs straddle ours
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        const v2f lo = lds_pair(p + 2 * kThreads * j);
                        const v2f hi = lds_pair(p + 2 * kThreads * j + 2);
                        sb[j].x = tap_op<MODE>(sb[j].x, lo.y, w);
                        sb[j].y = tap_op<MODE>(sb[j].y, hi.x, w);
                    }
                }
            }
            if (has_seg) {
                if (a.apply_gain) {
                    const float gain = a.seg_gain[s_begin + s];
#pragma unroll
                    for (int j = 0; j < R; ++j) { sb[j].x = sb[j].x * gain; sb[j].y = sb[j].y * gain; }
                }
#pragma unroll
                for (int j = 0; j < R; ++j) { out[c][j].x = out[c][j].x + sb[j].x; out[c][j].y = out[c][j].y + sb[j].y; }
            } else {
#pragma unroll
                for (int j = 0; j < R; ++j) out[c][j] = sb[j];
            }
        }
    }

    // ---- interleaved store: 2 frames x CG channels per lane and pair ------------
    const bool st_vec = (C == CG) && ((((uintptr_t)ys) & 15) == 0) && (CG == 2 || CG == 4);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int64_t n0 = t0 + lane_base + 2 * kThreads * j;
        if (n0 >= a.n) continue;
        float *dst = ys + n0 * C + c0;
        const bool two = n0 + 1 < a.n;
        if constexpr (CG == 2) {
            if (st_vec && two) {
                *(float4 *)dst = make_float4(out[0][j].x, out[1][j].x, out[0][j].y, out[1][j].y);
                continue;
            }
        }
        if constexpr (CG == 4) {
            if (st_vec) {
                *(float4 *)dst = make_float4(out[0][j].x, out[1][j].x, out[2][j].x, out[3][j].x);
                if (two) *(float4 *)(dst + C) = make_float4(out[0][j].y, out[1][j].y, out[2][j].y, out[3][j].y);
                continue;
            }
        }
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            dst[c] = out[c][j].x;
            if (two) dst[C + c] = out[c][j].y;
        }
    }
}

// Fallback without LDS staging, for halos that do not fit a workgroup's LDS
// (very long FIRs): one lane per (frame, channel), taps gathered through L1/L2.
template <int MODE>
__global__ __launch_bounds__(kThreads) void conv_direct_kernel(const KArgs a)
{
    const int64_t per_stream = a.n * a.C;
    const int64_t total = per_stream * (int64_t)a.tiles;   // tiles carries the batch here
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * kThreads) {
        const int64_t b = e / per_stream;
        const int64_t r = e - b * per_stream;
        const int64_t n0 = r / a.C;
        const int ch = (int)(r - n0 * a.C);
        const float *__restrict__ xs = a.x + b * per_stream;
        if (a.chan_flags != nullptr && (a.chan_flags[ch] & 1)) { a.y[e] = xs[r]; continue; }
        const bool has_seg = a.seg_off != nullptr;
        int k = a.tap_off[ch];
        const int k_last = a.tap_off[ch + 1];
        const int s_begin = has_seg ? a.seg_off[ch] : 0;
        const int nseg = has_seg ? a.seg_off[ch + 1] - s_begin : 1;
        float out = 0.0f;
        for (int s = 0; s < nseg; ++s) {
            const int kend = has_seg ? a.seg_end[s_begin + s] : k_last;
            float sb = 0.0f;
            for (; k < kend; ++k) {
                const Tap tp = a.taps[k];
                const int64_t m = n0 + tp.idx;
                const float v = m < a.n ? xs[m * a.C + ch] : 0.0f;
                sb = tap_op<MODE>(sb, v, tp.w);
            }
            if (has_seg) {
                if (a.apply_gain) sb = sb * a.seg_gain[s_begin + s];
                out = out + sb;
            } else {
                out = sb;
            }
        }
        a.y[e] = out;
    }
}

}  // namespace vnd
