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
    int32_t wide;                    // every stream of x and y starts 16-byte aligned (8-byte for a mono x): the sums kernel loads two frames per access
    int32_t seq_split;               // sequential sums, stereo: TWO workgroups per stream - one for x's chains, one for y's (grid = 2 * batch)
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
// splitting (measured against np.cumsum for n up to 2.9e6).  The recurrence
// cannot be re-associated in floating point, but it can be settled in INTEGERS a group at a
// time: while acc = A * ulp stays inside one binade (A in [2^23, 2^24), ulp = 2^(e-23)), adding
// s >= 0 gives  f32(acc + s) = (A + q + [r > ulp/2]) * ulp  with q = floor(s/ulp), r = s - q*ulp,
// unless r == ulp/2 exactly (a tie, which rounds on the parity of the running sum).  So for a
// group of 256 squares a wave computes every lane's q + [r > ulp/2] from the bit patterns (four
// squares per lane), adds them across the wave, and if no lane saw a tie, a square as large as
// acc, or anything non-finite, and A + Q < 2^24, then acc <- (A + Q) * ulp is exactly what 256
// sequential float additions would have produced.  Otherwise (a few dozen groups per 10 s signal:
// one per binade the sum climbs through, the tie-prone start, the rare later tie) the group is
// added one square after the other.  A staged block of 2048 squares is first tried as a whole.  One wave per (stream, array, channel) chain, the chains of
// a stream in one workgroup that stages blocks of 2048 frames into LDS, next block in flight.
constexpr int kSeqFrames = 2048;     // frames per staged block (generic channel counts)
constexpr int kSeqFramesStereo = 2048;   // stereo (4096 measured slower: a block with one tie costs twice as much)
constexpr int kSeqFramesWide = 256;  // 11 to 32 channels: smaller blocks, so that the 2C rows still fit LDS
constexpr int kSeqGroup = 256;       // squares a wave settles at once (4 per lane)
constexpr int kSeqMaxWaves = 16;

// sum over the wave's 64 lanes by DPP row shifts and row broadcasts (a few cycles each; the
// ds_bpermute ladder of __shfl_xor costs an LDS round trip per step)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);    // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);    // row_shr:8: lane 15 of a row = row total
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true);    // row_bcast:15 into rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true);    // row_bcast:31 into rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// inclusive prefix sum over the wave's lanes: Hillis-Steele inside each row of 16 (DPP row_shr, lanes
// without a source add 0), then the row totals (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2-3)
__device__ __forceinline__ uint32_t wave_prefix_u32(uint32_t x, int)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}

struct SeqTally {                 // one lane's view of some squares against acc's binade
    v2f q = {0.f, 0.f};           // sum of round(s / ulp): integers, exact in float while below 2^24
    v2f smax = {0.f, 0.f};        // largest s / ulp seen
    v2f rmax = {0.f, 0.f}, rmin = {0.f, 0.f};     // extreme rounding remainders: +-1/2 means a tie
    uint32_t any_bits = 0;
};

// 2^(150 - eb) as a float: s / ulp = s * that.  eb in [23, 254] (smaller sums take the slow path).
__device__ __forceinline__ float seq_scale(int eb) { return __uint_as_float((uint32_t)(277 - eb) << 23); }

__device__ __forceinline__ void seq_tally(SeqTally &t, const float4 &v4, float scale)
{
    // s / ulp by a power-of-two multiply (exact; a product that underflows is far below 1/2 and
    // rounds away either way), rounded to the nearest integer with the 2^23 trick - packed, two
    // squares per instruction, and no comparison inside the loop: what can go wrong is only
    // tracked (largest scaled square, extreme remainders) and judged once per block in seq_settle.
    const v2f sc = {scale, scale}, magic = {8388608.0f, 8388608.0f};
    const v2f in[2] = {v2f{v4.x, v4.y}, v2f{v4.z, v4.w}};
    t.any_bits |= __float_as_uint(v4.x) | __float_as_uint(v4.y) | __float_as_uint(v4.z) | __float_as_uint(v4.w);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const v2f scaled = in[i] * sc;
        const v2f whole = (scaled + magic) - magic;
        const v2f rem = scaled - whole;
        t.q = t.q + whole;
        t.smax = __builtin_elementwise_max(t.smax, scaled);
        t.rmax = __builtin_elementwise_max(t.rmax, rem);
        t.rmin = __builtin_elementwise_min(t.rmin, rem);
    }
}

// true: acc has been advanced over the tallied squares (or they were all +0 on a zero acc)
__device__ __forceinline__ bool seq_settle(const SeqTally &t, float &acc, int eb)
{
    const uint32_t ab = __float_as_uint(acc);
    const float lane_total = t.q.x + t.q.y;                // NaN if any square was NaN
    // a square within a factor 4 of acc (or inf) breaks the 2^23 trick's range; a remainder of
    // exactly +-1/2 is a tie (the rounding would depend on the running sum's parity)
    const bool bad = eb < 23 || eb >= 255 || !(lane_total < 16777216.0f) ||
                     !(fmaxf(t.smax.x, t.smax.y) < 4194304.0f) ||
                     fmaxf(t.rmax.x, t.rmax.y) == 0.5f || fminf(t.rmin.x, t.rmin.y) == -0.5f;
    const uint32_t grown = ((ab & 0x7fffffu) | 0x800000u) + wave_sum_u32(bad ? 0u : (uint32_t)lane_total);
    if (__ballot(bad) == 0 && grown < (1u << 24)) {
        acc = __uint_as_float(((uint32_t)eb << 23) | (grown & 0x7fffffu));
        return true;
    }
    return ab == 0 && __ballot(t.any_bits != 0) == 0;               // +0 + +0 ... : still +0
}

// Squares that seq_settle turned away because of TIES, still in integers.
// fl(acc + s) with acc = A * ulp rounds A + s/ulp to the nearest integer, and a tie (s/ulp = f + 1/2) to the EVEN
// one: the square then counts f + [(A_at + f) odd], where A_at is the running sum's mantissa when its turn comes.
// The 2^23 trick has already rounded the tie to even by itself (q = f for even f, remainder +1/2; q = f + 1 for odd f,
// remainder -1/2), so against that tally a tie is off by delta = [A_at odd] * (+1 for even f, -1 for odd f), and
// A_at's parity is that of A + (tallies before the element) + (the deltas before it) - only the PARITY of the
// mantissa the scan starts from matters.  A TieScan therefore carries both answers (start even / start odd) over
// any number of 256-square groups (four squares per lane, in order): the lanes form their tallies and a wave prefix
// sum per group, and the ties are walked in order with scalar bit operations - a handful per tie instead of 256
// dependent additions per group.  Valid while the sum stays inside its binade (the caller checks the final mantissa:
// the running sum never decreases) and every square is inside the trick's range (tie_scan_group returns false).
struct TieScan {
    uint32_t lane_total = 0;          // this lane's share of the round-to-even tallies (tie_scan_total adds the wave's)
    uint32_t parity = 0;              // parity of the tallies so far (a ballot per group: no prefix sum is needed, only parities)
    int delta0 = 0, delta1 = 0;       // what the ties add to it when the scan starts from an even / odd mantissa
    uint32_t flip0 = 0, flip1 = 0;    // (scalars, not arrays: a run-time index would send them to scratch memory)
    bool ties = false;
};

__device__ __forceinline__ bool tie_scan_group(TieScan &sc, const float4 &v4, float scale, int lane)
{
    // every lane: its four tallies, packed as in seq_tally (two squares per instruction)
    const v2f scl = {scale, scale}, magic = {8388608.0f, 8388608.0f};
    const v2f s01 = v2f{v4.x, v4.y} * scl, s23 = v2f{v4.z, v4.w} * scl;
    const v2f w01 = (s01 + magic) - magic, w23 = (s23 + magic) - magic;
    const v2f r01 = s01 - w01, r23 = s23 - w23;
    const v2f smax = __builtin_elementwise_max(s01, s23);
    const float lane_qf = (w01.x + w01.y) + (w23.x + w23.y);             // integers below 2^24: exact; NaN if a square was NaN
    // (max ignores a NaN operand, the sum does not; an infinite square fails the range test)
    const bool bad = !(fmaxf(smax.x, smax.y) < 4194304.0f) || !(lane_qf < 16777216.0f);
    const v2f rmax = __builtin_elementwise_max(r01, r23), rmin = __builtin_elementwise_min(r01, r23);
    const bool tied = fmaxf(rmax.x, rmax.y) == 0.5f || fminf(rmin.x, rmin.y) == -0.5f;
    if (__ballot(bad) != 0) return false;
    const uint32_t lane_q = (uint32_t)lane_qf;
    const uint64_t odd_lanes = __ballot((lane_q & 1u) != 0);
    uint64_t tmask = __ballot(tied);
    if (tmask != 0) {
        // the lanes that hold a tie describe their four squares: bit i: element i ties / its remainder is -1/2 /
        // parity of this lane's tallies before it (few lanes do; the branch is taken by the whole wave)
        const float wh[4] = {w01.x, w01.y, w23.x, w23.y}, rm[4] = {r01.x, r01.y, r23.x, r23.y};
        uint32_t info = 0, run = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            info |= (uint32_t)(rm[i] == 0.5f || rm[i] == -0.5f) << i;
            info |= (uint32_t)(rm[i] == -0.5f) << (4 + i);
            info |= (run & 1u) << (8 + i);
            run += (uint32_t)wh[i];
        }
        const uint32_t base = sc.parity;                       // parity of the tallies of the groups before
        sc.ties = true;
        while (tmask != 0) {
            const int l = __builtin_ctzll(tmask);
            tmask &= tmask - 1;
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)info, l);
            const uint32_t lanes_before = (uint32_t)__builtin_popcountll(odd_lanes & ((1ull << l) - 1ull));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if ((w >> i) & 1u) {
                    const uint32_t before = (base ^ lanes_before ^ (w >> (8 + i))) & 1u;
                    const int sign = ((w >> (4 + i)) & 1u) ? -1 : 1;
                    if ((before ^ sc.flip0) & 1u) { sc.delta0 += sign; sc.flip0 ^= 1u; }
                    if ((1u ^ before ^ sc.flip1) & 1u) { sc.delta1 += sign; sc.flip1 ^= 1u; }
                }
            }
        }
    }
    sc.parity ^= (uint32_t)__builtin_popcountll(odd_lanes) & 1u;
    sc.lane_total += lane_q;                                   // summed over the wave once, by tie_scan_total
    return true;
}

// the scan's tally: one wave reduction for all its groups (call once, after the last group)
__device__ __forceinline__ uint32_t tie_scan_total(const TieScan &sc)
{
    // a lane's share beyond 2^24 means the sum has left its binade long ago (and 64 such shares could wrap): saturate
    if (__ballot(sc.lane_total >= (1u << 24)) != 0) return 0xffffffffu;
    return wave_sum_u32(sc.lane_total);
}

// acc over the scanned squares, if the sum stayed inside its binade
__device__ __forceinline__ bool tie_scan_apply(const TieScan &sc, float &acc, int eb)
{
    const uint32_t mant = (__float_as_uint(acc) & 0x7fffffu) | 0x800000u;
    const uint32_t total = tie_scan_total(sc);
    if (total >= (1u << 24)) return false;
    const uint32_t grown = mant + total + (uint32_t)((mant & 1u) ? sc.delta1 : sc.delta0);
    if (grown >= (1u << 24)) return false;
    acc = __uint_as_float(((uint32_t)eb << 23) | (grown & 0x7fffffu));
    return true;
}

// one group of 256 squares
__device__ __forceinline__ bool seq_settle_ties(const float4 &v4, float &acc, int eb, int lane)
{
    if (eb < 23 || eb >= 255) return false;
    TieScan sc;
    return tie_scan_group(sc, v4, seq_scale(eb), lane) && tie_scan_apply(sc, acc, eb);
}

// the 256 dependent float32 additions of one group (every lane runs the chain on broadcast reads; the reads
// of the next 32 squares are in flight while the current 32 are added, so only the adds' own latency is paid)
__device__ __forceinline__ float seq_add_group(const float *sqs, float t)
{
    float4 v[2][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[0][u] = *(const float4 *)(sqs + 4 * u);
#pragma unroll
    for (int i = 0; i < kSeqGroup / 32; ++i) {
        if (i + 1 < kSeqGroup / 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[(i + 1) & 1][u] = *(const float4 *)(sqs + 32 * (i + 1) + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 q = v[i & 1][u];
            t = t + q.x; t = t + q.y; t = t + q.z; t = t + q.w;
        }
    }
    return t;
}

// PIPELINED: the wave has the CU to itself (block-parallel kernels) and pays 64 registers for the reads in
// flight; the one-workgroup-per-stream kernel keeps the short form (its other waves hide the LDS latency,
// and the registers would spill under its 256-thread bound)
template <bool PIPELINED>
__device__ __forceinline__ float seq_sum_group(const float *row, float acc, int lane)
{
    const int eb = (int)(__float_as_uint(acc) >> 23);   // acc is a sum of squares: sign 0 (a NaN may set it: eb > 255)
    SeqTally t;
    const float4 mine = *(const float4 *)(row + 4 * lane);
    seq_tally(t, mine, seq_scale(eb));
    if (seq_settle(t, acc, eb)) return acc;
    if (seq_settle_ties(mine, acc, eb, lane)) return acc;
    // one after the other; every lane does the same adds on the same (broadcast) LDS words
    if constexpr (PIPELINED) {
        return seq_add_group(row, acc);
    } else {
#pragma unroll 1
        for (int i = 0; i < kSeqGroup; i += 16) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const float4 *)(row + i + 4 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = acc + v[u].x;
                acc = acc + v[u].y;
                acc = acc + v[u].z;
                acc = acc + v[u].w;
            }
        }
        return acc;
    }
}

// A whole staged block at once when nothing in it needs care (one reduction per 2048 squares),
// else group by group.
template <int FRAMES, bool PIPELINED = false>
__device__ __forceinline__ float seq_sum_block(const float *row, float acc, int lane)
{
    const int eb = (int)(__float_as_uint(acc) >> 23);
    SeqTally t;
    const float scale = seq_scale(eb);
#pragma unroll
    for (int g = 0; g < FRAMES; g += kSeqGroup) seq_tally(t, *(const float4 *)(row + g + 4 * lane), scale);
    if (seq_settle(t, acc, eb)) return acc;
    // ties (audio that came from integers is full of them): the whole block in integers - when ties are what
    // turned the block away (a sum about to leave its binade goes group by group at once)
    const bool in_range = (t.q.x + t.q.y) < 16777216.0f && fmaxf(t.smax.x, t.smax.y) < 4194304.0f;
    const bool tie_seen = fmaxf(t.rmax.x, t.rmax.y) == 0.5f || fminf(t.rmin.x, t.rmin.y) == -0.5f;
    if (eb >= 23 && eb < 255 && __ballot(!in_range) == 0 && __ballot(tie_seen) != 0) {
        TieScan sc;
        bool ok = true;
#pragma unroll 1
        for (int g = 0; g < FRAMES && ok; g += kSeqGroup) ok = tie_scan_group(sc, *(const float4 *)(row + g + 4 * lane), scale, lane);
        if (ok && tie_scan_apply(sc, acc, eb)) return acc;
    }
#pragma unroll 1
    for (int g = 0; g < FRAMES; g += kSeqGroup) acc = seq_sum_group<PIPELINED>(row + g, acc, lane);
    return acc;
}

// blockDim.x = 64 * W, W = min(2C, 16) waves; wave w owns chains w, w + W, ...
// STEREO: C == 2, branch-free vector staging (range-checked buffer loads: frames past the end
// read 0 and add +0, exact); MONO: x has one channel, fanned out.
// (Round 6 tried this kernel's loads on consecutive bytes per wave access, as the block-parallel staging now has them: 256 mono streams
//  1.350 -> 1.463 ms, stereo 1.401 -> 1.419 - and the mere presence of both mappings behind a uniform branch cost 19 % (569 -> 677 us):
//  the kernel is as it was.)
template <bool STEREO, bool MONO, int BF = (STEREO ? kSeqFramesStereo : kSeqFrames)>
__global__ __launch_bounds__(64 * kSeqMaxWaves) void epilogue_rms_seq_kernel(const EArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sq[];      // [2C][BF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves = blockDim.x >> 6;
    const int C = STEREO ? 2 : a.C, Cx = a.Cx;
    // (stereo, seq_split: the x chains and the y chains of a stream are independent recurrences over different arrays - a workgroup
    //  each doubles the loads in flight when there are fewer streams than two per CU: 256 x 10 s 0.57 -> 0.4x ms)
    const bool split = STEREO && a.seq_split != 0;
    const int arr = split ? (int)(blockIdx.x & 1) : 0;               // split: 0 = x's chains, 1 = y's
    const int chains = split ? C : 2 * C;
    const int64_t b = split ? blockIdx.x >> 1 : blockIdx.x;
    const int64_t n = a.n;
    const float *__restrict__ xs = a.x + b * a.n * Cx;
    const float *__restrict__ ys = a.y + b * a.n * C;
    constexpr int PER = STEREO ? BF / 1024 : 1;            // stereo: rounds of 4 frames per thread (256 threads)
    v2f xr[PER][4], yr[PER][4];                            // this thread's frames of the next block
    // descriptors are re-based at every block (offsets stay small: streams beyond 2 GiB are fine)
    auto fetch = [&](int64_t f0) {
        if constexpr (STEREO) {
            const v4i rx = make_rsrc(xs + f0 * Cx, (n - f0) * Cx * 4);
            const v4i ry = make_rsrc(ys + f0 * 2, (n - f0) * 8);
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                if (a.wide) {                                  // streams start 16-byte aligned: two frames per access
#pragma unroll
                    for (int k = 0; k < 4; k += 2) {
                        const int fr = u * (BF / PER) + 4 * tid + k;
                        if (!split || arr == 1) {
                            const v4f ty = buf_load4(ry, fr * 8, 0, 0);
                            yr[u][k] = v2f{ty.x, ty.y}; yr[u][k + 1] = v2f{ty.z, ty.w};
                        }
                        if (split && arr == 1) continue;
                        if constexpr (MONO) {
                            const v2f v = buf_load2(rx, fr * 4, 0, 0);
                            xr[u][k] = v2f{v.x, v.x}; xr[u][k + 1] = v2f{v.y, v.y};
                        } else {
                            const v4f tx = buf_load4(rx, fr * 8, 0, 0);
                            xr[u][k] = v2f{tx.x, tx.y}; xr[u][k + 1] = v2f{tx.z, tx.w};
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int fr = u * (BF / PER) + 4 * tid + k;
                    if (!split || arr == 1) yr[u][k] = buf_load2(ry, fr * 8, 0, 0);
                    if (split && arr == 1) continue;
                    if constexpr (MONO) {
                        const float v = buf_load1(rx, fr * 4, 0, 0);
                        xr[u][k] = v2f{v, v};
                    } else {
                        xr[u][k] = buf_load2(rx, fr * 8, 0, 0);
                    }
                }
            }
        }
    };
    auto stage = [&](int64_t f0) {
        if constexpr (STEREO) {
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                float4 *dst = (float4 *)(sq + u * (BF / PER) + 4 * tid);
                if (split) {                                   // this workgroup's array only: its two chains are rows 0 and 1
                    const v2f (&vr)[4] = arr ? yr[u] : xr[u];
                    dst[0 * BF / 4] = make_float4(vr[0].x * vr[0].x, vr[1].x * vr[1].x, vr[2].x * vr[2].x, vr[3].x * vr[3].x);
                    dst[1 * BF / 4] = make_float4(vr[0].y * vr[0].y, vr[1].y * vr[1].y, vr[2].y * vr[2].y, vr[3].y * vr[3].y);
                    continue;
                }
                dst[0 * BF / 4] = make_float4(xr[u][0].x * xr[u][0].x, xr[u][1].x * xr[u][1].x,
                                              xr[u][2].x * xr[u][2].x, xr[u][3].x * xr[u][3].x);
                dst[1 * BF / 4] = make_float4(xr[u][0].y * xr[u][0].y, xr[u][1].y * xr[u][1].y,
                                              xr[u][2].y * xr[u][2].y, xr[u][3].y * xr[u][3].y);
                dst[2 * BF / 4] = make_float4(yr[u][0].x * yr[u][0].x, yr[u][1].x * yr[u][1].x,
                                              yr[u][2].x * yr[u][2].x, yr[u][3].x * yr[u][3].x);
                dst[3 * BF / 4] = make_float4(yr[u][0].y * yr[u][0].y, yr[u][1].y * yr[u][1].y,
                                              yr[u][2].y * yr[u][2].y, yr[u][3].y * yr[u][3].y);
            }
        } else {
            const v4i rx = make_rsrc(xs + f0 * Cx, (n - f0) * Cx * 4);
            const v4i ry = make_rsrc(ys + f0 * C, (n - f0) * C * 4);
            for (int e = tid; e < BF * C; e += blockDim.x) {
                const int f = e / C, c = e - f * C;
                const float xv = buf_load1(rx, (f * Cx + c % Cx) * 4, 0, 0);
                const float yv = buf_load1(ry, (f * C + c) * 4, 0, 0);
                sq[c * BF + f] = xv * xv;                    // frames past the end read 0
                sq[(C + c) * BF + f] = yv * yv;
            }
        }
    };
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                  // chains wave, wave + W, ... (2C <= 64 => at most 4 each)
    fetch(0);
    for (int64_t f0 = 0; f0 < n; f0 += BF) {
        __syncthreads();                                   // every wave is done with the previous block
        stage(f0);
        __syncthreads();
        if (f0 + BF < n) fetch(f0 + BF);                   // (two blocks in flight: no faster in round 1; with the per-array split of round 4 slower - 256 x 10 s 0.51 -> 0.78 ms)
        int slot = 0;
        for (int ch = wave; ch < chains; ch += waves, ++slot)
            acc[slot] = seq_sum_block<BF>(sq + ch * BF, acc[slot], lane);
    }
    int slot = 0;
    for (int ch = wave; ch < chains; ch += waves, ++slot)
        if (lane == 0) a.partials[b * a.rows * (2 * C) + arr * C + ch] = (double)acc[slot];      // x's chains, then y's
}

// ---- the same sums, parallel over the stream (stereo tables) -----------------
// The sequential kernel above walks a 10 s signal block after block in ONE workgroup (437 us, 17 GB/s).
// The recurrence itself cannot be split, but what it needs per block can: whether a run of squares
// can be settled in integers depends only on the BINADE of the running sum when the run starts, and
// that is predictable from a float64 prefix of the block sums.  So:
//   rms_par_sum     one workgroup per block of 2048 frames: float64 sum of each chain's squares;
//   rms_par_tally   one workgroup per block: predicts the binade e at the block's start from the
//                   prefix of those sums and, where the sum is about to leave the binade, the group
//                   g* of 256 squares in which it will; computes the integer tallies per group against
//                   ulp(e) and against ulp(e + 1) with their tie / range flags, and for a block whose
//                   only trouble is TIES its tally plus the two corrections for an even / odd mantissa
//                   at its start (TieScan).  An extra workgroup sums block 0, whose starting sum is
//                   known - zero -, for good with the sequential code;
//   rms_par_stitch  one wave per chain walks the blocks in order.  Runs of blocks whose prediction
//                   holds are accepted 64 at a time (a wave prefix sum of their tallies finds the
//                   first one that does not fit); ties-only blocks take a few scalar operations each
//                   (the mantissa's low bit picks the correction); a block that crosses a binade goes
//                   group by group: integer adds, an integer tie scan, or - in the group where the
//                   sum really leaves its binade - the 256 dependent float additions (its squares
//                   prefetched when the kernel starts); what defeats the prediction is re-read.
// Every accepted step is exactly what the dependent float32 additions produce, every prediction
// is verified against the actual running sum, so the result is the sequential kernel's (NumPy's,
// utils/dsp.py:87-109) bit for bit.
constexpr int kParFrames = 2048;
constexpr int kParThreads = 256;
constexpr int kParGroups = kParFrames / kSeqGroup;       // 8
constexpr int kParSlots = 24;                            // groups prefetched per chain: binade crossings (a sum crosses each binade once) and ties
constexpr uint32_t kParZero = 1u << 9, kParBad = 1u << 10, kParTies = 1u << 11, kParHint = 1u << 12;

struct ParRec {                       // 16 bytes per (chain, block)
    uint32_t tag;                     // bits 0-8 predicted biased exponent e; flags; bits 16-18 g* (with kParHint)
    uint32_t qtot;                    // the whole block's tally against ulp(e) (valid unless kParBad)
    uint32_t bad;                     // bit g: group g cannot be settled against ulp(e); bit 8 + g: against ulp(e + 1)
    uint32_t need;                    // bit g: group g is expected to take the 256 dependent additions (g*, or bad in its binade);
                                      // with kParTies, bits 8-19 / 20-31: what the block's ties add to qtot when the sum's mantissa is
                                      // even / odd as the block starts (signed 12 bits each, TieScan)
};
struct ParGrp { uint32_t qe[kParGroups], qf[kParGroups]; };      // per-group tallies against ulp(e) and ulp(e + 1)

struct RArgs {
    const float *__restrict__ x;
    const float *__restrict__ y;
    int64_t n;
    int32_t Cx;                      // 1 (mono fanned out) or 2
    int32_t nblocks;
    double *__restrict__ blk_sum;    // [batch][4][nblocks]
    ParRec *__restrict__ rec;        // [batch][4][nblocks]
    ParGrp *__restrict__ grp;        // [batch][4][nblocks]
    float *__restrict__ first;       // [batch][4]: the sum after block 0
    int32_t prefixed;                // blk_sum already holds EXCLUSIVE prefix sums (rms_par_prefix_kernel ran)
    int32_t wide;                    // every stream of x and y starts 16-byte aligned: the staging loads are whole 16-byte accesses
    double *__restrict__ partials;   // [batch][4]: the sums, as the sequential kernel writes them
    // signals of more than two channels: a "stream" of these kernels is a CHANNEL PAIR of a stream - index b = stream * pairs + pair
    // everywhere (blk_sum, rec, grp, first: [batch * pairs][4][...]); the loaders read the pair's 8 bytes of every C-channel frame
    // and the stitch kernel writes the sums where the sequential kernel would: partials[stream][x_0 .. x_C-1, y_0 .. y_C-1]
    int32_t C;                       // channels per frame (2: stereo, the original form)
    int32_t pairs;                   // C / 2
};

// the pair's first sample of frame f0 of its stream, and the bytes from there to the stream's end
__device__ __forceinline__ const float *par_pair_base(const RArgs &a, const float *arr, int64_t b, int64_t f0, int ch, int64_t *bytes)
{
    const int64_t stream = b / a.pairs;
    const int pair = (int)(b - stream * a.pairs);
    *bytes = (a.n - f0) * (int64_t)a.C * 4 - (2 * pair + ch) * 4;
    return arr + (stream * a.n + f0) * a.C + 2 * pair + ch;
}

// squares of block `blk` of stream b into sq[4][2048] (chains: x ch0, x ch1, y ch0, y ch1); frames
// past the end of the stream read 0 and add +0, exact
template <bool MONO>
__device__ __forceinline__ void par_stage(const RArgs &a, int64_t b, int blk, float *sq, int tid)
{
    const int64_t f0 = (int64_t)blk * kParFrames;
    if (!MONO && a.C > 2) {
        // a channel pair of a wider signal: 8 bytes of every frame, frames C * 4 bytes apart
        int64_t xb, yb;
        const float *xs = par_pair_base(a, a.x, b, f0, 0, &xb), *ys = par_pair_base(a, a.y, b, f0, 0, &yb);
        const v4i rx = make_rsrc(xs, xb), ry = make_rsrc(ys, yb);
        const int fb = a.C * 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int fr = 4 * tid + 1024 * u;
            v2f xf[4], yf[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { xf[k] = buf_load2(rx, (fr + k) * fb, 0, 0); yf[k] = buf_load2(ry, (fr + k) * fb, 0, 0); }
            float4 *dst = (float4 *)(sq + fr);
            dst[0 * kParFrames / 4] = make_float4(xf[0].x * xf[0].x, xf[1].x * xf[1].x, xf[2].x * xf[2].x, xf[3].x * xf[3].x);
            dst[1 * kParFrames / 4] = make_float4(xf[0].y * xf[0].y, xf[1].y * xf[1].y, xf[2].y * xf[2].y, xf[3].y * xf[3].y);
            dst[2 * kParFrames / 4] = make_float4(yf[0].x * yf[0].x, yf[1].x * yf[1].x, yf[2].x * yf[2].x, yf[3].x * yf[3].x);
            dst[3 * kParFrames / 4] = make_float4(yf[0].y * yf[0].y, yf[1].y * yf[1].y, yf[2].y * yf[2].y, yf[3].y * yf[3].y);
        }
        return;
    }
    const float *xs = a.x + (b * a.n + f0) * (MONO ? 1 : 2);
    const float *ys = a.y + (b * a.n + f0) * 2;
    const v4i rx = make_rsrc(xs, (a.n - f0) * (MONO ? 4 : 8));
    const v4i ry = make_rsrc(ys, (a.n - f0) * 8);
    // 8 bytes per access where a stream's first sample is only 8-byte aligned (n odd, or a misaligned base): a.wide says when whole
    // 16-byte accesses are safe - they move the same bytes at 1.4-1.8x the rate (MI355X_MICROARCH.md: 8-byte accesses 0.54-0.70x)
    if (a.wide == 2) {
        // whole 16-byte accesses, a wave's 64 lanes on 1 KB of CONSECUTIVE bytes (8 cache lines per access; a lane taking 32 consecutive
        // bytes as two accesses made each of them 16 half-used lines - round 6): a lane holds frame pairs 2*tid + 512*k of y (and of a
        // stereo x), frames 4*tid + 1024*u .. + 3 of a mono x
        v4f yq[4], xq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) yq[k] = buf_load4(ry, tid * 16 + 4096 * k, 0, 0);
        if constexpr (MONO) {
#pragma unroll
            for (int u = 0; u < 2; ++u) xq[u] = buf_load4(rx, tid * 16 + 4096 * u, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) xq[k] = buf_load4(rx, tid * 16 + 4096 * k, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float2 *dst = (float2 *)(sq + 2 * tid + 512 * k);
            dst[2 * kParFrames / 2] = make_float2(yq[k].x * yq[k].x, yq[k].z * yq[k].z);
            dst[3 * kParFrames / 2] = make_float2(yq[k].y * yq[k].y, yq[k].w * yq[k].w);
            if constexpr (!MONO) {
                dst[0 * kParFrames / 2] = make_float2(xq[k].x * xq[k].x, xq[k].z * xq[k].z);
                dst[1 * kParFrames / 2] = make_float2(xq[k].y * xq[k].y, xq[k].w * xq[k].w);
            }
        }
        if constexpr (MONO) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float4 m2 = make_float4(xq[u].x * xq[u].x, xq[u].y * xq[u].y, xq[u].z * xq[u].z, xq[u].w * xq[u].w);
                float4 *dst = (float4 *)(sq + 4 * tid + 1024 * u);
                dst[0 * kParFrames / 4] = m2;
                dst[1 * kParFrames / 4] = m2;
            }
        }
        return;
    }
    v4f yv[2][2], xv[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {                         // frames 4*tid + 1024*u + {0..3}
        const int fr = 4 * tid + 1024 * u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (a.wide) {
                yv[u][h] = buf_load4(ry, (fr + 2 * h) * 8, 0, 0);
                if constexpr (!MONO) { xv[u][h] = buf_load4(rx, (fr + 2 * h) * 8, 0, 0); continue; }
            } else {
                const v2f y0 = buf_load2(ry, (fr + 2 * h) * 8, 0, 0), y1 = buf_load2(ry, (fr + 2 * h + 1) * 8, 0, 0);
                yv[u][h] = v4f{y0.x, y0.y, y1.x, y1.y};
            }
            if constexpr (MONO) {
                const float m0 = buf_load1(rx, (fr + 2 * h) * 4, 0, 0), m1 = buf_load1(rx, (fr + 2 * h + 1) * 4, 0, 0);
                xv[u][h] = v4f{m0, m0, m1, m1};
            } else {
                const v2f x0 = buf_load2(rx, (fr + 2 * h) * 8, 0, 0), x1 = buf_load2(rx, (fr + 2 * h + 1) * 8, 0, 0);
                xv[u][h] = v4f{x0.x, x0.y, x1.x, x1.y};
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float4 *dst = (float4 *)(sq + 4 * tid + 1024 * u);
        const v4f p = xv[u][0], q = xv[u][1], r = yv[u][0], t = yv[u][1];
        dst[0 * kParFrames / 4] = make_float4(p.x * p.x, p.z * p.z, q.x * q.x, q.z * q.z);
        dst[1 * kParFrames / 4] = make_float4(p.y * p.y, p.w * p.w, q.y * q.y, q.w * q.w);
        dst[2 * kParFrames / 4] = make_float4(r.x * r.x, r.z * r.z, t.x * t.x, t.z * t.z);
        dst[3 * kParFrames / 4] = make_float4(r.y * r.y, r.w * r.w, t.y * t.y, t.w * t.w);
    }
}

// Which (block, stream-or-pair) a workgroup of the sum / tally kernels takes.  Stereo: blockIdx.x = block, blockIdx.y = stream.  Wider
// signals: the P channel pairs of a block read the SAME cache lines (8 bytes each of every C-channel frame), so they must meet in one L2:
// blockIdx.y = stream, and blockIdx.x runs XCD by XCD (blocks b, b + 8, ... share an L2; MI355X_MICROARCH.md) with the pair fastest -
// the P workgroups of a block are neighbours in their XCD's queue, and the lines come from HBM once, not P times.  grid.x =
// ceil(nblocks / 8) * 8 * P; returns false for the padding.
// PW: channel pairs per workgroup (2: a channel QUAD - 16 bytes of every frame per access, par_stage_quad; *b = its first pair).
template <int PW = 1>
__device__ __forceinline__ bool par_unit(const RArgs &a, int *blk, int64_t *b)
{
    if (a.pairs <= 1) { *blk = (int)blockIdx.x; *b = blockIdx.y; return true; }
    const unsigned per_block = (unsigned)a.pairs / PW;
    const unsigned id = blockIdx.x, xcd = id & 7u, j = id >> 3;
    const unsigned unit = j % per_block;
    *blk = (int)((j / per_block) * 8u + xcd);
    *b = (int64_t)blockIdx.y * a.pairs + unit * PW;
    return *blk < a.nblocks;
}

// a channel QUAD of a wider signal (pairs b0 and b0 + 1 of its stream): whole 16-byte accesses, 512 threads x 4 frames, squares into
// sq[8][2048]: rows 0-3 the first pair's chains (x ch0, x ch1, y ch0, y ch1), rows 4-7 the second pair's
__device__ __forceinline__ void par_stage_quad(const RArgs &a, int64_t b0, int blk, float *sq, int tid)
{
    const int64_t f0 = (int64_t)blk * kParFrames;
    int64_t xb, yb;
    const float *xs = par_pair_base(a, a.x, b0, f0, 0, &xb), *ys = par_pair_base(a, a.y, b0, f0, 0, &yb);
    const v4i rx = make_rsrc(xs, xb), ry = make_rsrc(ys, yb);
    // (a lane takes frames tid, tid + 512, ...: a wave's access is 64 CONSECUTIVE frames - 16 cache lines for a signal of 8 channels.
    //  Four consecutive frames per lane - one 128-byte line each, 16 bytes of it per access - made every access 64 lines: the tally
    //  of cfg5's pool ran 441 us for 0.98 GB, all of it from HBM once - profiles/r06_f1_c8_kernels.txt)
    const int fb = a.C * 4;
    v4f xq[4], yq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { xq[k] = buf_load4(rx, (tid + 512 * k) * fb, 0, 0); yq[k] = buf_load4(ry, (tid + 512 * k) * fb, 0, 0); }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float *dst = sq + tid + 512 * k;
        dst[0 * kParFrames] = xq[k].x * xq[k].x; dst[1 * kParFrames] = xq[k].y * xq[k].y;
        dst[2 * kParFrames] = yq[k].x * yq[k].x; dst[3 * kParFrames] = yq[k].y * yq[k].y;
        dst[4 * kParFrames] = xq[k].z * xq[k].z; dst[5 * kParFrames] = xq[k].w * xq[k].w;
        dst[6 * kParFrames] = yq[k].z * yq[k].z; dst[7 * kParFrames] = yq[k].w * yq[k].w;
    }
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_xor(v, sh);
    return v;
}

// the same total by DPP row shifts and row broadcasts (no LDS round trips: the __shfl_xor ladder is twelve ds_bpermute per sum), in
// another order of additions - for sums that are PREDICTIONS (the tally kernel's group sums: every step they lead to is verified)
__device__ __forceinline__ double wave_sum_f64_dpp(double v)
{
    auto shifted = [](double x, auto ctrl, auto rows) {
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, x);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)bits, decltype(ctrl)::value, decltype(rows)::value, 0xf, true);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(bits >> 32), decltype(ctrl)::value, decltype(rows)::value, 0xf, true);
        return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    };
    using std::integral_constant;
    v += shifted(v, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});      // row_shr:1
    v += shifted(v, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});      // row_shr:2
    v += shifted(v, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});      // row_shr:4
    v += shifted(v, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});      // row_shr:8: lane 15 of a row = row total
    v += shifted(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});      // row_bcast:15 into rows 1, 3
    v += shifted(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});      // row_bcast:31 into rows 2, 3
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <bool MONO, int PW = 1>
__global__ __launch_bounds__(kParThreads * PW) void rms_par_sum_kernel(const RArgs a)
{
    __shared__ __attribute__((aligned(16))) float sq[PW * 4 * kParFrames];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, chain = wave & 3;
    int blk;
    int64_t b;
    if (!par_unit<PW>(a, &blk, &b)) return;
    if constexpr (PW == 2) par_stage_quad(a, b, blk, sq, tid);
    else par_stage<MONO>(a, b, blk, sq, tid);
    b += wave >> 2;                                        // this wave's pair
    __syncthreads();
    const float *row = sq + wave * kParFrames;
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < kParFrames; g += 256) {
        const float4 v = *(const float4 *)(row + g + 4 * lane);
        s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    }
    s = wave_sum_f64(s);
    if (lane == 0) a.blk_sum[(b * 4 + chain) * a.nblocks + blk] = s;
}

// Long streams (more than kParPrefixBlocks blocks): the block sums are turned into exclusive prefix sums in
// place by one workgroup per chain - every block adding up its own predecessors is quadratic, 1.3 s for a
// 300 M-frame stream.  Each thread scans a contiguous stretch; the stretch totals are scanned by thread 0.
constexpr int kParPrefixBlocks = 2048;
// Streams beyond 4096 blocks (8.4 M frames) keep the sequential kernel: past ~2^24 frames NumPy's float32
// running sum stops growing (every square is below half an ulp) while the float64 prediction keeps climbing,
// and a stitch that mispredicts every block costs far more than it saves (300 M frames: 1.47 s against 0.2 s).
constexpr int kParMaxBlocks = 4096;

__global__ __launch_bounds__(kParThreads) void rms_par_prefix_kernel(const RArgs a)
{
    __shared__ double totals[kParThreads];
    double *sums = a.blk_sum + (int64_t)blockIdx.x * a.nblocks;
    const int per = (a.nblocks + kParThreads - 1) / kParThreads;
    const int lo = min((int)threadIdx.x * per, a.nblocks), hi = min(lo + per, a.nblocks);
    double t = 0.0;
    for (int j = lo; j < hi; ++j) t += sums[j];
    totals[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double run = 0.0;
        for (int k = 0; k < kParThreads; ++k) { const double v = totals[k]; totals[k] = run; run += v; }
    }
    __syncthreads();
    double run = totals[threadIdx.x];
    for (int j = lo; j < hi; ++j) { const double v = sums[j]; sums[j] = run; run += v; }
}

// sum of round(s / ulp) over the groups [g0, g1) of a staged row, with the flags of seq_settle
__device__ __forceinline__ uint32_t par_tally_groups(const float *row, int g0, int g1, int eb, int lane, bool *bad, bool *zero)
{
    SeqTally t;
    const bool eb_ok = eb >= 23 && eb < 255;
    const float scale = seq_scale(eb_ok ? eb : 127);
    for (int g = g0; g < g1; ++g) seq_tally(t, *(const float4 *)(row + g * kSeqGroup + 4 * lane), scale);
    const float lane_total = t.q.x + t.q.y;
    const bool lane_bad = !eb_ok || !(lane_total < 16777216.0f) || !(fmaxf(t.smax.x, t.smax.y) < 4194304.0f) ||
                          fmaxf(t.rmax.x, t.rmax.y) == 0.5f || fminf(t.rmin.x, t.rmin.y) == -0.5f;
    const bool any_bad = __ballot(lane_bad) != 0;
    const uint32_t q = wave_sum_u32(lane_bad ? 0u : (uint32_t)lane_total);
    *bad = any_bad || q >= (1u << 24);
    *zero = __ballot(t.any_bits != 0) == 0;
    return q;
}

template <bool MONO, int PW = 1>
__global__ __launch_bounds__(kParThreads * PW) void rms_par_tally_kernel(const RArgs a)
{
    __shared__ __attribute__((aligned(16))) float sq[PW * 4 * kParFrames];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, chain = wave & 3;
    int unit;
    int64_t b;
    if (!par_unit<PW>(a, &unit, &b)) return;
    const int64_t b0 = b;
    b += wave >> 2;                                        // this wave's pair
    if (unit == a.nblocks - 1) {
        // the extra workgroup: the chains' START - block 0 by the recurrence itself, from +0 (its first groups
        // are all ties and binade crossings) - runs beside the tallies of the other blocks instead of holding
        // up the block sums that they wait for
        if constexpr (PW == 2) par_stage_quad(a, b0, 0, sq, tid);
        else par_stage<MONO>(a, b, 0, sq, tid);
        __syncthreads();
        const float acc = seq_sum_block<kParFrames, true>(sq + wave * kParFrames, 0.0f, lane);
        if (lane == 0) a.first[b * 4 + chain] = acc;
        return;
    }
    const int blk = unit + 1;
    // where the running sum stands when this block starts, to within float64 rounding: only its
    // binade matters, and a wrong guess merely sends the block down the sequential path
    // (its loads are issued BEFORE the block's own: the two round trips overlap - the kernel is bound by latency, five workgroups per CU)
    const double *sums = a.blk_sum + (b * 4 + chain) * a.nblocks;
    double pre = 0.0;
    if (a.prefixed) {
        pre = sums[blk];
    } else {                                               // short streams: every block adds up its predecessors itself
        for (int j = lane; j < blk; j += 64) pre += sums[j];
    }
    if constexpr (PW == 2) par_stage_quad(a, b0, blk, sq, tid);
    else par_stage<MONO>(a, b, blk, sq, tid);
    if (!a.prefixed) pre = wave_sum_f64_dpp(pre);
    const int eb = (int)(__float_as_uint((float)pre) >> 23);
    __syncthreads();
    const float *row = sq + wave * kParFrames;
    // group sums (float64) locate the group g* in which the sum is expected to reach the next power of two
    const double next = (double)__uint_as_float((uint32_t)min(eb + 1, 254) << 23);       // 2^(e + 1 - 127)
    int gstar = kParGroups;
    double run = pre;
#pragma unroll
    for (int g = 0; g < kParGroups; ++g) {
        const float4 v = *(const float4 *)(row + g * kSeqGroup + 4 * lane);
        const double after = run + wave_sum_f64_dpp((double)v.x + (double)v.y + (double)v.z + (double)v.w);
        if (run < next && after >= next && gstar == kParGroups) gstar = g;
        run = after;
    }
    // per group: the tally against ulp(e) and against ulp(e + 1) (the stitch uses whichever binade the
    // running sum is really in), and whether each can be trusted.  The tallies against ulp(e + 1) are only computed where the sum
    // may reach the next binade inside this block - a crossing is foreseen, or the block ends within a thousandth of it (the float32
    // recurrence drifts from these float64 sums by ~1e-4): elsewhere they are flagged untrustworthy, and a block the stitch finds
    // in the next binade after all takes its groups' additions (exact either way; a third of this kernel's work for nine blocks in ten)
    const bool want_f = gstar < kParGroups || run >= next * 0.999;
    // Nine blocks in ten are PLAIN: no tie, no square as large as the sum, no crossing in sight.  Those are tallied as a whole - one
    // accumulation over the eight groups, one set of flags, one reduction - and the stitch accepts them by their total alone; their
    // per-group tallies are never computed and are flagged untrustworthy (a plain block that the stitch has to open after all - the
    // float32 sum left its binade where these float64 sums saw no crossing - takes its groups' additions: exact, and rare).
    bool whole_bad = false, whole_zero = true;
    const uint32_t q_whole = par_tally_groups(row, 0, kParGroups, eb, lane, &whole_bad, &whole_zero);
    const bool per_group = whole_bad || want_f;
    uint32_t bad_bits = 0xffffu, qtot = q_whole, mine_e = 0, mine_f = 0;
    bool all_zero = whole_zero;
    if (per_group) {
        bad_bits = want_f ? 0u : 0xff00u; qtot = 0; all_zero = true;
#pragma unroll
        for (int g = 0; g < kParGroups; ++g) {
            bool be = false, bf = false, ze = true, zf = true;
            const uint32_t qe = par_tally_groups(row, g, g + 1, eb, lane, &be, &ze);
            const uint32_t qf = want_f ? par_tally_groups(row, g, g + 1, eb + 1, lane, &bf, &zf) : 0u;
            if (be) bad_bits |= 1u << g;
            if (bf) bad_bits |= 1u << (8 + g);
            all_zero &= ze;
            qtot += qe;
            if (lane == g) mine_e = qe;
            if (lane == 8 + g) mine_f = qf;
        }
    }
    const uint32_t real_bad = per_group ? bad_bits : 0u;           // what the block's own data flagged (the stitch's prefetch list, kParBad)
    // a block whose only trouble is ties: its round-to-even tally and the two corrections (start even / odd), so that the
    // stitch accepts it with three scalar operations (audio that came from integers has ties in most blocks)
    TieScan sc;
    uint32_t scan_total = 0;
    bool scan_ok = eb >= 23 && eb < 255 && (real_bad & 0xffu) != 0;
    if (scan_ok) {
        const float scale = seq_scale(eb);
#pragma unroll 1
        for (int g = 0; g < kParGroups && scan_ok; ++g) scan_ok = tie_scan_group(sc, *(const float4 *)(row + g * kSeqGroup + 4 * lane), scale, lane);
        if (scan_ok) scan_total = tie_scan_total(sc);
        scan_ok = scan_ok && sc.ties && scan_total < (1u << 24) && abs(sc.delta0) < 2048 && abs(sc.delta1) < 2048;
    }
    const int64_t at = (b * 4 + chain) * a.nblocks + blk;
    if (lane < kParGroups) a.grp[at].qe[lane] = mine_e;
    else if (lane < 2 * kParGroups) a.grp[at].qf[lane - kParGroups] = mine_f;
    if (lane == 0) {
        ParRec r;
        r.tag = (uint32_t)(eb & 0x1ff);
        if (all_zero) r.tag |= kParZero;
        if (scan_ok) { r.tag |= kParTies; qtot = scan_total; }
        else if ((real_bad & 0xffu) != 0 || qtot >= (1u << 24)) r.tag |= kParBad;
        if (gstar < kParGroups) r.tag |= kParHint | ((uint32_t)gstar << 16);
        r.qtot = qtot; r.bad = bad_bits;
        // the groups the stitch will most likely have to add one by one: g* itself, before it the groups
        // that cannot be settled against ulp(e), after it those that cannot against ulp(e + 1)
        r.need = gstar < kParGroups ? ((1u << gstar) | (bad_bits & ((1u << gstar) - 1u)) | ((bad_bits >> 8) & 0xffu & ~((2u << gstar) - 1u)))
                                    : (scan_ok ? 0u : (real_bad & 0xffu));      // a ties-only block is settled without its squares
        if (scan_ok) r.need |= (((uint32_t)sc.delta0 & 0xfffu) << 8) | (((uint32_t)sc.delta1 & 0xfffu) << 20);
        a.rec[at] = r;
    }
}

// four consecutive samples of one chain for this lane: frames f0 + 4*lane .. + 3 of stream b (0 past the end)
template <bool MONO>
__device__ __forceinline__ float4 par_fetch4(const RArgs &a, int64_t b, int chain, int64_t f0, int lane)
{
    const bool from_x = chain < 2;
    const int ch = chain & 1;
    const int fr = 4 * lane;
    if (from_x && MONO) {
        const v4i rs = make_rsrc(a.x + b * a.n + f0, (a.n - f0) * 4);
        return make_float4(buf_load1(rs, (fr + 0) * 4, 0, 0), buf_load1(rs, (fr + 1) * 4, 0, 0),
                           buf_load1(rs, (fr + 2) * 4, 0, 0), buf_load1(rs, (fr + 3) * 4, 0, 0));
    }
    int64_t bytes;
    const float *src = par_pair_base(a, from_x ? a.x : a.y, b, f0, ch, &bytes);       // (stereo: C = 2, one pair - the stream itself)
    const v4i rs = make_rsrc(src, bytes);
    const int fb = a.C * 4;
    return make_float4(buf_load1(rs, (fr + 0) * fb, 0, 0), buf_load1(rs, (fr + 1) * fb, 0, 0),
                       buf_load1(rs, (fr + 2) * fb, 0, 0), buf_load1(rs, (fr + 3) * fb, 0, 0));
}

// squares of `count` frames of one chain, starting at frame f0 of stream b, into dst (4 per lane and round)
template <bool MONO>
__device__ __forceinline__ void par_load_squares(const RArgs &a, int64_t b, int chain, int64_t f0, int count, float *dst, int lane)
{
    const bool from_x = chain < 2;
    const int ch = chain & 1;
    if (from_x && MONO) {
        const v4i rs = make_rsrc(a.x + b * a.n + f0, (a.n - f0) * 4);
        for (int u = 0; u < count; u += 256) {
            const int fr = u + 4 * lane;
            const float s0 = buf_load1(rs, (fr + 0) * 4, 0, 0), s1 = buf_load1(rs, (fr + 1) * 4, 0, 0);
            const float s2 = buf_load1(rs, (fr + 2) * 4, 0, 0), s3 = buf_load1(rs, (fr + 3) * 4, 0, 0);
            *(float4 *)(dst + fr) = make_float4(s0 * s0, s1 * s1, s2 * s2, s3 * s3);
        }
    } else {
        int64_t bytes;
        const float *src = par_pair_base(a, from_x ? a.x : a.y, b, f0, ch, &bytes);
        const v4i rs = make_rsrc(src, bytes);
        const int fb = a.C * 4;
        for (int u = 0; u < count; u += 256) {
            const int fr = u + 4 * lane;
            const float s0 = buf_load1(rs, (fr + 0) * fb, 0, 0), s1 = buf_load1(rs, (fr + 1) * fb, 0, 0);
            const float s2 = buf_load1(rs, (fr + 2) * fb, 0, 0), s3 = buf_load1(rs, (fr + 3) * fb, 0, 0);
            *(float4 *)(dst + fr) = make_float4(s0 * s0, s1 * s1, s2 * s2, s3 * s3);
        }
    }
}

// one wave per (stream, chain): blockIdx.x = stream * 4 + chain
template <bool MONO>
__global__ __launch_bounds__(64) void rms_par_stitch_kernel(const RArgs a)
{
    __shared__ __attribute__((aligned(16))) float crossing[kParSlots][kSeqGroup];      // squares of the foreseen groups
    __shared__ uint32_t grp_lds[kParSlots][2 * kParGroups];                            // and their blocks' group tallies
    __shared__ __attribute__((aligned(16))) float scratch[kSeqGroup];                  // a group fetched on demand
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x >> 2;
    const int chain = blockIdx.x & 3;
    const ParRec *rec = a.rec + (b * 4 + chain) * a.nblocks;
    const ParGrp *grp = a.grp + (b * 4 + chain) * a.nblocks;

    // the groups the walk is expected to add one by one (binade crossings, ties: ParRec::need): fetch them
    // and their blocks' group tallies before the walk, all in flight together, so that the walk does not
    // wait for memory there; anything not foreseen is fetched on demand
    __shared__ int slot_key[kParSlots];                    // block * 8 + group, ascending
    int slots = 0;
    for (int base = 1; base < a.nblocks && slots < kParSlots; base += 256) {
        uint32_t need[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int mine = base + 64 * u + lane;
            need[u] = mine < a.nblocks ? (rec[mine].need & 0xffu) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t cnt = (uint32_t)__builtin_popcount(need[u]);
            const uint32_t incl = wave_prefix_u32(cnt, lane);
            int at = slots + (int)(incl - cnt);
            for (uint32_t nm = need[u]; nm != 0; nm &= nm - 1, ++at)
                if (at < kParSlots) slot_key[at] = (base + 64 * u + lane) * 8 + (int)__builtin_ctz(nm);
            slots += __builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    slots = min(slots, kParSlots);
    __syncthreads();
    {
        float4 raw[kParSlots];
        uint32_t tally[kParSlots];
#pragma unroll
        for (int s = 0; s < kParSlots; ++s) {
            raw[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            tally[s] = 0u;
            if (s < slots) {
                const int key = slot_key[s];
                raw[s] = par_fetch4<MONO>(a, b, chain, (int64_t)(key >> 3) * kParFrames + (int64_t)(key & 7) * kSeqGroup, lane);
                if (lane < 2 * kParGroups) tally[s] = ((const uint32_t *)&grp[key >> 3])[lane];
            }
        }
#pragma unroll
        for (int s = 0; s < kParSlots; ++s) {
            if (s < slots) {
                const float4 v = raw[s];
                *(float4 *)(crossing[s] + 4 * lane) = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
                if (lane < 2 * kParGroups) grp_lds[s][lane] = tally[s];
            }
        }
    }
    __syncthreads();
    const int my_key = lane < slots ? slot_key[lane] : -1;         // lane s knows slot s (kParSlots <= 64)
    float acc = a.first[b * 4 + chain];
    ParRec nxt = {kParZero, 0u, 0u, 0u};
    if (1 + lane < a.nblocks) nxt = rec[1 + lane];
    for (int base = 1; base < a.nblocks; base += 64) {
        const ParRec r = nxt;
        nxt = ParRec{kParZero, 0u, 0u, 0u};
        if (base + 64 + lane < a.nblocks) nxt = rec[base + 64 + lane];     // in flight while this chunk is walked
        const int count = min(64, a.nblocks - base);
        int cur = 0;
        while (cur < count) {
            // every block from `cur` on that can be settled against acc's binade, as far as the integer
            // sum stays inside it: one prefix sum accepts the whole run
            const uint32_t ab = __float_as_uint(acc);
            const uint32_t eb = ab >> 23;
            const bool zero = (r.tag & kParZero) != 0;
            const bool settled = (r.tag & kParBad) == 0 && (r.tag & 0x1ffu) == eb;      // tallied against the binade acc is in
            const bool plain = zero || (settled && (r.tag & kParTies) == 0);
            const uint32_t q = (lane >= cur && plain && !zero) ? r.qtot : 0u;
            const uint32_t pfx = wave_prefix_u32(q, lane);
            const uint32_t mant = (ab & 0x7fffffu) | 0x800000u;
            const bool fits = lane < cur || (plain && (zero || (eb >= 23 && eb < 255 && mant + pfx < (1u << 24))));
            uint64_t stop = __ballot(!fits);
            if (count < 64) stop |= ~0ull << count;
            const int f = stop ? (int)__builtin_ctzll(stop) : 64;           // first block that does not fit
            if (f > cur) {
                const uint32_t add = (uint32_t)__builtin_amdgcn_readlane((int)pfx, f - 1);
                if (add != 0) acc = __uint_as_float((eb << 23) | ((mant + add) & 0x7fffffu));
                cur = f;
                continue;
            }
            // blocks whose only trouble is ties: the tally kernel left both answers (TieScan), the parity of the
            // mantissa picks one - a few scalar operations per block, as many blocks in a row as there are
            {
                const uint64_t tied = __ballot(settled && !zero && (r.tag & kParTies) != 0);
                uint32_t m = mant;
                int c = cur;
                while (c < count && ((tied >> c) & 1ull) != 0 && eb >= 23 && eb < 255) {
                    const uint32_t qt = (uint32_t)__builtin_amdgcn_readlane((int)r.qtot, c);
                    const uint32_t nd = (uint32_t)__builtin_amdgcn_readlane((int)r.need, c);
                    const int d = (m & 1u) ? ((int)nd >> 20) : ((int)(nd << 12) >> 20);          // signed 12-bit fields
                    const uint32_t grown = m + qt + (uint32_t)d;
                    if (grown >= (1u << 24)) break;           // the sum leaves its binade inside this block: group by group below
                    m = grown;
                    ++c;
                }
                if (c > cur) {
                    acc = __uint_as_float((eb << 23) | (m & 0x7fffffu));
                    cur = c;
                    continue;
                }
            }
            // block `cur` needs care: group by group, each either one integer add (its tally against the
            // binade acc is really in) or its 256 dependent additions - exact whatever the data
            const uint32_t tag = (uint32_t)__builtin_amdgcn_readlane((int)r.tag, cur);
            const uint32_t bad = (uint32_t)__builtin_amdgcn_readlane((int)r.bad, cur);
            const int blk = base + cur;
            ++cur;
            const int e0 = (int)(tag & 0x1ffu);
            const uint64_t of_block = __ballot((my_key >> 3) == blk);       // the slots holding groups of this block
            uint32_t gq = 0;                               // lane g: qe[g], lane 8 + g: qf[g]
            if (of_block != 0) { if (lane < 2 * kParGroups) gq = grp_lds[__builtin_ctzll(of_block)][lane]; }
            else if (lane < 2 * kParGroups) gq = ((const uint32_t *)&grp[blk])[lane];
            for (int g = 0; g < kParGroups; ++g) {
                const uint32_t cb = __float_as_uint(acc);
                const int k = (int)(cb >> 23) - e0;       // 0: still in the predicted binade, 1: the next one
                const uint32_t m2 = (cb & 0x7fffffu) | 0x800000u;
                if ((k == 0 || k == 1) && e0 >= 23 && e0 + k < 255 && !((bad >> (8 * k + g)) & 1u)) {
                    const uint32_t qg = (uint32_t)__builtin_amdgcn_readlane((int)gq, 8 * k + g);
                    if (m2 + qg < (1u << 24)) {
                        acc = __uint_as_float((cb & 0xff800000u) | ((m2 + qg) & 0x7fffffu));
                        continue;
                    }
                }
                const uint64_t held = __ballot(my_key == blk * 8 + g);
                const float *sqs = scratch;
                if (held != 0) {
                    sqs = crossing[__builtin_ctzll(held)];
                } else {
                    par_load_squares<MONO>(a, b, chain, (int64_t)blk * kParFrames + (int64_t)g * kSeqGroup, kSeqGroup, scratch, lane);
                }
                // ties are settled in integers; a group in which the sum crosses a binade takes the additions
                if (!seq_settle_ties(*(const float4 *)(sqs + 4 * lane), acc, (int)(__float_as_uint(acc) >> 23), lane))
                    acc = seq_add_group(sqs, acc);
            }
        }
    }
    if (lane == 0) {
        // where the sequential kernel leaves a stream's sums: x's C chains, then y's (stereo: b * 4 + chain)
        const int64_t stream = b / a.pairs;
        const int pair = (int)(b - stream * a.pairs);
        a.partials[stream * 2 * a.C + (chain >= 2 ? a.C : 0) + 2 * pair + (chain & 1)] = (double)acc;
    }
}

// ---- NumPy's sum of squares of a SINGLE-channel float32 signal, bit for bit ------------------------
// np.mean(np.square(a), axis=0) on an (n, 1) (or 1-D) float32 array is not the row-by-row recurrence
// above: the reduction runs over contiguous memory, which NumPy's add loop sums PAIRWISE, 8192 elements
// (its buffer size) at a time (utils/dsp.py:107-109; numpy loops_utils: pairwise_sum):
//   sum   = p(chunk 0) + p(chunk 1) + ...                       left to right, float32
//   p(m)  = m < 8:     0 + a0 + a1 + ...
//           m <= 128:  eight strided accumulators r_j = a_j + a_(8+j) + ..., ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
//                      then the m % 8 leftovers one by one
//           else:      p(first h) + p(rest),  h = m/2 rounded down to a multiple of 8
// (tests/test_properties_cpu.py holds this model against np.add.reduce for many lengths).  The tree of one chunk
// is at most 7 levels deep; thread t of a chain walks it by the bits of t, the threads that arrive at
// a leaf first sum it exactly as NumPy does, and the inner nodes are formed level by level in LDS.
constexpr int kPwChunk = 8192, kPwLeaf = 128, kPwDepth = 7, kPwThreads = 1 << kPwDepth;

struct PwArgs {
    const float *__restrict__ x;       // [batch][n], one channel
    const float *__restrict__ y;
    int64_t n;
    int32_t nchunks;
    float *__restrict__ chunk_sums;    // [batch][2][nchunks]
    double *__restrict__ partials;     // [batch][2]: sum x^2, sum y^2 (as the reduce kernel reads them with one row)
};

// the node reached from the root of an m-element chunk along the top `steps` bits of `path` (kPwDepth bits,
// most significant first); stops at a leaf.  Returns the depth reached.
__device__ __forceinline__ int pw_walk(int m, int path, int steps, int &off, int &len)
{
    off = 0;
    len = m;
    int d = 0;
    for (; d < steps && len > kPwLeaf; ++d) {
        int half = len / 2;
        half -= half % 8;
        if ((path >> (kPwDepth - 1 - d)) & 1) { off += half; len -= half; }
        else len = half;
    }
    return d;
}

__device__ __forceinline__ float pw_leaf(const float *__restrict__ a, int len)
{
    if (len < 8) {
        float r = 0.0f;
        for (int i = 0; i < len; ++i) r = r + a[i] * a[i];
        return r;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j] * a[j];
    int i = 8;
    for (; i < len - (len % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j] * a[i + j];
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < len; ++i) res = res + a[i] * a[i];
    return res;
}

// grid (chunks, batch), 2 * kPwThreads threads: the x chain and the y chain of one 8192-sample chunk
__global__ __launch_bounds__(2 * kPwThreads) void rms_pairwise_kernel(const PwArgs a)
{
    __shared__ float heap[2][2 << kPwDepth];              // node (depth d, path p) at (1 << d) + p
    const int chain = threadIdx.x >> kPwDepth, t = threadIdx.x & (kPwThreads - 1);
    const int64_t b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int64_t first = (int64_t)chunk * kPwChunk;
    const float *src = (chain == 0 ? a.x : a.y) + b * a.n + first;
    const int m = (int)min((int64_t)kPwChunk, a.n - first);
    int off, len;
    const int d = pw_walk(m, t, kPwDepth, off, len);
    // the threads sharing the leaf's path prefix all arrive at it: the one whose unused bits are zero sums it
    if ((t & ((1 << (kPwDepth - d)) - 1)) == 0) heap[chain][(1 << d) + (t >> (kPwDepth - d))] = pw_leaf(src + off, len);
    __syncthreads();
    for (int lvl = kPwDepth - 1; lvl >= 0; --lvl) {
        if (t < (1 << lvl)) {
            const int reached = pw_walk(m, t << (kPwDepth - lvl), lvl, off, len);
            if (reached == lvl && len > kPwLeaf)           // an inner node: its children sit one level down
                heap[chain][(1 << lvl) + t] = heap[chain][(2 << lvl) + 2 * t] + heap[chain][(2 << lvl) + 2 * t + 1];
        }
        __syncthreads();
    }
    if (t == 0) a.chunk_sums[(b * 2 + chain) * a.nchunks + chunk] = heap[chain][1];
}

// the chunks' sums, left to right (one lane per chain; a 10 s signal has 59 of them)
__global__ __launch_bounds__(64) void rms_pairwise_fold_kernel(const PwArgs a)
{
    const int64_t b = blockIdx.x;
    const int chain = threadIdx.x;
    if (chain < 2) {
        const float *cs = a.chunk_sums + (b * 2 + chain) * a.nchunks;
        float acc = cs[0];
        for (int k = 1; k < a.nchunks; ++k) acc = acc + cs[k];
        a.partials[b * 2 + chain] = (double)acc;
    }
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
    } else if (C % 4 == 0 && (((uintptr_t)ys) & 15) == 0) {
        // 4k channels: 16 bytes per lane - a channel quad of a frame (the scalar form below ran cfg5's shape at 4.0 TB/s where the
        // stereo form runs 6.6: a 4-byte access and a 64-bit modulo per element).  A chunk starts on a frame, so a lane's quad is
        // (4 * lane + k * 4 * threads) mod C: constant over the loop whenever 4 * threads is a multiple of C (4, 8, 16, 32 ... channels)
        int at = (int)((4 * threadIdx.x) % (unsigned)C);
        const int step = (int)((4 * kEpiThreads) % (unsigned)C);
        float4 sc = make_float4(scale[at], scale[at + 1], scale[at + 2], scale[at + 3]);
        for (int64_t e = e0 + 4 * threadIdx.x; e < e1; e += 4 * kEpiThreads) {
            float4 v = *(float4 *)(ys + e);
            v.x = v.x * sc.x; v.y = v.y * sc.y; v.z = v.z * sc.z; v.w = v.w * sc.w;
            *(float4 *)(ys + e) = v;
            if (step != 0) {
                at += step; if (at >= C) at -= C;
                sc = make_float4(scale[at], scale[at + 1], scale[at + 2], scale[at + 3]);
            }
        }
    } else {
        for (int64_t e = e0 + threadIdx.x; e < e1; e += kEpiThreads) ys[e] = ys[e] * scale[(int)(e % C)];
    }
}

}  // namespace vnd
