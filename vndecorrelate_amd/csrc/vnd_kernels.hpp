// vnd_kernels.hpp - gfx950 (CDNA4, wave64) kernels of the velvet-noise tap sum.
//
//   y[b,n,c] = sum_k w[c,k] * x[b, n + i[c,k], c]        (n + i >= N drops out)
//
// What the reference does with K NumPy slice-adds per channel
// (ckonst/VNDecorrelate src/vndecorrelate/decorrelation.py:649-658 and
// :402-414) is, on the GPU, a *gather* with a forward halo of max(i) frames.
// It is HBM-streaming work (8 algorithmic bytes per output sample) whose
// on-chip cost is K LDS reads per output, so the design is about LDS bandwidth
// and instruction issue, not FLOPs - no MFMA anywhere:
//
//   * one workgroup = one time tile of one stream (x CG channels);
//   * the tile + halo is read once from HBM with coalesced, range-checked buffer
//     loads and de-interleaved into per-channel LDS planes, so every later LDS
//     read is a unit-stride, conflict-free access;
//   * each lane owns PAIRS of consecutive frames: a tap is one ds_read_b64 (the
//     widest conflict-free form, 256 B/clk/CU) feeding one packed FMA;
//   * tap records are wave-uniform and live in SGPRs;
//   * outputs leave as 16-byte interleaved buffer stores, 1 KiB per wave-instruction.
//
// Two kernels share the staging and store code:
//   conv_ordered_kernel  taps in table order with the reference's association
//                        (VND_MODE_EXACT bit-identical, VND_MODE_FMA);
//   conv_fast_kernel     free summation order, every tap an aligned read
//                        (VND_MODE_FAST, the throughput mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

#include "vnd_polar.hpp"

namespace vnd {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

struct Tap {          // ordered kernels: table order, reference association
    int32_t idx;
    float w;
};

// Fast-mode record.  The weight comes FIRST: records are fetched into SGPR pairs
// and hipcc feeds v_pk_fma_f32 the LOW half of an even-aligned pair as the
// splatted scalar operand.
struct FastTap {
    float w;          // weight * segment gain
    int32_t off;      // LDS byte offset (i & ~1) * 4
};

struct KArgs {
    const float *__restrict__ x;
    float *__restrict__ y;
    const Tap *__restrict__ taps;
    const FastTap *__restrict__ taps_fast;  // fast mode: per channel [even taps | odd taps], zero-padded
    const FastTap *__restrict__ taps_ord;   // ordered modes: table order, {w, idx * 4}, zero-padded
    const int32_t *__restrict__ fast_off;   // [C+1] start of each channel's list in taps_fast
    const int32_t *__restrict__ fast_even;  // [C]   number of even-offset taps (the odd ones follow)
    const int32_t *__restrict__ tap_off;    // [C+1]
    const int32_t *__restrict__ seg_off;    // [C+1] or nullptr (function-path table)
    const int32_t *__restrict__ seg_end;    // exclusive ends, absolute tap positions
    const float *__restrict__ seg_gain;
    const uint8_t *__restrict__ chan_flags; // bit0: pass-through, or nullptr
    int64_t n;                 // frames per stream
    int32_t C;                 // interleaved output channels (= channels of the tap table)
    int32_t Cx;                // interleaved input channels; output channel c reads input channel c % Cx
    int32_t groups;            // C / CG
    int32_t tiles;             // tiles per stream
    int32_t W;                 // floats per LDS plane (tile + halo, even)
    int32_t apply_gain;
    int32_t stream_out;        // 1: non-temporal stores (large outputs)
    uint32_t nblocks;
    // fused decorrelate epilogue (fast kernel, EPI instantiation; vnd_epilogue.hpp has the two-pass form)
    double *__restrict__ epi_partials;      // [batch][tiles][2*C]: sum x_c^2, then sum y_c^2, per tile
    int32_t epi_ms_encode, epi_use_width, epi_normalize;
    float epi_w_mid, epi_w_side;
    // moments sink (EPI instantiations, candidate scan of SURVEY.md 8 f3): when set, the workgroup's
    // channel pair is one candidate's (L, R); its tile is reduced to the eight polar moments
    // (vnd_polar.hpp) and NOT written - y never exists.  [stream][tile][groups][8]
    double *__restrict__ sink_partials;
};

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Map
// them so that one XCD walks CONSECUTIVE logical tiles: a tile's halo is the
// next tile's head, and channel groups of one tile share cache lines, so both
// re-reads hit that XCD's L2 instead of going back to the fabric (measured:
// FETCH_SIZE equals the algorithmic read bytes).  Bijective for any grid size.
// Placement only changes speed, never results.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nwg)
{
    const uint32_t q = nwg >> 3, r = nwg & 7u, xcd = b & 7u;
    const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (b >> 3);
}

struct BlockCoord {
    int group, tile;
    int64_t stream;
};

__device__ __forceinline__ BlockCoord decode_block(const KArgs &a)
{
    uint32_t lid = xcd_remap(blockIdx.x, a.nblocks);
    BlockCoord bc;
    bc.group = lid % (uint32_t)a.groups;
    lid /= (uint32_t)a.groups;
    bc.tile = lid % (uint32_t)a.tiles;
    bc.stream = lid / (uint32_t)a.tiles;
    return bc;
}

// Workgroup reduction of one tile's polar moments into sink_partials (fixed order: deterministic).
template <int NT>
__device__ __forceinline__ void sink_reduce_store(const KArgs &a, const BlockCoord &bc, const PolarAcc &acc, int tid)
{
    __shared__ double sink_red[NT / 64][kMoments];
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
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < kMoments; ++k) sink_red[tid >> 6][k] = v[k];
    }
    __syncthreads();
    if (tid < kMoments) {
        double t = sink_red[0][tid];
        for (int w = 1; w < NT / 64; ++w) t = tid == 4 ? fmax(t, sink_red[w][tid]) : t + sink_red[w][tid];
        a.sink_partials[(((int64_t)bc.stream * a.tiles + bc.tile) * a.groups + bc.group) * kMoments + tid] = t;
    }
}

// ---- global memory: raw buffer descriptors ---------------------------------------
// All global traffic goes through raw buffer descriptors whose num_records is
// the number of bytes left in the stream from the tile start.  The hardware
// range check is per dword for dword/x2/x4 accesses: loads past the end of the
// stream return 0 - exactly the reference's "term drops out when n + i >= N"
// (decorrelation.py:656-658; adding 0.0f is exact) - and stores past the end are
// discarded, so neither the halo of a stream's last tiles nor a ragged final
// pair needs a branch, and no access can leave the stream's allocation.
//
// The intrinsics are bound by name: the clang builtin
// __builtin_amdgcn_raw_buffer_load_b64 of ROCm 7.2 lowers to a 32-bit load and
// splats it.
__device__ float buf_load1(v4i rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ v2f buf_load2(v4i rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ v4f buf_load4(v4i rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void buf_store1(float d, v4i rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ void buf_store2(v2f d, v4i rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
__device__ void buf_store4(v4f d, v4i rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");

// Descriptor of a raw (stride 0) buffer: base, num_records in BYTES, gfx9 dword format.
// Built from kernel arguments and blockIdx only, so it lives in SGPRs.
__device__ __forceinline__ v4i make_rsrc(const void *base, int64_t bytes)
{
    const int64_t cap = 0x7fffffff;            // one tile's window is far smaller
    const uint64_t addr = (uint64_t)base;
    v4i r;
    r.x = (int)(uint32_t)addr;
    r.y = (int)((uint32_t)(addr >> 32) & 0xffffu);
    r.z = (int)(bytes < 0 ? 0 : (bytes > cap ? cap : bytes));
    r.w = 0x00020000;
    return r;
}

// A lane moves PAIRS of frames.  Access shape is picked per workgroup from the
// alignment of its first sample (all three are the same arithmetic downstream):
//   kPair   one access of 2*CG dwords   (block owns all channels, base 8*CG-aligned)
//   kFrame  one access of CG dwords per frame            (base 4*CG-aligned)
//   kDword  dword accesses, any alignment, any channel stride
// Byte offsets are written as  index * constant  so that hipcc can prove the
// natural alignment and keeps the wide buffer_load/store_dwordx2/x4 forms.
enum { kPair = 0, kFrame = 1, kDword = 2 };

template <int CG>
__device__ __forceinline__ int access_shape(const void *base, int C)
{
    const uintptr_t p = (uintptr_t)base;
    if (C == CG && (p & (8 * CG - 1)) == 0) return kPair;
    if ((p & (4 * CG - 1)) == 0) return kFrame;      // C % CG == 0 by construction
    return kDword;
}

// strideG = C / CG  (frame stride in units of CG floats);  q = pair index in the window
// Cache policy of the streaming traffic (aux operand of the buffer intrinsics:
// bit0 sc0, bit1 nt, bit4 sc1).  Build-time knobs for experiments.
#ifndef VND_LOAD_AUX
#define VND_LOAD_AUX 0
#endif
#ifndef VND_STORE_AUX
#define VND_STORE_AUX 0
#endif
constexpr int kLoadAux = VND_LOAD_AUX, kStoreAux = VND_STORE_AUX;

template <int CG, int SHAPE>
__device__ __forceinline__ void load_pair(v4i rsrc, int q, int strideG, int C, float (&v)[2 * CG])
{
    if constexpr (SHAPE == kPair) {
        const int off = q * (8 * CG);
        if constexpr (CG == 1) {
            const v2f t = buf_load2(rsrc, off, 0, kLoadAux);
            v[0] = t.x; v[1] = t.y;
        } else if constexpr (CG == 2) {
            const v4f t = buf_load4(rsrc, off, 0, kLoadAux);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            const v4f p = buf_load4(rsrc, off, 0, kLoadAux), t = buf_load4(rsrc, off + 16, 0, kLoadAux);
            v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w;
            v[4] = t.x; v[5] = t.y; v[6] = t.z; v[7] = t.w;
        }
    } else if constexpr (SHAPE == kFrame) {
        const int off0 = (2 * q) * strideG * (4 * CG);
        const int off1 = (2 * q + 1) * strideG * (4 * CG);
        if constexpr (CG == 1) {
            v[0] = buf_load1(rsrc, off0, 0, kLoadAux); v[1] = buf_load1(rsrc, off1, 0, kLoadAux);
        } else if constexpr (CG == 2) {
            const v2f p = buf_load2(rsrc, off0, 0, kLoadAux), t = buf_load2(rsrc, off1, 0, kLoadAux);
            v[0] = p.x; v[1] = p.y; v[2] = t.x; v[3] = t.y;
        } else {
            const v4f p = buf_load4(rsrc, off0, 0, kLoadAux), t = buf_load4(rsrc, off1, 0, kLoadAux);
            v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w;
            v[4] = t.x; v[5] = t.y; v[6] = t.z; v[7] = t.w;
        }
    } else {
        const int off0 = (2 * q) * C * 4;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            v[c] = buf_load1(rsrc, off0 + 4 * c, 0, kLoadAux);
            v[CG + c] = buf_load1(rsrc, off0 + C * 4 + 4 * c, 0, kLoadAux);
        }
    }
}

// out pair -> interleaved frames;  v[c] = frame 2q, v[CG + c] = frame 2q+1
template <int CG, int SHAPE, int AUX>
__device__ __forceinline__ void store_pair(v4i rdst, int q, int strideG, int C, const float (&v)[2 * CG])
{
    if constexpr (SHAPE == kPair) {
        const int off = q * (8 * CG);
        if constexpr (CG == 1) {
            v2f t; t.x = v[0]; t.y = v[1];
            buf_store2(t, rdst, off, 0, AUX);
        } else if constexpr (CG == 2) {
            v4f t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
            buf_store4(t, rdst, off, 0, AUX);
        } else {
            v4f t, u;
            t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
            u.x = v[4]; u.y = v[5]; u.z = v[6]; u.w = v[7];
            buf_store4(t, rdst, off, 0, AUX);
            buf_store4(u, rdst, off + 16, 0, AUX);
        }
    } else if constexpr (SHAPE == kFrame) {
        const int off0 = (2 * q) * strideG * (4 * CG);
        const int off1 = (2 * q + 1) * strideG * (4 * CG);
        if constexpr (CG == 1) {
            buf_store1(v[0], rdst, off0, 0, AUX);
            buf_store1(v[1], rdst, off1, 0, AUX);
        } else if constexpr (CG == 2) {
            v2f t, u; t.x = v[0]; t.y = v[1]; u.x = v[2]; u.y = v[3];
            buf_store2(t, rdst, off0, 0, AUX);
            buf_store2(u, rdst, off1, 0, AUX);
        } else {
            v4f t, u;
            t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
            u.x = v[4]; u.y = v[5]; u.z = v[6]; u.w = v[7];
            buf_store4(t, rdst, off0, 0, AUX);
            buf_store4(u, rdst, off1, 0, AUX);
        }
    } else {
        const int off0 = (2 * q) * C * 4;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            buf_store1(v[c], rdst, off0 + 4 * c, 0, AUX);
            buf_store1(v[CG + c], rdst, off0 + C * 4 + 4 * c, 0, AUX);
        }
    }
}

// ---- staging: HBM -> registers -> per-channel LDS planes ------------------------
// kStageDepth independent loads are issued before the first LDS write, so every
// wave keeps several KiB in flight; with one load per lane and iteration the
// kernel is bound by HBM latency, not bandwidth.
#ifndef VND_STAGE_DEPTH
#define VND_STAGE_DEPTH 4
#endif
constexpr int kStageDepth = VND_STAGE_DEPTH;

template <int CG>
__device__ __forceinline__ void write_pair(float *plane, int W, int f, const float (&v)[2 * CG])
{
#pragma unroll
    for (int c = 0; c < CG; ++c) *(float2 *)(plane + c * W + f) = make_float2(v[c], v[CG + c]);
}

template <int NT, int CG, int SHAPE>
__device__ __forceinline__ void stage_window_shape(float *plane, v4i rsrc, int C, int W, int tid)
{
    const int npairs = W >> 1;
    const int strideG = C / CG;
    for (int q0 = tid; q0 < npairs; q0 += NT * kStageDepth) {
        float v[kStageDepth][2 * CG];
        int q[kStageDepth];
        // Lanes past the window re-do its last pair (same value to the same slot)
        // rather than branch: the loads stay one straight-line burst.
#pragma unroll
        for (int u = 0; u < kStageDepth; ++u) {
            q[u] = min(q0 + u * NT, npairs - 1);
            load_pair<CG, SHAPE>(rsrc, q[u], strideG, C, v[u]);   // in range, or zero-filled by the descriptor
        }
#pragma unroll
        for (int u = 0; u < kStageDepth; ++u) write_pair<CG>(plane, W, 2 * q[u], v[u]);
    }
}

// src = this block's first sample; bytes_left = bytes from there to the end of the stream
template <int NT, int CG>
__device__ __forceinline__ void stage_window(float *plane, const float *src, int64_t bytes_left, int C, int W,
                                             int tid)
{
    const v4i rsrc = make_rsrc(src, bytes_left);
    const int shape = access_shape<CG>(src, C);          // workgroup-uniform
    if (shape == kPair)       stage_window_shape<NT, CG, kPair>(plane, rsrc, C, W, tid);
    else if (shape == kFrame) stage_window_shape<NT, CG, kFrame>(plane, rsrc, C, W, tid);
    else                      stage_window_shape<NT, CG, kDword>(plane, rsrc, C, W, tid);
}

// One (frame pair, CG channels) result per lane and j: v[c] = frame 2q, v[CG+c] = frame 2q+1.
// Range-checked buffer stores: frames past the end of the stream are dropped.
// AUX = 2 marks the stores non-temporal: a large output is not read again before it has left
// the caches, and streaming it past them measured +0.5...2 % on three boxes (it is a little less
// energy per byte, and the kernel runs on the power cap).
template <int CG, int AUX>
__device__ __forceinline__ void store_result_aux(v4i rdst, int shape, int q, int strideG, int C,
                                                 const float (&v)[2 * CG])
{
    if (shape == kPair)       store_pair<CG, kPair, AUX>(rdst, q, strideG, C, v);
    else if (shape == kFrame) store_pair<CG, kFrame, AUX>(rdst, q, strideG, C, v);
    else                      store_pair<CG, kDword, AUX>(rdst, q, strideG, C, v);
}

template <int CG>
__device__ __forceinline__ void store_result(v4i rdst, int shape, int stream_out, int q, int strideG, int C,
                                             const float (&v)[2 * CG])
{
    if (stream_out) store_result_aux<CG, 2>(rdst, shape, q, strideG, C, v);
    else            store_result_aux<CG, kStoreAux>(rdst, shape, q, strideG, C, v);
}

// ---- LDS reads ---------------------------------------------------------------------
__device__ __forceinline__ unsigned lds_addr(const float *p)
{
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p;
}

// =====================================================================================
// Fast kernel (VND_MODE_FAST): same tap sum, free summation order, one FMA per tap.
//
// The LDS read is the scarce resource (K reads per output), so every tap must be
// ONE aligned ds_read_b64 per output pair.  A lane's pair (2q, 2q+1) is aligned
// for even offsets only; for an odd offset i the pair (2q-1, 2q) is the aligned
// one (it reads x[2q + (i-1)], x[2q + i]).  So each lane keeps TWO accumulator
// sets: accE for outputs (2q, 2q+1) fed by the even taps, accO for outputs
// (2q-1, 2q) fed by the odd taps - both read at offset (i & ~1) - and the two are
// merged once per tile:  y[2q] = accE.x + accO.y ,  y[2q+1] = accE.y + accO(q+1).x,
// the neighbour's value going through LDS after the tap loops.  The odd part of
// the tile's very last output has no lane; one wave reduces it from the tap list.
//
// The host splits each channel's taps into an even and an odd list
// (vnd_taps_create) with the segment gain folded into the weight.
//
// The LDS reads of the tap loops are inline asm so that (a) each stays a plain
// ds_read_b64 with an immediate offset and (b) the waits are counted by hand:
// hipcc's own bookkeeping drains the queue at loop headers (WAW on recycled
// registers).  Protocol (guide 5.7 form ii): "=v" loads, then ONE wait statement
// that names every destination "+v" before its first consumer; LDS returns in
// order, so lgkmcnt(N) with the N newest reads belonging to the next buffer
// means "this buffer has landed".  Buffers never cross a branch or loop edge: a
// loop-carried buffer makes hipcc copy registers whose LDS data has not arrived.
// =====================================================================================
template <int NT, int J>
__device__ __forceinline__ v2f ds_read_pair(unsigned addr)
{
    static_assert(J * 2 * NT * 4 <= 65535, "ds offset field is 16 bits");
    v2f r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(J * 2 * NT * 4));
    return r;
}

template <int NT, int R, int... Js>
__device__ __forceinline__ void issue_reads_seq(v2f (&buf)[R], unsigned addr, std::integer_sequence<int, Js...>)
{
    ((buf[Js] = ds_read_pair<NT, Js>(addr)), ...);
}

template <int NT, int R>
__device__ __forceinline__ void issue_reads(v2f (&buf)[R], unsigned addr)
{
    issue_reads_seq<NT, R>(buf, addr, std::make_integer_sequence<int, R>{});
}

template <int NT, int J, int BYTES>
__device__ __forceinline__ float ds_read_dword(unsigned addr)
{
    static_assert(J * 2 * NT * 4 + BYTES <= 65535, "ds offset field is 16 bits");
    float r;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(J * 2 * NT * 4 + BYTES));
    return r;
}

template <int NT, int R, int BYTES, int... Js>
__device__ __forceinline__ void issue_dwords_seq(float (&buf)[R], unsigned addr, std::integer_sequence<int, Js...>)
{
    ((buf[Js] = ds_read_dword<NT, Js, BYTES>(addr)), ...);
}

// the dword BYTES past each of the lane's R pair addresses
template <int NT, int R, int BYTES>
__device__ __forceinline__ void issue_dwords(float (&buf)[R], unsigned addr)
{
    issue_dwords_seq<NT, R, BYTES>(buf, addr, std::make_integer_sequence<int, R>{});
}

template <int R>
__device__ __forceinline__ void wait_dwords(float (&a)[R], float (&b)[R])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]));
#pragma unroll
    for (int j = 1; j < R; ++j) asm volatile("" : "+v"(a[j]));
#pragma unroll
    for (int j = 0; j < R; ++j) asm volatile("" : "+v"(b[j]));
}

template <int N, int R>
__device__ __forceinline__ void wait_reads(v2f (&buf)[R])
{
    static_assert(N <= 15, "lgkmcnt is a 4-bit field");
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(buf[0]) : "n"(N));
#pragma unroll
    for (int j = 1; j < R; ++j) asm volatile("" : "+v"(buf[j]));
}

template <int R>
__device__ __forceinline__ void consume(float2 (&acc)[R], const v2f (&buf)[R], float w)
{
#pragma unroll
    for (int j = 0; j < R; ++j) {
        acc[j].x = __builtin_fmaf(buf[j].x, w, acc[j].x);
        acc[j].y = __builtin_fmaf(buf[j].y, w, acc[j].y);
    }
}

// Four taps in straight-line code: the reads of the next tap are in flight while
// a tap is consumed.  Per tap the vector pipe issues ONE v_add (address = lane
// base + scalar byte offset) plus the R packed FMAs.
template <int NT, int R>
__device__ __forceinline__ void tap_group4(const FastTap (&t)[16], int g, unsigned lane_addr, float2 (&acc)[R])
{
    v2f b0[R], b1[R], b2[R], b3[R];
    issue_reads<NT, R>(b0, lane_addr + (unsigned)t[4 * g + 0].off);
    issue_reads<NT, R>(b1, lane_addr + (unsigned)t[4 * g + 1].off);
    wait_reads<R, R>(b0);          // only b1's R reads may still be outstanding
    consume<R>(acc, b0, t[4 * g + 0].w);
    issue_reads<NT, R>(b2, lane_addr + (unsigned)t[4 * g + 2].off);
    wait_reads<R, R>(b1);
    consume<R>(acc, b1, t[4 * g + 1].w);
    issue_reads<NT, R>(b3, lane_addr + (unsigned)t[4 * g + 3].off);
    wait_reads<R, R>(b2);
    consume<R>(acc, b2, t[4 * g + 2].w);
    wait_reads<0, R>(b3);
    consume<R>(acc, b3, t[4 * g + 3].w);
}

template <int NT, int R>
__device__ __forceinline__ void tap_single(const FastTap &t, unsigned lane_addr, float2 (&acc)[R])
{
    v2f b0[R];
    issue_reads<NT, R>(b0, lane_addr + (unsigned)t.off);
    wait_reads<0, R>(b0);
    consume<R>(acc, b0, t.w);
}

// 16 records into 16 SGPR pairs with one burst of scalar loads.  Inline asm on
// purpose: (w, off) of a record then sit in ONE even-aligned pair, which is the
// only scalar operand shape hipcc (ROCm 7.2) encodes correctly for the splatted
// weight of v_pk_fma_f32 - fed from sub-registers of a wider s_buffer_load tuple
// it multiplies every tap by the tuple's first dword.  Loads and their wait live
// in one statement (guide 5.7 form i), outputs early-clobber.  Profiling an
// earlier version that broadcast each record with two v_readlane and walked a
// ballot mask with s_ff1 showed the kernel bound by instruction issue (VALU ~70 %
// busy, SALU ~55 %).
__device__ __forceinline__ void load_taps16(const FastTap *ptr_any, FastTap (&t)[16])
{
    // "s" operands must be provably wave-uniform: re-assert it on both halves
    const uint64_t p64 = (uint64_t)ptr_any;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p64);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p64 >> 32));
    const FastTap *ptr = (const FastTap *)(((uint64_t)hi << 32) | lo);
    v2i r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
    asm volatile(
        "s_load_dwordx2 %0, %16, 0x0\n\ts_load_dwordx2 %1, %16, 0x8\n\t"
        "s_load_dwordx2 %2, %16, 0x10\n\ts_load_dwordx2 %3, %16, 0x18\n\t"
        "s_load_dwordx2 %4, %16, 0x20\n\ts_load_dwordx2 %5, %16, 0x28\n\t"
        "s_load_dwordx2 %6, %16, 0x30\n\ts_load_dwordx2 %7, %16, 0x38\n\t"
        "s_load_dwordx2 %8, %16, 0x40\n\ts_load_dwordx2 %9, %16, 0x48\n\t"
        "s_load_dwordx2 %10, %16, 0x50\n\ts_load_dwordx2 %11, %16, 0x58\n\t"
        "s_load_dwordx2 %12, %16, 0x60\n\ts_load_dwordx2 %13, %16, 0x68\n\t"
        "s_load_dwordx2 %14, %16, 0x70\n\ts_load_dwordx2 %15, %16, 0x78\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3), "=&s"(r4), "=&s"(r5), "=&s"(r6), "=&s"(r7),
          "=&s"(r8), "=&s"(r9), "=&s"(r10), "=&s"(r11), "=&s"(r12), "=&s"(r13), "=&s"(r14), "=&s"(r15)
        : "s"(ptr));
    const v2i r[16] = {r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15};
#pragma unroll
    for (int i = 0; i < 16; ++i) { t[i].w = __builtin_bit_cast(float, r[i].x); t[i].off = r[i].y; }
}

// One list of taps (all even or all odd offsets) into one accumulator set.
template <int NT, int R>
__device__ __forceinline__ void run_tap_array(const FastTap *__restrict__ tp, int n, unsigned lane_addr,
                                              float2 (&acc)[R])
{
    for (int k0 = 0; k0 < n; k0 += 16) {
        FastTap t[16];
        load_taps16(tp + k0, t);                 // the table is zero-padded by 16 records
        const int m = n - k0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (4 * g + 4 <= m) {
                tap_group4<NT, R>(t, g, lane_addr, acc);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * g + i < m) tap_single<NT, R>(t[4 * g + i], lane_addr, acc);
            }
        }
    }
}


// The decorrelate epilogue's pointwise steps on one lane's stereo frame pair, in the reference's
// float32 operation order (bit-identical to NumPy; the file is built with -ffp-contract=off).
// v = {L0, R0, L1, R1} of the convolution, xin the same of the input.
__device__ __forceinline__ void epi_pointwise(const KArgs &a, float (&v)[4], const float (&xin)[4])
{
#pragma unroll
    for (int f = 0; f < 2; ++f) {                    // the pair's two frames
        float y0 = v[2 * f], y1 = v[2 * f + 1];
        if (a.epi_ms_encode) {                       // utils/dsp.py:59-63
            const float mid = xin[2 * f] + xin[2 * f + 1];
            const float side = (y0 - y1) * 0.5f;
            y0 = (mid + side) * 0.5f;
            y1 = (mid - side) * 0.5f;
        }
        if (a.epi_use_width) {                       // utils/dsp.py:34-37
            float m = (y0 + y1) * 0.5f, sd = (y0 - y1) * 0.5f;
            m = m * a.epi_w_mid;
            sd = sd * a.epi_w_side;
            y0 = m + sd;
            y1 = m - sd;
        }
        v[2 * f] = y0; v[2 * f + 1] = y1;
    }
}

// EPI: the decorrelate epilogue's pointwise steps (side-channel encode, stereo width;
// reference utils/dsp.py:21-63) are applied to the tile before it is stored, in the
// reference's float32 operation order, and the tile's sums of x^2 and y^2 go to
// epi_partials for the scaling pass - the tile's input is still in LDS, so the fused
// form saves a full read of x and a read + write of y.  The exchange buffer then lives
// in the halo part of the planes (the host only picks EPI when it fits there).
//
// BC (fan-out of a mono input, Cx == 1): ONE plane is staged and every output channel of
// the group reads it; the exchange buffer then follows the plane instead of reusing it.
template <int NT, int CG, int R, bool EPI = false, bool BC = false>
__global__ __launch_bounds__(NT) void conv_fast_kernel(const KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = 2 * NT * R;
    constexpr int PG = BC ? 1 : CG;                          // planes staged
    const int tid = threadIdx.x;
    const int W = a.W;
    const BlockCoord bc = decode_block(a);
    const int C = a.C, Cx = a.Cx;
    const int c0 = bc.group * CG;
    const int cx0 = BC ? 0 : (Cx == C ? c0 : c0 % Cx);       // the host keeps Cx % CG == 0 unless BC
    const int64_t t0 = (int64_t)bc.tile * T;
    const float *__restrict__ xs = a.x + bc.stream * a.n * Cx;
    float *__restrict__ ys = a.y + bc.stream * a.n * C;
    const int64_t bytes_left = ((a.n - t0) * C - c0) * 4;
    float *plane = lds;                                      // [PG][W]
    stage_window<NT, PG>(plane, xs + t0 * Cx + cx0, ((a.n - t0) * Cx - cx0) * 4, Cx, W, tid);
    __syncthreads();

    float2 accE[CG][R], accO[CG][R];
    float edge[CG];                       // odd-tap part of the tile's last output (frame T-1)
    const int lane = tid & 63;
    const int lane_base = 2 * tid;

#pragma unroll
    for (int c = 0; c < CG; ++c) {
        const int ch = c0 + c;
        const float *pc = plane + (BC ? 0 : c * W);
        const float *pa = pc + lane_base;
        edge[c] = 0.0f;
#pragma unroll
        for (int j = 0; j < R; ++j) { accE[c][j] = make_float2(0.0f, 0.0f); accO[c][j] = make_float2(0.0f, 0.0f); }
        if (a.chan_flags != nullptr && (a.chan_flags[ch] & 1)) {       // unfiltered: copy through
#pragma unroll
            for (int j = 0; j < R; ++j) accE[c][j] = *(const float2 *)(pa + 2 * NT * j);
            continue;
        }
        const int first = __builtin_amdgcn_readfirstlane(a.fast_off[ch]);
        const FastTap *__restrict__ tp = a.taps_fast + first;
        const int n_all = __builtin_amdgcn_readfirstlane(a.fast_off[ch + 1]) - first;
        const int n_even = __builtin_amdgcn_readfirstlane(a.fast_even[ch]);
        const int n_odd = n_all - n_even;
        run_tap_array<NT, R>(tp, n_even, lds_addr(pa), accE[c]);
        run_tap_array<NT, R>(tp + n_even, n_odd, lds_addr(pa), accO[c]);
        // frame T-1 pairs with frame T, which no lane owns: its odd taps, x[T-1+i] = plane[T + (i-1)],
        // are reduced across the last wave (lane l takes odd tap l, l+64, ...)
        if (tid >= NT - 64) {                      // only the last wave's last lane publishes it
            float part = 0.0f;
            for (int k = lane; k < n_odd; k += 64) {
                const FastTap t = tp[n_even + k];
                part = __builtin_fmaf(pc[T + (t.off >> 2)], t.w, part);
            }
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) part += __shfl_xor(part, sh);
            edge[c] = part;
        }
    }

    // ---- merge the two accumulator sets: neighbour's accO.x through LDS -----------
    __syncthreads();                                   // every wave is done reading the planes
    constexpr int XS = T / 2 + 1;
    // exchange buffer [CG][T/2 + 1]: over the dead window, or (EPI) in each plane's halo part,
    // which keeps the tile's own input x[0 .. T) readable for the epilogue, or (BC) behind the plane
    float *xo = BC ? lds + W : (EPI ? lds + T : lds);
    const int xs_stride = (EPI && !BC) ? W : XS;
#pragma unroll
    for (int c = 0; c < CG; ++c) {
#pragma unroll
        for (int j = 0; j < R; ++j) xo[c * xs_stride + tid + NT * j] = accO[c][j].x;
        if (tid == NT - 1) xo[c * xs_stride + T / 2] = edge[c];
    }
    __syncthreads();

    float *dst = ys + t0 * C + c0;
    const v4i rdst = make_rsrc(dst, bytes_left);
    const int shape = access_shape<CG>(dst, C);
    const int strideG = C / CG;
    float sum_x[CG], sum_y[CG];
    PolarAcc pacc;
#pragma unroll
    for (int c = 0; c < CG; ++c) { sum_x[c] = 0.0f; sum_y[c] = 0.0f; }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int q = tid + NT * j;
        float v[2 * CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            v[c] = accE[c][j].x + accO[c][j].y;
            v[CG + c] = accE[c][j].y + xo[c * xs_stride + q + 1];
        }
        if constexpr (EPI) {
            float xin[2 * CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                const float2 xp = *(const float2 *)(lds + (BC ? 0 : c * W) + 2 * q);
                xin[c] = xp.x; xin[CG + c] = xp.y;
            }
            if constexpr (CG == 2) epi_pointwise(a, v, xin);
#pragma unroll
            for (int c = 0; c < CG; ++c) {               // frames past the stream's end are zeros
                sum_x[c] += xin[c] * xin[c] + xin[CG + c] * xin[CG + c];
                sum_y[c] += v[c] * v[c] + v[CG + c] * v[CG + c];
            }
        }
        if constexpr (EPI && CG == 2) {
            if (a.sink_partials != nullptr) {            // frames past the stream's end are (0, 0): r = 0, theta = 0
                polar_add(pacc, v[0], v[1]);
                polar_add(pacc, v[2], v[3]);
                continue;
            }
        }
        store_result<CG>(rdst, shape, a.stream_out, q, strideG, C, v);
    }
    if constexpr (EPI && CG == 2) {
        if (a.sink_partials != nullptr) { sink_reduce_store<NT>(a, bc, pacc, tid); return; }
    }
    if constexpr (EPI) {
        if (a.epi_normalize) {
            __shared__ double red[NT / 64][2 * CG];
            double r[2 * CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) { r[c] = (double)sum_x[c]; r[CG + c] = (double)sum_y[c]; }
#pragma unroll
            for (int i = 0; i < 2 * CG; ++i) {
#pragma unroll
                for (int sh = 32; sh > 0; sh >>= 1) r[i] += __shfl_xor(r[i], sh);
            }
            if ((tid & 63) == 0) {
#pragma unroll
                for (int i = 0; i < 2 * CG; ++i) red[tid >> 6][i] = r[i];
            }
            __syncthreads();
            if (tid < 2 * CG) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < NT / 64; ++w) t += red[w][tid];          // fixed order
                double *row = a.epi_partials + ((int64_t)bc.stream * a.tiles + bc.tile) * 2 * C;
                row[tid < CG ? c0 + tid : C + c0 + (tid - CG)] = t;
            }
        }
    }
}

// =====================================================================================
// Ordered kernel: taps in table order, the reference's association.
//   MODE 0  acc = f32(acc + f32(x*w))   bit-identical to NumPy's  out += x * w
//           (one v_pk_mul_f32 + one v_pk_add_f32 per pair; file built with -ffp-contract=off)
//   MODE 1  acc = fma(x, w, acc)
// Class-path tables add the segment loop: seg = (+0 ∓x ...) ; seg *= gain ; out += seg.
// An odd tap offset is 4-byte-misaligned for a b64 pair, so two aligned reads
// straddle it and the pair (lo.y, hi.x) is used.  (A second LDS plane shifted by
// one frame was measured slower: it halves the workgroups a CU can hold.)
// Records {w, byte offset} come 16 at a time into SGPRs, like the fast kernel's.
// =====================================================================================
template <int MODE, int R>
__device__ __forceinline__ void ordered_consume(v2f (&sb)[R], const v2f (&v)[R], float w)
{
    const v2f ww = {w, w};
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if constexpr (MODE == 0) {
            const v2f p = v[j] * ww;          // two roundings: exactly NumPy's  out += x * w
            sb[j] = sb[j] + p;
        } else {
            sb[j] = __builtin_elementwise_fma(v[j], ww, sb[j]);
        }
    }
}

// The parity branch only fetches the tap's R pairs; the accumulators are updated after the
// arms have joined, in straight-line code - updated inside the arms they become phi nodes
// and hipcc copies all 2R accumulator registers after every tap.
template <int NT, int R, int MODE>
__device__ __forceinline__ void ordered_tap(const FastTap &t, unsigned lane_addr, v2f (&sb)[R])
{
    v2f val[R];
    if ((t.off & 4) == 0) {
        issue_reads<NT, R>(val, lane_addr + (unsigned)t.off);
        wait_reads<0, R>(val);
    } else {
        // an odd offset is 4-byte-misaligned for a b64 pair: its two dwords come as two b32
        // reads straight into the halves of the pair.  (Two aligned b64 reads straddling it plus
        // one v_pk_mov_b32 per pair measured 2 % slower: fewer LDS cycles, more vector issue.)
        float a[R], b[R];
        issue_dwords<NT, R, 0>(a, lane_addr + (unsigned)t.off);
        issue_dwords<NT, R, 4>(b, lane_addr + (unsigned)t.off);
        wait_dwords<R>(a, b);
#pragma unroll
        for (int j = 0; j < R; ++j) val[j] = v2f{a[j], b[j]};
    }
    ordered_consume<MODE, R>(sb, val, t.w);
}

// EPI (CG == 2): the pointwise epilogue steps are applied before the store - the input planes are
// still intact in LDS - which saves the separate read-modify-write pass over y in exact mode.
template <int NT, int CG, int R, int MODE, bool BC = false, bool EPI = false>
__global__ __launch_bounds__(NT) void conv_ordered_kernel(const KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int T = 2 * NT * R;
    constexpr int PG = BC ? 1 : CG;
    const int tid = threadIdx.x;
    const int W = a.W;
    const BlockCoord bc = decode_block(a);
    const int C = a.C, Cx = a.Cx;
    const int c0 = bc.group * CG;
    const int cx0 = BC ? 0 : (Cx == C ? c0 : c0 % Cx);
    const int64_t t0 = (int64_t)bc.tile * T;
    const float *__restrict__ xs = a.x + bc.stream * a.n * Cx;
    float *__restrict__ ys = a.y + bc.stream * a.n * C;
    const int64_t bytes_left = ((a.n - t0) * C - c0) * 4;

    stage_window<NT, PG>(lds, xs + t0 * Cx + cx0, ((a.n - t0) * Cx - cx0) * 4, Cx, W, tid);
    __syncthreads();

    v2f out[CG][R];
    const bool has_seg = a.seg_off != nullptr;
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        const int ch = c0 + c;
        const float *pa = lds + (BC ? 0 : c * W) + 2 * tid;
        if (a.chan_flags != nullptr && (a.chan_flags[ch] & 1)) {      // unfiltered: copy through
#pragma unroll
            for (int j = 0; j < R; ++j) out[c][j] = *(const v2f *)(pa + 2 * NT * j);
            continue;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) out[c][j] = v2f{0.0f, 0.0f};
        const unsigned lane_addr = lds_addr(pa);
        int k = __builtin_amdgcn_readfirstlane(a.tap_off[ch]);
        const int k_last = __builtin_amdgcn_readfirstlane(a.tap_off[ch + 1]);
        const int s_begin = has_seg ? __builtin_amdgcn_readfirstlane(a.seg_off[ch]) : 0;
        const int nseg = has_seg ? __builtin_amdgcn_readfirstlane(a.seg_off[ch + 1]) - s_begin : 1;
        for (int s = 0; s < nseg; ++s) {
            const int kend = has_seg ? __builtin_amdgcn_readfirstlane(a.seg_end[s_begin + s]) : k_last;
            v2f sb[R];
#pragma unroll
            for (int j = 0; j < R; ++j) sb[j] = v2f{0.0f, 0.0f};
            while (k < kend) {
                FastTap t[16];
                load_taps16(a.taps_ord + k, t);          // zero-padded by 16 records
                const int m = kend - k;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (i < m) ordered_tap<NT, R, MODE>(t[i], lane_addr, sb);
                k += m < 16 ? m : 16;
            }
            if (has_seg) {          // class path: seg *= envelope (unless identity); out += seg
                if (a.apply_gain) {
                    const float gain = a.seg_gain[s_begin + s];
                    const v2f gg = {gain, gain};
#pragma unroll
                    for (int j = 0; j < R; ++j) sb[j] = sb[j] * gg;
                }
#pragma unroll
                for (int j = 0; j < R; ++j) out[c][j] = out[c][j] + sb[j];
            } else {
#pragma unroll
                for (int j = 0; j < R; ++j) out[c][j] = sb[j];
            }
        }
    }

    float *dst = ys + t0 * C + c0;
    const v4i rdst = make_rsrc(dst, bytes_left);
    const int shape = access_shape<CG>(dst, C);
    const int strideG = C / CG;
    PolarAcc pacc;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        float v[2 * CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) { v[c] = out[c][j].x; v[CG + c] = out[c][j].y; }
        if constexpr (EPI && CG == 2) {
            const int q = tid + NT * j;
            const float2 x0 = *(const float2 *)(lds + 2 * q), x1 = *(const float2 *)(lds + (BC ? 0 : W) + 2 * q);
            const float xin[4] = {x0.x, x1.x, x0.y, x1.y};
            epi_pointwise(a, v, xin);
            if (a.sink_partials != nullptr) {
                polar_add(pacc, v[0], v[1]);
                polar_add(pacc, v[2], v[3]);
                continue;
            }
        }
        store_result<CG>(rdst, shape, a.stream_out, tid + NT * j, strideG, C, v);
    }
    if constexpr (EPI && CG == 2) {
        if (a.sink_partials != nullptr) sink_reduce_store<NT>(a, bc, pacc, tid);
    }
}

// =====================================================================================
// Fallback without LDS staging, for halos that do not fit a workgroup's LDS
// (very long FIRs): one lane per (frame, channel), taps gathered through L1/L2,
// table order and association (MODE as in the ordered kernel).
// =====================================================================================
constexpr int kDirectThreads = 256;

template <int MODE>
__device__ __forceinline__ float tap_op(float acc, float v, float w)
{
    if constexpr (MODE == 0) {
        float p = v * w;          // -ffp-contract=off: two roundings
        return acc + p;
    } else {
        return __builtin_fmaf(v, w, acc);
    }
}

template <int MODE>
__global__ __launch_bounds__(kDirectThreads) void conv_direct_kernel(const KArgs a)
{
    const int64_t per_stream = a.n * a.C;
    const int64_t total = per_stream * (int64_t)a.tiles;   // tiles carries the batch here
    for (int64_t e = (int64_t)blockIdx.x * kDirectThreads + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * kDirectThreads) {
        const int64_t b = e / per_stream;
        const int64_t r = e - b * per_stream;
        const int64_t n0 = r / a.C;
        const int ch = (int)(r - n0 * a.C);
        const int Cx = a.Cx, cx = ch % Cx;
        const float *__restrict__ xs = a.x + b * a.n * Cx;
        if (a.chan_flags != nullptr && (a.chan_flags[ch] & 1)) { a.y[e] = xs[n0 * Cx + cx]; continue; }
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
                if (m >= a.n) continue;          // the slice does not reach this output: the term DROPS (decorrelation.py:656-658)
                sb = tap_op<MODE>(sb, xs[m * Cx + cx], tp.w);
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

// =====================================================================================
// Promoting form of the function path.  NumPy multiplies a float64 signal (or any signal by
// a float64 filter such as VelvetNoise.FIR) in float64 and adds that product to the float32
// output in float64, rounding to float32 at every tap
// (decorrelation.py:656-658: out[:N-i] += x[i:] * value):
//     acc = f32( f64(acc) + f64(x) * f64(w) )
// A plain gather, one lane per output sample, table order: this path exists for parity with
// that corner of the reference, not for speed.
// =====================================================================================
struct PArgs {
    const void *__restrict__ x;        // [batch][n][C] float32 or float64
    float *__restrict__ y;
    const int32_t *__restrict__ tap_off;
    const int32_t *__restrict__ idx;
    const double *__restrict__ w;
    int64_t n, total;                  // total = batch * n * C
    int32_t C, x_is_f64;
};

__global__ __launch_bounds__(kDirectThreads) void conv_promote_kernel(const PArgs a)
{
    const int64_t per_stream = a.n * a.C;
    for (int64_t e = (int64_t)blockIdx.x * kDirectThreads + threadIdx.x; e < a.total;
         e += (int64_t)gridDim.x * kDirectThreads) {
        const int64_t b = e / per_stream, r = e - b * per_stream;
        const int64_t n0 = r / a.C;
        const int ch = (int)(r - n0 * a.C);
        float acc = 0.0f;
        for (int k = a.tap_off[ch]; k < a.tap_off[ch + 1]; ++k) {
            const int64_t m = n0 + a.idx[k];
            if (m >= a.n) continue;                                   // the slice does not reach this output
            const int64_t at = b * per_stream + m * a.C + ch;
            const double xv = a.x_is_f64 ? ((const double *)a.x)[at] : (double)((const float *)a.x)[at];
            const double p = xv * a.w[k];
            acc = (float)((double)acc + p);
        }
        a.y[e] = acc;
    }
}

}  // namespace vnd
