// What the FP32 vector pipe of an MI355X CU really issues: v_pk_fma_f32 (two FMAs per lane) against v_fma_f32, as long chains
// of INDEPENDENT accumulators, no memory traffic - per wave-instruction cycles from s_memtime (shader cycles), and chip TFLOP/s
// from the wall clock, at 1, 2 and 3 waves per SIMD.  The question behind it: the dense tables (cfg3: 128 taps) run at 77 TFLOP/s
// = 0.49 of the 157.3 TFLOP/s "FP32 vector peak" - is that half of what the pipe can do, or all of it?
//   weight operand as in the product kernels: an SGPR pair with op_sel_hi:[1,0,1] (PK_S), or a VGPR pair (PK_V)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/pk_fma_rate tools/micro/pk_fma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int NACC = 16;        // independent accumulators (pairs for the packed forms)
constexpr int INNER = 64;       // unrolled instructions per accumulator set and loop trip: NACC * INNER instructions per trip

// KIND 0: v_pk_fma_f32 acc, x, s[w:w+1] (SGPR weight, broadcast low), acc      1: v_pk_fma_f32 with a VGPR weight pair
//      2: v_fma_f32 acc, x, s_w, acc (2 x NACC single accumulators)            3: v_pk_add_f32      4: v_pk_mul_f32 into acc (acc = acc * w)
template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float *out, long long *cycles, int trips, float w0)
{
    v2f acc[NACC];
    float sacc[2 * NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) { acc[k] = v2f{(float)threadIdx.x, (float)k}; sacc[2 * k] = (float)k; sacc[2 * k + 1] = (float)threadIdx.x; }
    const v2f x = {1.0f + 1e-7f * threadIdx.x, 1.0f - 1e-7f * threadIdx.x};
    v2f wv = {w0, w0};
    asm volatile("" : "+v"(wv));
    float ws = __builtin_amdgcn_readfirstlane(w0);
    const unsigned wbits = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(w0));
    const unsigned long long wpair = ((unsigned long long)wbits << 32) | wbits;          // an SGPR pair, as the product kernels' weights
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < INNER; ++i) {
#pragma unroll
            for (int k = 0; k < NACC; ++k) {
                if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(x), "s"(wpair));
                else if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(x), "v"(wv));
                else if constexpr (KIND == 2) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sacc[2 * k]) : "v"(x.x), "s"(ws));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sacc[2 * k + 1]) : "v"(x.y), "s"(ws));
                } else if constexpr (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[k]) : "v"(x));
                else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[k]) : "v"(wv));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NACC; ++k) s += acc[k].x + acc[k].y + sacc[2 * k] + sacc[2 * k + 1];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

// The product's tap phase in miniature: one conflict-free ds_read_b128 (64 consecutive 16-byte pieces per wave, base register +
// immediate) kept LA reads ahead, F packed FMAs on each chunk (its two halves, SGPR-pair weights, NACC accumulators in turn),
// a sched_barrier per chunk as in the generated code - cfg3 runs 5.6 FMAs per read, cfg2 2.9.  BATCH: reads issued two at a time.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f lds_v4f;
#define RD(base, imm) (*(const volatile lds_v4f *)((base) + (imm)))
template <int F, int LA, int BATCH, int G, bool SB_PER_BATCH = false>
__global__ __launch_bounds__(256) void mix_kernel(float *out, int trips, float w0)
{
    extern __shared__ __attribute__((aligned(16))) float lds_generic[];
    __attribute__((address_space(3))) char *base = (__attribute__((address_space(3))) char *)(__attribute__((address_space(3))) float *)lds_generic + threadIdx.x * 16;
    for (int i = threadIdx.x; i < 8 * 1024; i += 256) lds_generic[i] = 1.0f + 1e-6f * i;
    __syncthreads();
    v2f acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = v2f{(float)threadIdx.x, (float)k};
    const float w1 = __builtin_amdgcn_readfirstlane(w0), w2 = __builtin_amdgcn_readfirstlane(w0 * 0.5f);
    // G: chunks per loop trip, fully unrolled - 32 is a 2 KB loop body, 768 is 43 KB of straight-line code (the product's tile)
    constexpr int RING = LA + BATCH;
    v4f q[RING];
#pragma unroll
    for (int k = 0; k < LA; ++k) q[k] = RD(base, (k % 8) * 4096);
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g % BATCH == 0) {
#pragma unroll
                for (int b = 0; b < BATCH; ++b) q[(g + LA + b) % RING] = RD(base, ((g + LA + b) % 8) * 4096);
            }
            const v4f c = q[g % RING];
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int k = (g * F + f) % NACC;
                const v2f x = (f & 1) ? v2f{c.z, c.w} : v2f{c.x, c.y};
                acc[k] = __builtin_elementwise_fma(x, (f & 2) ? v2f{w2, w2} : v2f{w1, w1}, acc[k]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NACC; ++k) s += acc[k].x + acc[k].y;
    if (s == 123.456f) out[0] = s;
}

// The same mix as STRAIGHT-LINE code of the product's size: hipcc stops unrolling mix_kernel's chunk loop long before 43 KB (it
// falls back to s_set_gpr_idx moves), so the body is repeated textually: a block of 5 chunks (ring of 5 reads, 4 ahead, 6 packed
// FMAs each on 15 accumulators in turn) x REPS.  REPS 1: 280 B ... 154: 42 KB per loop trip.
#define MIX_CHUNK(g)                                                                                             \
    q[((g) + 4) % 5] = RD(base, (((g) + 4) % 8) * 4096);                                                         \
    acc[((g) * 6 + 0) % 15] = __builtin_elementwise_fma(v2f{q[(g) % 5].x, q[(g) % 5].y}, v2f{w1, w1}, acc[((g) * 6 + 0) % 15]); \
    acc[((g) * 6 + 1) % 15] = __builtin_elementwise_fma(v2f{q[(g) % 5].z, q[(g) % 5].w}, v2f{w1, w1}, acc[((g) * 6 + 1) % 15]); \
    acc[((g) * 6 + 2) % 15] = __builtin_elementwise_fma(v2f{q[(g) % 5].x, q[(g) % 5].y}, v2f{w2, w2}, acc[((g) * 6 + 2) % 15]); \
    acc[((g) * 6 + 3) % 15] = __builtin_elementwise_fma(v2f{q[(g) % 5].z, q[(g) % 5].w}, v2f{w2, w2}, acc[((g) * 6 + 3) % 15]); \
    acc[((g) * 6 + 4) % 15] = __builtin_elementwise_fma(v2f{q[(g) % 5].x, q[(g) % 5].y}, v2f{w1, w1}, acc[((g) * 6 + 4) % 15]); \
    acc[((g) * 6 + 5) % 15] = __builtin_elementwise_fma(v2f{q[(g) % 5].z, q[(g) % 5].w}, v2f{w2, w2}, acc[((g) * 6 + 5) % 15]); \
    __builtin_amdgcn_sched_barrier(0);
#define MIX_BLOCK MIX_CHUNK(0) MIX_CHUNK(1) MIX_CHUNK(2) MIX_CHUNK(3) MIX_CHUNK(4)
#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int SIZE>      // 0: one block (280 B), 1: 16 blocks (4.4 KB), 2: 64 (17.5 KB), 3: 144 (39 KB)
__global__ __launch_bounds__(256) void straight_kernel(float *out, int trips, float w0)
{
    extern __shared__ __attribute__((aligned(16))) float lds_generic[];
    __attribute__((address_space(3))) char *base = (__attribute__((address_space(3))) char *)(__attribute__((address_space(3))) float *)lds_generic + threadIdx.x * 16;
    for (int i = threadIdx.x; i < 8 * 1024; i += 256) lds_generic[i] = 1.0f + 1e-6f * i;
    __syncthreads();
    v2f acc[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) acc[k] = v2f{(float)threadIdx.x, (float)k};
    const float w1 = __builtin_amdgcn_readfirstlane(w0), w2 = __builtin_amdgcn_readfirstlane(w0 * 0.5f);
    v4f q[5];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = RD(base, (k % 8) * 4096);
    for (int t = 0; t < trips; ++t) {
        if constexpr (SIZE == 0) { MIX_BLOCK }
        else if constexpr (SIZE == 1) { REP16(MIX_BLOCK) }
        else if constexpr (SIZE == 2) { REP64(MIX_BLOCK) }
        else { REP64(MIX_BLOCK) REP64(MIX_BLOCK) REP16(MIX_BLOCK) }
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 15; ++k) s += acc[k].x + acc[k].y;
    if (s == 123.456f) out[0] = s;
}

template <int SIZE>
static void run_straight(int waves_per_simd, float *out)
{
    const int blocks_of_5 = SIZE == 0 ? 1 : SIZE == 1 ? 16 : SIZE == 2 ? 64 : 144;
    const int trips = 2304 / blocks_of_5;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) straight_kernel<SIZE><<<blocks, 256, 32768>>>(out, trips, 0.999f);
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) straight_kernel<SIZE><<<blocks, 256, 32768>>>(out, trips, 0.999f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1));
        ms.push_back(m / 10);
    }
    std::sort(ms.begin(), ms.end());
    const double t = ms[2] * 1e-3;
    const double flop = (double)blocks * 256 * trips * blocks_of_5 * 5 * 6 * 4;
    printf("straight-line: (ds_read_b128 + 6 v_pk_fma_f32) x %4d per loop trip = %5.1f KB of code, waves/SIMD %d: %8.4f ms  %6.1f TFLOP/s\n",
           blocks_of_5 * 5, blocks_of_5 * 5 * 7 * 8 / 1024.0, waves_per_simd, ms[2], flop / t / 1e12);
}

template <int F, int LA, int BATCH, int G = 32, bool SBB = false>
static void run_mix(int waves_per_simd, float *out)
{
    const int trips = 400 * 32 / G;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) mix_kernel<F, LA, BATCH, G, SBB><<<blocks, 256, 32768>>>(out, trips, 0.999f);
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) mix_kernel<F, LA, BATCH, G, SBB><<<blocks, 256, 32768>>>(out, trips, 0.999f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1));
        ms.push_back(m / 10);
    }
    std::sort(ms.begin(), ms.end());
    const double t = ms[2] * 1e-3;
    const double flop = (double)blocks * 256 * trips * G * F * 4;
    const double lds_bytes = (double)blocks * 256 * trips * G * 16;
    printf("%sds_read_b128 + %2d v_pk_fma_f32, %d reads ahead, batches of %d, loop body %5.1f KB, waves/SIMD %d: %8.4f ms  %6.1f TFLOP/s  LDS %5.1f TB/s\n", SBB ? "[one sched_barrier per batch] " : "", F, LA, BATCH,
           G * (F + 1) * 8 / 1024.0, waves_per_simd, ms[2], flop / t / 1e12, lds_bytes / t / 1e12);
}

template <int KIND>
static void run(const char *name, int waves_per_simd, float *out, long long *cyc)
{
    const int trips = 200;
    const int blocks = 256 * waves_per_simd;          // 256-thread workgroups: 4 waves, one per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) rate_kernel<KIND><<<blocks, 256>>>(out, cyc, trips, 0.999f);
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) rate_kernel<KIND><<<blocks, 256>>>(out, cyc, trips, 0.999f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1));
        ms.push_back(m / 10);
    }
    std::sort(ms.begin(), ms.end());
    long long c = 0;
    CK(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));
    const double instr_per_wave = (double)trips * INNER * NACC * (KIND == 2 ? 2 : 1);
    const double fmas = (double)blocks * 256 * trips * INNER * NACC * 2;        // lane-FMAs (or adds / multiplies) per launch
    const double flop = fmas * ((KIND == 3 || KIND == 4) ? 1.0 : 2.0);
    // s_memtime ticks are shader cycles (MI355X_MICROARCH.md): ticks of wave 0 / its instructions / the waves sharing its SIMD
    // = SIMD cycles per wave-instruction; ticks / wall time = the clock the chip held
    const double t = ms[2] * 1e-3;
    printf("%-30s waves/SIMD %d: %8.4f ms  %7.1f T%s/s   SIMD cycles per wave-instruction %.2f   clock %.2f GHz\n", name,
           waves_per_simd, ms[2], flop / t / 1e12, (KIND == 3 || KIND == 4) ? "OP" : "FLOP", (double)c / instr_per_wave / waves_per_simd,
           (double)c / t / 1e9);
}

int main()
{
    float *out; long long *cyc;
    CK(hipMalloc(&out, 1024)); CK(hipMalloc(&cyc, 64));
    for (int w = 1; w <= 3; ++w) {
        run<0>("v_pk_fma_f32 (SGPR weight)", w, out, cyc);
        run<1>("v_pk_fma_f32 (VGPR weight)", w, out, cyc);
        run<2>("v_fma_f32 x 2 (SGPR weight)", w, out, cyc);
        run<3>("v_pk_add_f32", w, out, cyc);
        run<4>("v_pk_mul_f32", w, out, cyc);
    }
    for (int w = 1; w <= 3; ++w) {
        run_mix<3, 4, 1>(w, out);
        run_mix<6, 4, 1>(w, out);
        run_mix<6, 8, 1>(w, out);
        run_mix<6, 4, 2>(w, out);
        run_mix<12, 4, 1>(w, out);
        run_mix<24, 4, 1>(w, out);
        run_mix<6, 4, 2, 32, true>(w, out);
        run_mix<6, 8, 4, 32, true>(w, out);
        run_mix<3, 4, 2, 32, true>(w, out);
        run_mix<3, 8, 4, 32, true>(w, out);
        run_straight<0>(w, out);
        run_straight<1>(w, out);
        run_straight<2>(w, out);
        run_straight<3>(w, out);
    }
    return 0;
}
