// Where in the 160 KB of LDS do a workgroup's waves read?  cfg5's octet kernel streams its window reads at ~8 clocks per ds_read_b128 and
// CU, pk_fma_rate.hip's two 32 KB workgroups at 4.3.  One workgroup of NW waves, every wave a stream of conflict-free ds_read_b128 (64
// consecutive 16-byte pieces, base + immediate, 4 in flight, nothing else), the waves' bases laid out by MODE over a region of `span` bytes.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_regions tools/micro/lds_regions.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f lds_v4f;
typedef __attribute__((address_space(3))) char lds_char;
#define RD(base, imm) (*(const volatile lds_v4f *)((base) + (imm)))

template <int NT>
__global__ __launch_bounds__(NT) void reads_kernel(float *out, int trips, int wave_stride, int lane_stride, int step)
{
    extern __shared__ __attribute__((aligned(16))) float lds_generic[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds_char *base = (lds_char *)(__attribute__((address_space(3))) float *)lds_generic + (wave_stride >= 0 ? wave * wave_stride : (wave & 1) * 80 * 1024 + (wave >> 1) * -wave_stride) + lane * lane_stride;
    v4f q[4];
    v4f acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = RD(base, k * 1024);
    for (int t = 0; t < trips; ++t) {
        lds_char *b = base + (t & 7) * step;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const v4f c = q[g % 4];
            q[g % 4] = RD(b, (g % 8) * 1024);
            acc += c;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

// the generated tap function's shape: NBASE per-lane base registers 128 B apart, immediates = entry * 16 + chunk plane * 2528, descending
template <int NT>
__global__ __launch_bounds__(NT) void bases_kernel(float *out, int trips, int wave_stride, int ring = 0, int start = 0)
{
    extern __shared__ __attribute__((aligned(16))) float lds_generic[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds_char *b[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        // ring > 0: the lanes' entries wrap as in the product - base k holds entry (start + lane + 8 k) mod ring
        const int ent = ring > 0 ? (start + lane + 8 * k) % ring : lane + ((k * 8 + 64) % 86);
        b[k] = (lds_char *)(__attribute__((address_space(3))) float *)lds_generic + wave * wave_stride + ent * 16;
        asm volatile("" : "+v"(b[k]));
    }
    v4f q[5];
    v4f acc = {0, 0, 0, 0};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = RD(b[10], (7 - k) * 2528 + 5 * 16);
#pragma unroll
        for (int g = 0; g < 80; ++g) {
            const int n = g + 4, kb = 10 - (n / 8) % 11, r = 7 - n % 8;
            q[n % 5] = RD(b[kb], r * 2528 + ((n * 3) % 8) * 16);
            acc += q[g % 5];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

// ... as STRAIGHT-LINE code, 336 reads per trip (a tile's tap function), the same function for every wave or one per wave (the octet
// kernel runs eight different ones): is it the instruction stream?
#define SR_STEP(g, W) q[((g) + 4) % 5] = RD(b[10 - (((g) + W) % 11)], (7 - ((g) + W) % 8) * 2528 + ((((g) + W) * 3) % 8) * 16); acc += q[(g) % 5]; __builtin_amdgcn_sched_barrier(0);
#define SR_BLOCK(W) SR_STEP(0, W) SR_STEP(1, W) SR_STEP(2, W) SR_STEP(3, W) SR_STEP(4, W)
#define SR2(x) x x
#define SR4(x) SR2(x) SR2(x)
#define SR16(x) SR4(x) SR4(x) SR4(x) SR4(x)
#define SR64(x) SR16(x) SR16(x) SR16(x) SR16(x)
template <int W>
__device__ __forceinline__ void straight_reads(lds_char *const (&b)[11], v4f &acc)
{
    v4f q[5];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = RD(b[10], (7 - k) * 2528 + 5 * 16 + W * 16);
    SR64(SR_BLOCK(W)) SR2(SR_BLOCK(W)) SR_BLOCK(W)          // 67 blocks of 5 reads
}
template <int NT, bool PER_WAVE, int NPAD = 0>
__global__ __launch_bounds__(NT) void straight_kernel(float *out, int trips, int wave_stride)
{
    float pad[NPAD + 1];                                  // (NPAD registers held live through the loop: the product's lanes carry ~220)
#pragma unroll
    for (int i = 0; i < NPAD; ++i) pad[i] = (float)(threadIdx.x + i);
    extern __shared__ __attribute__((aligned(16))) float lds_generic[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    lds_char *b[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        b[k] = (lds_char *)(__attribute__((address_space(3))) float *)lds_generic + wave * wave_stride + lane * 16 + ((k * 8 * 16 + 64 * 16) % (86 * 16));
        asm volatile("" : "+v"(b[k]));
    }
    v4f acc = {0, 0, 0, 0};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < NPAD; ++i) asm volatile("" : "+v"(pad[i]));
        if constexpr (PER_WAVE) {
            switch (wave) {
            case 0: straight_reads<0>(b, acc); break; case 1: straight_reads<1>(b, acc); break; case 2: straight_reads<2>(b, acc); break;
            case 3: straight_reads<3>(b, acc); break; case 4: straight_reads<4>(b, acc); break; case 5: straight_reads<5>(b, acc); break;
            case 6: straight_reads<6>(b, acc); break; default: straight_reads<7>(b, acc); break;
            }
        } else {
            straight_reads<0>(b, acc);
        }
    }
    float ps = 0.0f;
#pragma unroll
    for (int i = 0; i < NPAD; ++i) ps += pad[i];
    if (acc.x + acc.y + acc.z + acc.w + ps == 123.456f) out[0] = acc.x;
}
template <int NT, bool PER_WAVE, int NPAD = 0>
static void run_straight(const char *what, int grid, int lds_bytes, int wave_stride)
{
    float *out;
    CK(hipMalloc(&out, 4));
    CK(hipFuncSetAttribute((const void *)straight_kernel<NT, PER_WAVE, NPAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int trips = 200;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    straight_kernel<NT, PER_WAVE, NPAD><<<grid, NT, lds_bytes>>>(out, trips, wave_stride);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        straight_kernel<NT, PER_WAVE, NPAD><<<grid, NT, lds_bytes>>>(out, trips, wave_stride);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double reads = (double)grid * (NT / 64) * trips * 335;
    printf("%-92s %8.4f ms  %6.1f TB/s over the chip = %6.1f B per ns and CU\n", what, best, reads * 1024 / best / 1e9, reads * 1024 / best / 1e6 / 256);
    CK(hipFree(out));
}

template <int NT>
static void run_bases(const char *what, int grid, int lds_bytes, int wave_stride, int ring = 0, int start = 0)
{
    float *out;
    CK(hipMalloc(&out, 4));
    CK(hipFuncSetAttribute((const void *)bases_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int trips = 800;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    bases_kernel<NT><<<grid, NT, lds_bytes>>>(out, trips, wave_stride, ring, start);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        bases_kernel<NT><<<grid, NT, lds_bytes>>>(out, trips, wave_stride, ring, start);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double reads = (double)grid * (NT / 64) * trips * 84;
    printf("%-92s %8.4f ms  %6.1f TB/s over the chip = %6.1f B per ns and CU\n", what, best, reads * 1024 / best / 1e9, reads * 1024 / best / 1e6 / 256);
    CK(hipFree(out));
}

template <int NT>
static void run(const char *what, int grid, int lds_bytes, int wave_stride, int lane_stride, int step)
{
    float *out;
    CK(hipMalloc(&out, 4));
    CK(hipFuncSetAttribute((const void *)reads_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int trips = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    reads_kernel<NT><<<grid, NT, lds_bytes>>>(out, trips, wave_stride, lane_stride, step);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        reads_kernel<NT><<<grid, NT, lds_bytes>>>(out, trips, wave_stride, lane_stride, step);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double reads = (double)grid * (NT / 64) * trips * 16;
    printf("%-92s %8.4f ms  %6.1f TB/s over the chip = %6.1f B per ns and CU\n", what, best, reads * 1024 / best / 1e9, reads * 1024 / best / 1e6 / 256);
    CK(hipFree(out));
}

int main()
{
    const int cus = 256;
    printf("one workgroup per CU unless said; ds_read_b128 streams, 4 in flight per wave, no other work\n");
    run<256>("4 waves, 32 KB of LDS, waves 1 KB apart (pk_fma_rate's layout)", cus, 32 * 1024, 1024, 16, 0);
    run<256>("4 waves, 32 KB, TWO workgroups per CU", 2 * cus, 32 * 1024, 1024, 16, 0);
    run<256>("4 waves, 80 KB each, TWO workgroups per CU (the stereo kernels' residency), waves 16 KB apart", 2 * cus, 80 * 1024, 16 * 1024, 16, 0);
    run<512>("8 waves, 32 KB, waves 1 KB apart", cus, 32 * 1024, 1024, 16, 0);
    run<512>("8 waves, 158 KB, waves 1 KB apart (all in the first 16 KB)", cus, 158 * 1024, 1024, 16, 0);
    run<512>("8 waves, 158 KB, waves 19 KB apart (a channel's planes each: the octet kernel)", cus, 158 * 1024, 19 * 1024, 16, 0);
    run<512>("8 waves, 158 KB, waves 19 KB apart, the stream walking 8 x 1264 B further every 16 reads", cus, 158 * 1024, 19 * 1024, 16, 1264);
    run<512>("8 waves, 158 KB, waves 9.5 KB apart (first half of LDS only)", cus, 158 * 1024, 9728, 16, 0);
    run<512>("8 waves, 158 KB, even waves 16 KB apart in the first half, odd waves 80 KB further", cus, 158 * 1024, -16 * 1024, 16, 0);
    run<256>("4 waves, 158 KB, waves 38 KB apart", cus, 158 * 1024, 38 * 1024, 16, 0);
    run<512>("8 waves, 158 KB, waves 19 KB apart, lanes 32 B apart (2-way conflicts)", cus, 158 * 1024, 19 * 1024, 32, 0);
    // a wave's 1 KB starting at any 16-byte slot (the ring position of its first lane), planes 2528 B apart as in the octet kernel
    run<512>("8 waves, 158 KB, waves 19 KB + 16 B apart (starts 16 B past a 1 KB boundary, 32 B, ...)", cus, 158 * 1024, 19 * 1024 + 16, 16, 0);
    run<512>("8 waves, 158 KB, waves 19 KB + 112 B apart", cus, 158 * 1024, 19 * 1024 + 112, 16, 0);
    run_bases<512>("8 waves, 158 KB: eleven per-lane bases, immediates over eight chunk planes 2528 B apart (the tap function's reads)", cus, 158 * 1024, 19 * 1024);
    run_bases<256>("4 waves, the same", cus, 158 * 1024, 38 * 1024);
    run_bases<512>("8 waves: ... the lanes' entries wrapping in a ring of 144 (a multiple of 16), tile starting at entry 0", cus, 158 * 1024, 19 * 1024, 144, 0);
    run_bases<512>("8 waves: ... ring of 144, tile starting at entry 48 (some bases straddle the ring's end)", cus, 158 * 1024, 19 * 1024, 144, 48);
    run_bases<512>("8 waves: ... ring of 144, tile starting at entry 112", cus, 158 * 1024, 19 * 1024, 144, 112);
    run_bases<512>("8 waves: ... ring of 149 (not a multiple of 16), tile starting at entry 43", cus, 158 * 1024, 19 * 1024, 149, 43);
    run_straight<512, false>("8 waves, 158 KB: the same reads as 336 straight-line instructions per trip, one function for all waves", cus, 158 * 1024, 19 * 1024);
    run_straight<512, true>("8 waves, 158 KB: ... a function of its own per wave (8 x 336 reads of code)", cus, 158 * 1024, 19 * 1024);
    run_straight<512, true, 190>("8 waves, 158 KB: ... and 190 more registers held live per lane (the product's lanes carry ~220)", cus, 158 * 1024, 19 * 1024);
    return 0;
}
