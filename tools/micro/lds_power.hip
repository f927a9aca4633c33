// cfg5's window reads run at 385 B per ns and CU inside the octet kernel and at 550-600 in lds_regions.hip's imitation of them.  Two things
// differ that lds_regions.hip never looked at: WHAT the LDS holds (the probe read whatever the array held - the same words over and over;
// the kernel reads audio) and HOW LONG the stream runs (1 ms launches never meet the board's power cap; the bench's launches run for seconds).
// This probe runs the octet kernel's read shape - 8 waves, 158 KB, a channel's planes 19 KB apart, conflict-free ds_read_b128, 4 in
// flight per wave - for ~1.5 s per case, back to back, over zeros / one repeated word / uniform random floats / random floats in [-1, 1)
// (audio-like: sign and mantissa bits toggle, the exponent mostly does not), alone and with packed FMAs consuming what is read, and
// samples the card's hwmon power and shader clock every 10 ms from a host thread.  Reported per case: B per ns and CU, the median clock,
// B per CLOCK and CU (the array's peak is 256), board watts.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_power tools/micro/lds_power.hip -lpthread
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>
#include <unistd.h>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v4f lds_v4f;
typedef __attribute__((address_space(3))) char lds_char;
#define RD(base, imm) (*(const volatile lds_v4f *)((base) + (imm)))

// FILL: 0 zeros, 1 one word repeated, 2 random bits as floats in [1, 2) (mantissa only), 3 uniform [-1, 1) (sign + mantissa + a few exponent bits)
// FMAS: packed FMAs per 16-byte read (0: reads alone; 3 = the octet kernel's 2048 FMAs per 333 reads, lane and tile, as v_pk_fma_f32)
// OP (round 6): what consumes a read - 0 v_pk_fma_f32 (acc += x * w), 1 v_pk_add_f32 (acc +- x: the reference's class-path association,
// decorrelation.py:402-414 - add and subtract inside a segment, ONE multiply per segment; the sign is a neg_lo / neg_hi source modifier),
// 2 three v_pk_add_f32 to one v_pk_fma_f32 (a 30-tap table's mix once the segment boundaries of both accumulator sets are FMAs)
template <int NT, int FMAS, int OP = 0>
__global__ __launch_bounds__(NT) void reads_kernel(float *out, int trips, int wave_stride, int fill, unsigned seed)
{
    extern __shared__ __attribute__((aligned(16))) float lds_generic[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __attribute__((address_space(3))) float *lds = (__attribute__((address_space(3))) float *)lds_generic;
    const int words = 158 * 1024 / 4;
    unsigned s = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u + 12345u;
    for (int i = threadIdx.x; i < words; i += NT) {
        s = s * 1664525u + 1013904223u;
        const unsigned r = s ^ (s >> 15);
        float v = 0.0f;
        if (fill == 1) v = 0.7236f;
        else if (fill == 2) v = __uint_as_float(0x3f800000u | (r >> 9));
        else if (fill == 3) v = (float)(int)(r >> 8) * (1.0f / 8388608.0f) - 1.0f;
        lds[i] = v;
    }
    __syncthreads();
    lds_char *base = (lds_char *)lds + wave * wave_stride + lane * 16;
    v4f q[4];
    v2f acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = v2f{0.0f, 0.0f};
    const v2f w = {0.3f + lane * 1e-3f, -0.2f};
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = RD(base, k * 1024);
    for (int t = 0; t < trips; ++t) {
        lds_char *b = base + (t & 7) * 1264;                       // (the stream walks through the channel's planes)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const v4f c = q[g % 4];
            q[g % 4] = RD(b, (g % 8) * 1024 + (g / 8) * 16);
            if constexpr (FMAS == 0) {
                acc[0] += v2f{c.x, c.y}; acc[1] += v2f{c.z, c.w};
            } else {
#pragma unroll
                for (int f = 0; f < FMAS; ++f) {
                    const v2f xv = (f & 1) ? v2f{c.z, c.w} : v2f{c.x, c.y};
                    if (OP == 0 || (OP == 2 && f % 4 == 3)) acc[f % 8] = __builtin_elementwise_fma(xv, w, acc[f % 8]);
                    else if ((f >> 1) & 1) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[1,0] neg_hi:[1,0]" : "=v"(acc[f % 8]) : "v"(xv), "0"(acc[f % 8]));
                    else asm("v_pk_add_f32 %0, %1, %2" : "=v"(acc[f % 8]) : "v"(xv), "0"(acc[f % 8]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += acc[k].x + acc[k].y;
    if (sum == 123.456f) out[0] = sum;
}

// The candidate data flow "windows from the neighbouring lanes' registers": an FMA whose x operand is another lane's register costs
// nothing extra to issue when the shift is a DPP modifier of the FMA itself (v_fmac_f32 ... row_shr:1) - but DPP exists on the single
// v_fmac_f32 only, not on v_pk_fma_f32: one FMA per lane and instruction instead of two.  KIND 0: v_pk_fma_f32 on held registers;
// 1: v_fmac_f32 with a row_shr:1 operand; 2: v_mov_b32 row_shr:1 into a temporary + v_pk_fma_f32 (a shifted PAIR costs two moves).
// Round 6 - is an ADD cheaper than an FMA in watts?  KIND 4: v_pk_add_f32 acc += x; 5: v_pk_fma_f32 with the product's sign alternating
// per pass (neg modifier: the accumulator walks instead of growing); 6: v_pk_add_f32 with the same alternating sign; 7: v_pk_mul_f32.
template <int KIND>
__global__ __launch_bounds__(512) void fma_kernel(float *out, int trips, unsigned seed)
{
    const int lane = threadIdx.x;
    unsigned s = seed * 2654435761u + lane * 40503u + blockIdx.x * 9176u + 12345u;
    float x[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) { s = s * 1664525u + 1013904223u; x[j] = (float)(int)((s ^ (s >> 15)) >> 8) * (1.0f / 8388608.0f) - 1.0f; }
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.0f;
    const float w = 0.37f;
    for (int t = 0; t < trips; t += 8) {
        // (eight passes per loop trip: a 16-instruction loop body spends half its time in the branch - first version of this probe)
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
        for (int j = 0; j < 32; j += 2) {
            if constexpr (KIND == 0) {
                v2f a = {acc[j], acc[j + 1]};
                a = __builtin_elementwise_fma(v2f{x[j], x[j + 1]}, v2f{w, w}, a);
                acc[j] = a.x; acc[j + 1] = a.y;
            } else if constexpr (KIND >= 4) {
                v2f a = {acc[j], acc[j + 1]};
                const v2f xv = {x[j], x[j + 1]}, wv = {w, w};
                if constexpr (KIND == 4) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a) : "v"(xv));
                else if constexpr (KIND == 5) { if (rep & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(a) : "v"(xv), "v"(wv));
                                                else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(xv), "v"(wv)); }
                else if constexpr (KIND == 6) { if (rep & 1) asm volatile("v_pk_add_f32 %0, %1, %0 neg_lo:[1,0] neg_hi:[1,0]" : "+v"(a) : "v"(xv));
                                                else asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a) : "v"(xv)); }
                else asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(xv), "v"(wv));
                acc[j] = a.x; acc[j + 1] = a.y;
            } else if constexpr (KIND == 3) {
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j]) : "v"(x[j]), "v"(w));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j + 1]) : "v"(x[j + 1]), "v"(w));
            } else if constexpr (KIND == 1) {
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[j]) : "v"(x[j]), "v"(w));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[j + 1]) : "v"(x[j + 1]), "v"(w));
            } else {
                const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[j]), 0x111, 0xf, 0xf, true));
                const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[j + 1]), 0x111, 0xf, 0xf, true));
                v2f a = {acc[j], acc[j + 1]};
                a = __builtin_elementwise_fma(v2f{s0, s1}, v2f{w, w}, a);
                acc[j] = a.x; acc[j + 1] = a.y;
            }
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) asm volatile("" : "+v"(acc[j]), "+v"(x[j]));
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 32; ++j) sum += acc[j];
    if (sum == 123.456f) out[0] = sum;
}

struct Sampler {
    std::string power, clock;
    std::atomic<bool> stop{false};
    std::vector<double> watts, mhz;
    std::thread th;
    static long read_long(const std::string &path)
    {
        FILE *f = fopen(path.c_str(), "r");
        if (!f) return -1;
        long v = -1;
        if (fscanf(f, "%ld", &v) != 1) v = -1;
        fclose(f);
        return v;
    }
    bool find(int device)
    {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) return false;
        for (char *p = bus; *p; ++p) *p = (char)tolower(*p);
        DIR *d = opendir("/sys/class/drm");
        if (!d) return false;
        while (dirent *e = readdir(d)) {
            if (strncmp(e->d_name, "card", 4) != 0 || strchr(e->d_name, '-')) continue;
            const std::string dev = std::string("/sys/class/drm/") + e->d_name + "/device";
            char real[512];
            if (!realpath(dev.c_str(), real) || !strstr(real, bus)) continue;
            DIR *h = opendir((dev + "/hwmon").c_str());
            if (!h) continue;
            while (dirent *he = readdir(h)) {
                if (strncmp(he->d_name, "hwmon", 5) != 0) continue;
                const std::string base = dev + "/hwmon/" + he->d_name + "/";
                power = access((base + "power1_input").c_str(), R_OK) == 0 ? base + "power1_input" : base + "power1_average";
                clock = base + "freq1_input";
            }
            closedir(h);
        }
        closedir(d);
        return !power.empty() && read_long(power) >= 0;
    }
    void start() { stop = false; watts.clear(); mhz.clear(); th = std::thread([this] {
        while (!stop) {
            const long w = read_long(power), c = read_long(clock);
            if (w >= 0) watts.push_back(w / 1e6);
            if (c >= 0) mhz.push_back(c / 1e6);
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        } }); }
    void finish() { stop = true; th.join(); }
    static double median_tail(std::vector<double> v)
    {
        if (v.empty()) return 0.0;
        v.erase(v.begin(), v.begin() + v.size() / 2);               // the second half of the run: the power figure is a slow average
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
    }
};

template <int FMAS, int NT = 512, int OP = 0>
static void run(Sampler &smp, bool have, const char *what, int fill, double seconds)
{
    const int grid = 256, lds_bytes = 158 * 1024, trips = 2000;
    float *out;
    CK(hipMalloc(&out, 4));
    CK(hipFuncSetAttribute((const void *)reads_kernel<NT, FMAS, OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    reads_kernel<NT, FMAS, OP><<<grid, NT, lds_bytes>>>(out, trips, NT == 512 ? 19 * 1024 : 38 * 1024, fill, 1u);
    CK(hipDeviceSynchronize());
    if (have) smp.start();
    const auto t0 = std::chrono::steady_clock::now();
    double ms_total = 0.0, ms_last = 0.0;
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 20; ++r) reads_kernel<NT, FMAS, OP><<<grid, NT, lds_bytes>>>(out, trips, NT == 512 ? 19 * 1024 : 38 * 1024, fill, (unsigned)(launches + r));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ms_total += ms; ms_last = ms / 20; launches += 20;
    }
    if (have) smp.finish();
    const double reads = (double)grid * (NT / 64) * trips * 16;      // wave-level 1 KB reads per launch (the fill and the first four aside)
    const double b_per_ns = reads * 1024 / (ms_last * 1e6) / 256;
    const double mhz = have ? Sampler::median_tail(smp.mhz) : 0.0, w = have ? Sampler::median_tail(smp.watts) : 0.0;
    printf("%-64s %7.4f ms  %6.1f B/ns/CU  %5.0f MHz  %6.1f B/clk/CU  %6.0f W   (%ld launches, mean %.4f ms)\n", what, ms_last, b_per_ns, mhz,
           mhz > 0 ? b_per_ns * 1000.0 / mhz : 0.0, w, launches, ms_total / launches);
    fflush(stdout);
    CK(hipFree(out));
}

template <int KIND>
static void run_fma(Sampler &smp, bool have, const char *what, double seconds)
{
    const int grid = 256, trips = 4000;
    float *out;
    CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fma_kernel<KIND><<<grid, 512>>>(out, trips, 1u);
    CK(hipDeviceSynchronize());
    if (have) smp.start();
    const auto t0 = std::chrono::steady_clock::now();
    double ms_last = 0.0;
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 20; ++r) fma_kernel<KIND><<<grid, 512>>>(out, trips, (unsigned)(launches + r));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ms_last = ms / 20; launches += 20;
    }
    if (have) smp.finish();
    const double fmas = (double)grid * 512 * trips * 32;               // per launch
    const double per_ns = fmas / (ms_last * 1e6) / 256;
    const double mhz = have ? Sampler::median_tail(smp.mhz) : 0.0, w = have ? Sampler::median_tail(smp.watts) : 0.0;
    printf("%-64s %7.4f ms  %6.1f FMA/ns/CU  %5.0f MHz  %6.1f FMA/clk/CU (peak 128)  %6.0f W\n", what, ms_last, per_ns, mhz,
           mhz > 0 ? per_ns * 1000.0 / mhz : 0.0, w);
    fflush(stdout);
    CK(hipFree(out));
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 1.5;
    Sampler smp;
    const bool have = smp.find(0);
    printf("8 waves per CU, 158 KB of LDS, a channel's planes 19 KB apart, conflict-free ds_read_b128, 4 in flight per wave; %.1f s per case; hwmon %s\n",
           seconds, have ? smp.power.c_str() : "not found (no power / clock columns)");
    run<0>(smp, have, "reads alone, LDS holds zeros", 0, seconds);
    run<0>(smp, have, "reads alone, one word repeated", 1, seconds);
    run<0>(smp, have, "reads alone, random mantissas in [1, 2)", 2, seconds);
    run<0>(smp, have, "reads alone, uniform random in [-1, 1) (audio-like)", 3, seconds);
    run<3>(smp, have, "reads + 3 v_pk_fma_f32 per read (the octet kernel's ratio), zeros", 0, seconds);
    run<3>(smp, have, "reads + 3 v_pk_fma_f32 per read (the octet kernel's ratio), random", 3, seconds);
    run<6>(smp, have, "reads + 6 v_pk_fma_f32 per read, zeros", 0, seconds);
    run<6>(smp, have, "reads + 6 v_pk_fma_f32 per read, uniform random", 3, seconds);
    printf("-- fewer LDS bytes per FMA: 64-frame runs read 2.0 B per FMA (4 packed FMAs per read) where 32-frame runs read 2.6 (3 per read)\n");
    run<4>(smp, have, "8 waves: reads + 4 v_pk_fma_f32 per read, random", 3, seconds);
    run<4, 256>(smp, have, "4 waves (a half-wave per channel, 64-frame runs): + 4 per read, random", 3, seconds);
    run<3, 256>(smp, have, "4 waves: reads + 3 v_pk_fma_f32 per read, random", 3, seconds);
    printf("-- cfg3's tap phase: 4096 packed FMAs per 374 window reads, lane and tile = 11 per read (FMA rate = B/ns/CU / 1024 x 11 x 128 x 2 flop x 256 CUs)\n");
    run<11>(smp, have, "8 waves: reads + 11 v_pk_fma_f32 per read, zeros", 0, seconds);
    run<11>(smp, have, "8 waves: reads + 11 v_pk_fma_f32 per read, random", 3, seconds);
    printf("-- windows from the neighbouring lanes' registers: FMA streams on held registers, no LDS, 8 waves per CU, random operands\n");
    run_fma<0>(smp, have, "v_pk_fma_f32 (what the tap phase issues today)", seconds);
    run_fma<3>(smp, have, "v_fmac_f32 (one FMA per lane and instruction)", seconds);
    run_fma<1>(smp, have, "v_fmac_f32 with a DPP row_shr:1 operand (one FMA per lane and instruction)", seconds);
    run_fma<2>(smp, have, "v_mov_b32 row_shr:1 x 2 + v_pk_fma_f32 (a shifted pair)", seconds);
    printf("-- round 6: adds instead of FMAs (decorrelation.py:402-414: add / subtract inside a segment, one multiply per segment). Same reads, random data\n");
    run<3>(smp, have, "8 waves: reads + 3 v_pk_fma_f32 per read, random (again, for this box's clock)", 3, seconds);
    run<3, 512, 1>(smp, have, "8 waves: reads + 3 v_pk_add_f32 per read, random", 3, seconds);
    run<4>(smp, have, "8 waves: reads + 4 v_pk_fma_f32 per read, random (again)", 3, seconds);
    run<4, 512, 1>(smp, have, "8 waves: reads + 4 v_pk_add_f32 per read, random", 3, seconds);
    run<4, 512, 2>(smp, have, "8 waves: reads + 3 v_pk_add_f32 + 1 v_pk_fma_f32 per read, random", 3, seconds);
    run<6, 512, 1>(smp, have, "8 waves: reads + 6 v_pk_add_f32 per read, random", 3, seconds);
    run<11>(smp, have, "8 waves: reads + 11 v_pk_fma_f32 per read, random (again)", 3, seconds);
    run<11, 512, 1>(smp, have, "8 waves: reads + 11 v_pk_add_f32 per read, random", 3, seconds);
    run<11, 512, 2>(smp, have, "8 waves: reads + 11 per read, one in four an FMA, random", 3, seconds);
    printf("-- the instruction alone, held registers, random operands\n");
    run_fma<0>(smp, have, "v_pk_fma_f32 acc += x * w (again)", seconds);
    run_fma<4>(smp, have, "v_pk_add_f32 acc += x", seconds);
    run_fma<5>(smp, have, "v_pk_fma_f32 acc +- x * w, sign alternating per pass", seconds);
    run_fma<6>(smp, have, "v_pk_add_f32 acc +- x, sign alternating per pass", seconds);
    run_fma<7>(smp, have, "v_pk_mul_f32 acc = x * w", seconds);
    return 0;
}
