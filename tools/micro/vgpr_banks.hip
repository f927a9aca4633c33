// Does a v_pk_fma_f32 whose two VGPR-pair operands (x and the accumulator) lie in the SAME register banks (register number mod 4)
// issue slower than one whose pairs lie in different banks?  lds_power.hip's FMA stream - 16 accumulator pairs, 16 DIFFERENT x pairs,
// registers as hipcc allocated them: 12 of 16 instructions with both pairs in the same banks - ran at 63 FMAs per clock and CU where
// pk_fma_rate.hip (ONE x pair for every instruction) runs at 118-120 of the 128 peak.  Hand-allocated registers, 8 waves per CU:
//   acc pairs v[0:1] ... v[30:31]; x pairs v[32:33] ... v[62:63]; weight an SGPR pair (op_sel_hi broadcast), as the generated kernels'.
//   SAME:  instruction k uses x pair 32 + 2k      (banks equal to the accumulator's)
//   OTHER: instruction k uses x pair 32 + 2(k^1)  (the other two banks)
//   ONE:   every instruction uses x pair v[32:33]
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/vgpr_banks tools/micro/vgpr_banks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>
#include <unistd.h>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"
#define F(a, x) "v_pk_fma_f32 v[" #a ":" #a "+1], v[" #x ":" #x "+1], %0, v[" #a ":" #a "+1] op_sel_hi:[1,0,1]\n\t"
#define SAME16  F(0,32) F(2,34) F(4,36) F(6,38) F(8,40) F(10,42) F(12,44) F(14,46) F(16,48) F(18,50) F(20,52) F(22,54) F(24,56) F(26,58) F(28,60) F(30,62)
#define OTHER16 F(0,34) F(2,32) F(4,38) F(6,36) F(8,42) F(10,40) F(12,46) F(14,44) F(16,50) F(18,48) F(20,54) F(22,52) F(24,58) F(26,56) F(28,62) F(30,60)
#define ONE16   F(0,32) F(2,32) F(4,32) F(6,32) F(8,32) F(10,32) F(12,32) F(14,32) F(16,32) F(18,32) F(20,32) F(22,32) F(24,32) F(26,32) F(28,32) F(30,32)
// single FMAs: accumulator register a, x register x: same bank when (a - x) % 4 == 0
#define S(a, x) "v_fmac_f32 v" #a ", v" #x ", %1\n\t"
#define SSAME16  S(0,32) S(1,33) S(2,34) S(3,35) S(4,36) S(5,37) S(6,38) S(7,39) S(8,40) S(9,41) S(10,42) S(11,43) S(12,44) S(13,45) S(14,46) S(15,47)
#define SOTHER16 S(0,33) S(1,34) S(2,35) S(3,36) S(4,37) S(5,38) S(6,39) S(7,40) S(8,41) S(9,42) S(10,43) S(11,44) S(12,45) S(13,46) S(14,47) S(15,48)
#define X4(b) b b b b

template <int KIND>
__global__ __launch_bounds__(512) void k(float *out, int trips, float w0)
{
    const unsigned wbits = (unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(w0));
    const unsigned long long wpair = ((unsigned long long)wbits << 32) | wbits;
    const float ws = __builtin_amdgcn_readfirstlane(w0);
    const unsigned iseed = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) ^ 0x9e3779b9u;      // x registers: an LCG per lane, as floats uniform in [-1, 1)
    asm volatile("v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
                 "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t"
                 "v_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t"
                 "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\tv_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t"
                 "v_mov_b32 v32, %0\n\t"
                 "v_mul_lo_u32 v33, v32, %1\n\tv_add_u32 v33, 0x3c6ef35f, v33\n\t"
                 "v_mul_lo_u32 v34, v33, %1\n\tv_add_u32 v34, 0x3c6ef35f, v34\n\t"
                 "v_mul_lo_u32 v35, v34, %1\n\tv_add_u32 v35, 0x3c6ef35f, v35\n\t"
                 "v_mul_lo_u32 v36, v35, %1\n\tv_add_u32 v36, 0x3c6ef35f, v36\n\t"
                 "v_mul_lo_u32 v37, v36, %1\n\tv_add_u32 v37, 0x3c6ef35f, v37\n\t"
                 "v_mul_lo_u32 v38, v37, %1\n\tv_add_u32 v38, 0x3c6ef35f, v38\n\t"
                 "v_mul_lo_u32 v39, v38, %1\n\tv_add_u32 v39, 0x3c6ef35f, v39\n\t"
                 "v_mul_lo_u32 v40, v39, %1\n\tv_add_u32 v40, 0x3c6ef35f, v40\n\t"
                 "v_mul_lo_u32 v41, v40, %1\n\tv_add_u32 v41, 0x3c6ef35f, v41\n\t"
                 "v_mul_lo_u32 v42, v41, %1\n\tv_add_u32 v42, 0x3c6ef35f, v42\n\t"
                 "v_mul_lo_u32 v43, v42, %1\n\tv_add_u32 v43, 0x3c6ef35f, v43\n\t"
                 "v_mul_lo_u32 v44, v43, %1\n\tv_add_u32 v44, 0x3c6ef35f, v44\n\t"
                 "v_mul_lo_u32 v45, v44, %1\n\tv_add_u32 v45, 0x3c6ef35f, v45\n\t"
                 "v_mul_lo_u32 v46, v45, %1\n\tv_add_u32 v46, 0x3c6ef35f, v46\n\t"
                 "v_mul_lo_u32 v47, v46, %1\n\tv_add_u32 v47, 0x3c6ef35f, v47\n\t"
                 "v_mul_lo_u32 v48, v47, %1\n\tv_add_u32 v48, 0x3c6ef35f, v48\n\t"
                 "v_mul_lo_u32 v49, v48, %1\n\tv_add_u32 v49, 0x3c6ef35f, v49\n\t"
                 "v_mul_lo_u32 v50, v49, %1\n\tv_add_u32 v50, 0x3c6ef35f, v50\n\t"
                 "v_mul_lo_u32 v51, v50, %1\n\tv_add_u32 v51, 0x3c6ef35f, v51\n\t"
                 "v_mul_lo_u32 v52, v51, %1\n\tv_add_u32 v52, 0x3c6ef35f, v52\n\t"
                 "v_mul_lo_u32 v53, v52, %1\n\tv_add_u32 v53, 0x3c6ef35f, v53\n\t"
                 "v_mul_lo_u32 v54, v53, %1\n\tv_add_u32 v54, 0x3c6ef35f, v54\n\t"
                 "v_mul_lo_u32 v55, v54, %1\n\tv_add_u32 v55, 0x3c6ef35f, v55\n\t"
                 "v_mul_lo_u32 v56, v55, %1\n\tv_add_u32 v56, 0x3c6ef35f, v56\n\t"
                 "v_mul_lo_u32 v57, v56, %1\n\tv_add_u32 v57, 0x3c6ef35f, v57\n\t"
                 "v_mul_lo_u32 v58, v57, %1\n\tv_add_u32 v58, 0x3c6ef35f, v58\n\t"
                 "v_mul_lo_u32 v59, v58, %1\n\tv_add_u32 v59, 0x3c6ef35f, v59\n\t"
                 "v_mul_lo_u32 v60, v59, %1\n\tv_add_u32 v60, 0x3c6ef35f, v60\n\t"
                 "v_mul_lo_u32 v61, v60, %1\n\tv_add_u32 v61, 0x3c6ef35f, v61\n\t"
                 "v_mul_lo_u32 v62, v61, %1\n\tv_add_u32 v62, 0x3c6ef35f, v62\n\t"
                 "v_mul_lo_u32 v63, v62, %1\n\tv_add_u32 v63, 0x3c6ef35f, v63\n\t"
                 "v_cvt_f32_i32 v32, v32\n\tv_mul_f32 v32, 0x30000000, v32\n\t"
                 "v_cvt_f32_i32 v33, v33\n\tv_mul_f32 v33, 0x30000000, v33\n\t"
                 "v_cvt_f32_i32 v34, v34\n\tv_mul_f32 v34, 0x30000000, v34\n\t"
                 "v_cvt_f32_i32 v35, v35\n\tv_mul_f32 v35, 0x30000000, v35\n\t"
                 "v_cvt_f32_i32 v36, v36\n\tv_mul_f32 v36, 0x30000000, v36\n\t"
                 "v_cvt_f32_i32 v37, v37\n\tv_mul_f32 v37, 0x30000000, v37\n\t"
                 "v_cvt_f32_i32 v38, v38\n\tv_mul_f32 v38, 0x30000000, v38\n\t"
                 "v_cvt_f32_i32 v39, v39\n\tv_mul_f32 v39, 0x30000000, v39\n\t"
                 "v_cvt_f32_i32 v40, v40\n\tv_mul_f32 v40, 0x30000000, v40\n\t"
                 "v_cvt_f32_i32 v41, v41\n\tv_mul_f32 v41, 0x30000000, v41\n\t"
                 "v_cvt_f32_i32 v42, v42\n\tv_mul_f32 v42, 0x30000000, v42\n\t"
                 "v_cvt_f32_i32 v43, v43\n\tv_mul_f32 v43, 0x30000000, v43\n\t"
                 "v_cvt_f32_i32 v44, v44\n\tv_mul_f32 v44, 0x30000000, v44\n\t"
                 "v_cvt_f32_i32 v45, v45\n\tv_mul_f32 v45, 0x30000000, v45\n\t"
                 "v_cvt_f32_i32 v46, v46\n\tv_mul_f32 v46, 0x30000000, v46\n\t"
                 "v_cvt_f32_i32 v47, v47\n\tv_mul_f32 v47, 0x30000000, v47\n\t"
                 "v_cvt_f32_i32 v48, v48\n\tv_mul_f32 v48, 0x30000000, v48\n\t"
                 "v_cvt_f32_i32 v49, v49\n\tv_mul_f32 v49, 0x30000000, v49\n\t"
                 "v_cvt_f32_i32 v50, v50\n\tv_mul_f32 v50, 0x30000000, v50\n\t"
                 "v_cvt_f32_i32 v51, v51\n\tv_mul_f32 v51, 0x30000000, v51\n\t"
                 "v_cvt_f32_i32 v52, v52\n\tv_mul_f32 v52, 0x30000000, v52\n\t"
                 "v_cvt_f32_i32 v53, v53\n\tv_mul_f32 v53, 0x30000000, v53\n\t"
                 "v_cvt_f32_i32 v54, v54\n\tv_mul_f32 v54, 0x30000000, v54\n\t"
                 "v_cvt_f32_i32 v55, v55\n\tv_mul_f32 v55, 0x30000000, v55\n\t"
                 "v_cvt_f32_i32 v56, v56\n\tv_mul_f32 v56, 0x30000000, v56\n\t"
                 "v_cvt_f32_i32 v57, v57\n\tv_mul_f32 v57, 0x30000000, v57\n\t"
                 "v_cvt_f32_i32 v58, v58\n\tv_mul_f32 v58, 0x30000000, v58\n\t"
                 "v_cvt_f32_i32 v59, v59\n\tv_mul_f32 v59, 0x30000000, v59\n\t"
                 "v_cvt_f32_i32 v60, v60\n\tv_mul_f32 v60, 0x30000000, v60\n\t"
                 "v_cvt_f32_i32 v61, v61\n\tv_mul_f32 v61, 0x30000000, v61\n\t"
                 "v_cvt_f32_i32 v62, v62\n\tv_mul_f32 v62, 0x30000000, v62\n\t"
                 "v_cvt_f32_i32 v63, v63\n\tv_mul_f32 v63, 0x30000000, v63\n\t"
                 :: "v"(iseed), "s"(1664525u) : CLOB);
    for (int t = 0; t < trips; ++t) {
        if constexpr (KIND == 0) asm volatile(X4(X4(SAME16)) :: "s"(wpair), "s"(ws) : CLOB);
        else if constexpr (KIND == 1) asm volatile(X4(X4(OTHER16)) :: "s"(wpair), "s"(ws) : CLOB);
        else if constexpr (KIND == 2) asm volatile(X4(X4(ONE16)) :: "s"(wpair), "s"(ws) : CLOB);
        else if constexpr (KIND == 3) asm volatile(X4(X4(SSAME16)) :: "s"(wpair), "s"(ws) : CLOB);
        else asm volatile(X4(X4(SOTHER16)) :: "s"(wpair), "s"(ws) : CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v0, v31\n\tv_add_f32 %0, %0, v16" : "=v"(r) :: CLOB);
    if (r == 123.456f) out[0] = r;
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

static Sampler g_smp;
static bool g_have = false;
static double g_seconds = 0.0;

template <int KIND>
static void run(const char *what, int fmas_per_instr)
{
    float *out;
    CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int trips = 2000, grid = 256;
    k<KIND><<<grid, 512>>>(out, trips, 0.37f);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        k<KIND><<<grid, 512>>>(out, trips, 0.37f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double instrs = (double)grid * 8 * trips * 256;               // wave-instructions per launch
    const double fmas = instrs * 64 * fmas_per_instr;
    printf("%-78s %7.4f ms  %6.1f TFLOP/s  %6.1f FMA per ns and CU  (%.2f ns per wave-instruction and SIMD)\n", what, best, 2 * fmas / best / 1e9,
           fmas / best / 1e6 / 256, best * 1e6 / (instrs / (256 * 4)));
    if (g_seconds > 0.0) {
        // sustained: back-to-back launches for g_seconds, the card's power and clock sampled meanwhile (the cap acts within ~0.1 s)
        if (g_have) g_smp.start();
        const auto t0 = std::chrono::steady_clock::now();
        float last = 0.0f;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < g_seconds) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < 20; ++r) k<KIND><<<grid, 512>>>(out, trips, 0.37f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&last, e0, e1));
            last /= 20;
        }
        if (g_have) g_smp.finish();
        const double mhz = g_have ? Sampler::median_tail(g_smp.mhz) : 0.0, w = g_have ? Sampler::median_tail(g_smp.watts) : 0.0;
        printf("%-78s %7.4f ms  %6.1f TFLOP/s  %6.1f FMA per ns and CU  sustained %.1f s: %5.0f MHz %6.0f W  %5.1f FMA per clock and CU\n", "", last,
               2 * fmas / last / 1e9, fmas / last / 1e6 / 256, g_seconds, mhz, w, mhz > 0 ? fmas / last / 1e6 / 256 * 1000.0 / mhz : 0.0);
    }
    CK(hipFree(out));
}

int main(int argc, char **argv)
{
    g_seconds = argc > 1 ? atof(argv[1]) : 0.0;
    g_have = g_smp.find(0);
    printf("8 waves per CU (two per SIMD), 256 FMA instructions per loop trip, hand-allocated registers; 1 ms launches (uncapped clock, ~2.4 GHz)\n");
    run<2>("v_pk_fma_f32, ONE x pair for every instruction", 2);
    run<1>("v_pk_fma_f32, x pair in the OTHER two banks than the accumulator pair", 2);
    run<0>("v_pk_fma_f32, x pair in the SAME banks as the accumulator pair", 2);
    run<4>("v_fmac_f32, x register in another bank than the accumulator", 1);
    run<3>("v_fmac_f32, x register in the SAME bank as the accumulator", 1);
    return 0;
}
