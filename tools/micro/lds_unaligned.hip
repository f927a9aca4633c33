// Microbenchmark (diagnostic, not product): are 4-byte-aligned ds_read_b64 / ds_read_b128 legal on
// gfx950 under ROCm's LDS alignment mode, and at what rate do they run?
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/lds_unaligned.hip -o tools/micro/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// correctness: lane l reads b64 at byte 8*l + SHIFT and b128 at 16*l + SHIFT
template <int SHIFT>
__global__ void check_kernel(float *out64, float *out128)
{
    __shared__ __attribute__((aligned(16))) float s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)s;
    v2f a; v4f b;
    asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a) : "v"(base + threadIdx.x * 8), "n"(SHIFT));
    asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b) : "v"(base + threadIdx.x * 16), "n"(SHIFT));
    out64[threadIdx.x * 2] = a.x; out64[threadIdx.x * 2 + 1] = a.y;
    out128[threadIdx.x * 4] = b.x; out128[threadIdx.x * 4 + 1] = b.y; out128[threadIdx.x * 4 + 2] = b.z; out128[threadIdx.x * 4 + 3] = b.w;
}

// rate: every wave issues ITER x 8 reads of the given width at lane stride = width, base shifted by SHIFT bytes
template <int WIDTH, int SHIFT>
__global__ __launch_bounds__(256) void rate_kernel(float *sink, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float s[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = (float)(i & 255);
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)s + threadIdx.x * WIDTH + SHIFT;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (WIDTH == 8) {
            v2f r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b64 %0, %8 offset:0\n\tds_read_b64 %1, %8 offset:2048\n\tds_read_b64 %2, %8 offset:4096\n\t"
                         "ds_read_b64 %3, %8 offset:6144\n\tds_read_b64 %4, %8 offset:8192\n\tds_read_b64 %5, %8 offset:10240\n\t"
                         "ds_read_b64 %6, %8 offset:12288\n\tds_read_b64 %7, %8 offset:14336\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(base));
            acc += r0.x + r1.y + r2.x + r3.y + r4.x + r5.y + r6.x + r7.y;
        } else {
            v4f r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b128 %0, %8 offset:0\n\tds_read_b128 %1, %8 offset:4096\n\tds_read_b128 %2, %8 offset:8192\n\t"
                         "ds_read_b128 %3, %8 offset:12288\n\tds_read_b128 %4, %8 offset:16384\n\tds_read_b128 %5, %8 offset:20480\n\t"
                         "ds_read_b128 %6, %8 offset:24576\n\tds_read_b128 %7, %8 offset:28672\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(base));
            acc += r0.x + r1.y + r2.z + r3.w + r4.x + r5.y + r6.z + r7.w;
        }
    }
    if (acc == -1.0f) sink[0] = acc;
}

template <int WIDTH, int SHIFT>
static void run_rate(float *sink)
{
    const int iters = 4000, blocks = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = 36 * 1024;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((rate_kernel<WIDTH, SHIFT>), dim3(blocks), dim3(256), lds, 0, sink, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)blocks * 256 * iters * 8 * WIDTH;
        if (rep == 2) printf("ds_read_b%d shift %2d: %.3f ms, %.1f TB/s LDS\n", WIDTH * 8, SHIFT, ms, bytes / ms * 1e-9);
    }
}

template <int SHIFT>
static void run_check(float *d64, float *d128)
{
    hipLaunchKernelGGL((check_kernel<SHIFT>), dim3(1), dim3(64), 0, 0, d64, d128);
    CK(hipDeviceSynchronize());
    std::vector<float> h64(128), h128(256);
    CK(hipMemcpy(h64.data(), d64, 128 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h128.data(), d128, 256 * 4, hipMemcpyDeviceToHost));
    int bad64 = 0, bad128 = 0;
    for (int l = 0; l < 64; ++l) {
        for (int e = 0; e < 2; ++e) bad64 += h64[2 * l + e] != (float)(2 * l + e + SHIFT / 4);
        for (int e = 0; e < 4; ++e) bad128 += h128[4 * l + e] != (float)(4 * l + e + SHIFT / 4);
    }
    printf("shift %2d: b64 wrong %d/128 (lane1: %g %g), b128 wrong %d/256 (lane1: %g %g %g %g)\n", SHIFT, bad64, h64[2], h64[3],
           bad128, h128[4], h128[5], h128[6], h128[7]);
}

int main()
{
    float *d64, *d128, *sink;
    CK(hipMalloc(&d64, 4096)); CK(hipMalloc(&d128, 4096)); CK(hipMalloc(&sink, 64));
    run_check<0>(d64, d128); run_check<4>(d64, d128); run_check<8>(d64, d128); run_check<12>(d64, d128);
    run_rate<8, 0>(sink); run_rate<8, 4>(sink);
    run_rate<16, 0>(sink); run_rate<16, 4>(sink); run_rate<16, 8>(sink);
    return 0;
}
