// What a dependent kernel boundary costs on this box, by what the kernels are: back-to-back launches on one stream of
//   (a) an empty kernel, by grid and dynamic LDS;  (b) a streaming copy of a cfg4 N = 8 shard (49 MB each way, rotating
//   buffers) whose first / last workgroup stamp the 100 MHz wall clock - per-launch time minus the in-kernel span is the
//   boundary;  (c) the same with the launch's register / LDS footprint of the window kernel (256 VGPRs, 80 KB).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/boundary tools/micro/boundary.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void empty_kernel(int *p) { extern __shared__ int lds[]; if (p && threadIdx.x == 9999) p[0] = lds[0]; }

// copy with stamps: every workgroup records start and end (after its stores are acknowledged)
template <int AUX_NT>
__global__ __launch_bounds__(256) void copy_kernel(const v4f *x, v4f *y, long long quads, unsigned long long *stamps)
{
    extern __shared__ int lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long base = (long long)blockIdx.x * 256 * 4 + threadIdx.x; base < quads; base += stride) {
        v4f a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (base + k * 256 < quads) a[k] = AUX_NT ? __builtin_nontemporal_load(x + base + k * 256) : x[base + k * 256];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (base + k * 256 < quads) { if (AUX_NT) __builtin_nontemporal_store(a[k], y + base + k * 256); else y[base + k * 256] = a[k]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

static float time_launches(int iters, const std::function<void(int)> &launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 50; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;
}

int main()
{
    CK(hipFuncSetAttribute((const void *)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)copy_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)copy_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("(a) empty kernels, us per launch back to back on one stream\n");
    for (int grid : {1, 256, 384, 512, 2048})
        for (int lds : {0, 80 * 1024})
            for (int nt : {64, 256})
                printf("    grid %5d x %3d threads, %6d B LDS: %6.2f us\n", grid, nt, lds,
                       time_launches(2000, [&](int) { empty_kernel<<<grid, nt, lds>>>(nullptr); }));
    const long long shard = 128LL * 48000 * 2 * 4;                       // bytes each way
    const int buffers = 7;
    std::vector<v4f *> xs(buffers), ys(buffers);
    for (int b = 0; b < buffers; ++b) { CK(hipMalloc(&xs[b], shard)); CK(hipMalloc(&ys[b], shard)); CK(hipMemset(xs[b], 1, shard)); CK(hipMemset(ys[b], 0, shard)); }
    unsigned long long *stamps;
    CK(hipMalloc(&stamps, 2 * 65536 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(2 * 65536);
    printf("(b) copy of the N = 8 shard (49 MB each way, %d rotating buffers)\n", buffers);
    for (int nt = 0; nt < 2; ++nt)
        for (int grid : {256, 512, 1024, 2048, 6144})
            for (int lds : {0, 80 * 1024}) {
                if (lds && grid > 512) continue;
                auto launch = [&](int i) {
                    if (nt) copy_kernel<1><<<grid, 256, lds>>>(xs[i % buffers], ys[i % buffers], shard / 16, stamps);
                    else copy_kernel<0><<<grid, 256, lds>>>(xs[i % buffers], ys[i % buffers], shard / 16, stamps);
                };
                const float us = time_launches(700, launch);
                CK(hipMemcpy(h.data(), stamps, 2 * grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                unsigned long long lo = ~0ull, hi = 0, last_start = 0;
                for (int b = 0; b < grid; ++b) { lo = std::min(lo, h[2 * b]); hi = std::max(hi, h[2 * b + 1]); last_start = std::max(last_start, h[2 * b]); }
                printf("    %s grid %5d, %6d B LDS: %6.2f us per launch; in-kernel span %6.2f us (last workgroup started %5.2f us after the first); boundary %5.2f us\n",
                       nt ? "nt   " : "plain", grid, lds, us, (hi - lo) * 0.01, (last_start - lo) * 0.01, us - (hi - lo) * 0.01);
            }
    return 0;
}
