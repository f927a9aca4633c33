// The streaming ceiling of a mono -> stereo fan-out on this box: a kernel that reads 4 bytes and writes 8 per frame (each input
// frame duplicated into both output channels), 16-byte accesses, non-temporal or plain, at the bench leg's size (128 x 10 s) and
// at 2048 x 10 s.  What next_rows.mono_to_stereo_fast (12 B per frame) can at most reach.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/fanout_ceiling tools/micro/fanout_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void fan_kernel(const v4f *x, v4f *y, long long quads)      // quads of INPUT frames
{
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long base = (long long)blockIdx.x * 256 * 4 + threadIdx.x; base < quads; base += stride) {
        v4f a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (base + k * 256 < quads) a[k] = NT ? __builtin_nontemporal_load(x + base + k * 256) : x[base + k * 256];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (base + k * 256 < quads) {
            const v4f lo = {a[k].x, a[k].x, a[k].y, a[k].y}, hi = {a[k].z, a[k].z, a[k].w, a[k].w};
            if (NT) { __builtin_nontemporal_store(lo, y + 2 * (base + k * 256)); __builtin_nontemporal_store(hi, y + 2 * (base + k * 256) + 1); }
            else { y[2 * (base + k * 256)] = lo; y[2 * (base + k * 256) + 1] = hi; }
        }
    }
}

int main()
{
    for (long long streams : {128LL, 2048LL}) {
        const long long frames = streams * 480000, quads = frames / 4;
        v4f *x, *y;
        CK(hipMalloc(&x, frames * 4)); CK(hipMalloc(&y, frames * 8));
        CK(hipMemset(x, 1, frames * 4)); CK(hipMemset(y, 0, frames * 8));
        for (int nt = 0; nt < 2; ++nt)
            for (int grid : {2048, 8192, 65536}) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int i = 0; i < 5; ++i) { if (nt) fan_kernel<true><<<grid, 256>>>(x, y, quads); else fan_kernel<false><<<grid, 256>>>(x, y, quads); }
                CK(hipEventRecord(e0));
                const int iters = streams > 1000 ? 20 : 200;
                for (int i = 0; i < iters; ++i) { if (nt) fan_kernel<true><<<grid, 256>>>(x, y, quads); else fan_kernel<false><<<grid, 256>>>(x, y, quads); }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
                printf("%5lld x 10 s mono -> stereo, %s, grid %6d: %.4f ms  %5.2f TB/s of 12 B per frame = %.3f of 8 TB/s\n", streams, nt ? "non-temporal" : "plain       ", grid, ms,
                       frames * 12.0 / ms / 1e9, frames * 12.0 / ms / 1e9 / 8.0);
            }
        CK(hipFree(x)); CK(hipFree(y));
    }
    return 0;
}
