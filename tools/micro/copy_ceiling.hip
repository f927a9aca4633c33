// The HBM ceiling a streaming kernel can reach on this box at the headline's size: a plain copy of 7.86 GB (read) into
// 7.86 GB (write) - cfg2's pool of 2048 x 10 s stereo float32 - with 16-byte accesses, several grid shapes and cache
// policies; plus read-only and write-only passes.  The convolution kernel moves exactly these bytes (1.0002x, PMC).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/copy_ceiling tools/micro/copy_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// MODE 0 copy, 1 read only, 2 write only; NT: non-temporal loads and stores
template <int MODE, bool NT, int UNROLL>
__global__ __launch_bounds__(256) void stream_kernel(const v4f *x, v4f *y, long long quads)
{
    const long long stride = (long long)gridDim.x * 256 * UNROLL;
    v4f keep = {0.f, 0.f, 0.f, 0.f};
    for (long long base = (long long)blockIdx.x * 256 * UNROLL + threadIdx.x; base < quads; base += stride) {
        v4f a[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const long long q = base + k * 256;
            if (MODE != 2) a[k] = q < quads ? (NT ? __builtin_nontemporal_load(x + q) : x[q]) : keep;
            else a[k] = v4f{(float)q, 1.f, 2.f, 3.f};
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const long long q = base + k * 256;
            if (MODE != 1) { if (q < quads) { if (NT) __builtin_nontemporal_store(a[k], y + q); else y[q] = a[k]; } }
            else keep += a[k];
        }
    }
    if (MODE == 1 && keep.x == 123.456f) y[0] = keep;
}

template <int MODE, bool NT, int UNROLL>
static void run(const char *name, const v4f *x, v4f *y, long long quads, int grid)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) stream_kernel<MODE, NT, UNROLL><<<grid, 256>>>(x, y, quads);
    CK(hipEventRecord(e0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) stream_kernel<MODE, NT, UNROLL><<<grid, 256>>>(x, y, quads);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    const double bytes = (double)quads * 16 * (MODE == 0 ? 2 : 1);
    printf("%-44s grid %6d  %.3f ms  %5.2f TB/s\n", name, grid, ms, bytes / ms / 1e9);
}

int main()
{
    const long long quads = 2048LL * 480000 * 2 * 4 / 16;            // cfg2's pool: 7.86 GB
    v4f *x, *y;
    CK(hipMalloc(&x, quads * 16)); CK(hipMalloc(&y, quads * 16));
    CK(hipMemset(x, 0, quads * 16)); CK(hipMemset(y, 0, quads * 16));
    printf("%.2f GB each way\n", quads * 16 / 1e9);
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) CK(hipMemcpyAsync(y, x, quads * 16, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("%-44s              %.3f ms  %5.2f TB/s\n", "hipMemcpyAsync device to device", ms, 2.0 * quads * 16 / ms / 1e9);
        for (int grid : {2048, 8192, 65536}) {
            run<0, false, 4>("copy, unroll 4", x, y, quads, grid);
            run<0, true, 4>("copy, non-temporal, unroll 4", x, y, quads, grid);
            run<0, true, 8>("copy, non-temporal, unroll 8", x, y, quads, grid);
        }
        run<1, true, 8>("read only, non-temporal", x, y, quads, 8192);
        run<1, false, 8>("read only", x, y, quads, 8192);
        run<2, true, 8>("write only, non-temporal", x, y, quads, 8192);
        run<2, false, 8>("write only", x, y, quads, 8192);
    }
    return 0;
}
