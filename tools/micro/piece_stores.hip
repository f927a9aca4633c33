// What an interleaved 8-channel signal (32-byte frames) costs to READ and WRITE when a workgroup owns one channel
// PAIR (8 bytes of every frame) - the per-table kernels' access shape for cfg5 - under different lane mappings, against
// whole-frame accesses.  A copy kernel: no arithmetic, no LDS; spans of 8192 frames per workgroup, 8 accesses in flight.
//   pairs/2f : lane = a frame PAIR of its channel pair: two 8-byte accesses 32 bytes apart, lanes 64 bytes apart (today)
//   pairs/1f : lane = ONE frame: 8-byte accesses, lanes 32 bytes apart (two lanes share a 64-byte block, four a line)
//   pairs/q4 : as pairs/1f, but the lanes of a QUAD take four consecutive frames and the next quad starts eight frames
//              on (what a lane-quad exchange of pairs/2f registers gives: every other line per instruction)
//   quads/1f : a workgroup owns TWO channel pairs: 16-byte accesses, lanes 32 bytes apart
//   frames   : a workgroup owns whole frames: 16-byte accesses, consecutive (the device-copy shape)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/piece_stores tools/micro/piece_stores.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int C = 8, SPAN = 8192, NT = 256;

// MODE 0: pairs/2f   1: pairs/1f   2: quads/1f   3: frames   4: pairs/q4
template <int MODE, bool LOADS, bool STORES, bool REMAP = true>
__global__ __launch_bounds__(NT) void copy_kernel(const float *x, float *y, long long frames, int spans)
{
    constexpr int GROUPS = MODE == 3 ? 1 : (MODE == 2 ? 2 : 4);
    constexpr bool Q4 = MODE == 4;
    // blocks b and b + 8 share an XCD (and its L2): consecutive logical ids per XCD, as the product kernels do, so the
    // workgroups that own the pieces of one line share an L2 (REMAP = false: pieces of a line in different L2s)
    const unsigned q = gridDim.x >> 3, xcd = blockIdx.x & 7u;
    const unsigned first = REMAP ? xcd * q + (blockIdx.x >> 3) : blockIdx.x;
    for (long long unit = first; unit < (long long)spans * GROUPS; unit += gridDim.x) {
        const int g = (int)(unit % GROUPS);
        const long long f0 = (unit / GROUPS) * SPAN;
        const int tid = threadIdx.x;
        if constexpr (MODE == 0) {
            for (int it = 0; it < SPAN / (2 * NT * 4); ++it) {
                v2f a[4][2];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const long long f = f0 + 2 * ((it * 4 + k) * NT + tid);
                    if (LOADS) { a[k][0] = *(const v2f *)(x + f * C + 2 * g); a[k][1] = *(const v2f *)(x + (f + 1) * C + 2 * g); }
                    else { a[k][0] = v2f{(float)f, 1.f}; a[k][1] = v2f{2.f, (float)tid}; }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const long long f = f0 + 2 * ((it * 4 + k) * NT + tid);
                    if (STORES) { *(v2f *)(y + f * C + 2 * g) = a[k][0]; *(v2f *)(y + (f + 1) * C + 2 * g) = a[k][1]; }
                    else if (a[k][0].x == 123.456f && a[k][1].y == 3.f) y[f] = 1.f;
                }
            }
        } else if constexpr (MODE == 1 || MODE == 4) {
            for (int it = 0; it < SPAN / (NT * 8); ++it) {
                v2f a[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const long long f = Q4 ? f0 + (it * 8 + (k & ~1)) * NT + 8 * (tid >> 2) + 4 * (k & 1) + (tid & 3) : f0 + (it * 8 + k) * NT + tid;
                    a[k] = LOADS ? *(const v2f *)(x + f * C + 2 * g) : v2f{(float)f, (float)tid};
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const long long f = Q4 ? f0 + (it * 8 + (k & ~1)) * NT + 8 * (tid >> 2) + 4 * (k & 1) + (tid & 3) : f0 + (it * 8 + k) * NT + tid;
                    if (STORES) *(v2f *)(y + f * C + 2 * g) = a[k];
                    else if (a[k].x == 123.456f) y[f] = 1.f;
                }
            }
        } else if constexpr (MODE == 2) {
            for (int it = 0; it < SPAN / (NT * 8); ++it) {
                v4f a[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const long long f = f0 + (it * 8 + k) * NT + tid;
                    a[k] = LOADS ? *(const v4f *)(x + f * C + 4 * g) : v4f{(float)f, (float)tid, 0.f, 1.f};
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const long long f = f0 + (it * 8 + k) * NT + tid;
                    if (STORES) *(v4f *)(y + f * C + 4 * g) = a[k];
                    else if (a[k].x == 123.456f) y[f] = 1.f;
                }
            }
        } else {
            for (int it = 0; it < SPAN * 2 / (NT * 8); ++it) {          // 2 x 16 bytes per frame
                v4f a[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const long long q = f0 * 2 + (it * 8 + k) * NT + tid;
                    a[k] = LOADS ? *(const v4f *)(x + q * 4) : v4f{(float)q, (float)tid, 0.f, 1.f};
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const long long q = f0 * 2 + (it * 8 + k) * NT + tid;
                    if (STORES) *(v4f *)(y + q * 4) = a[k];
                    else if (a[k].x == 123.456f) y[q] = 1.f;
                }
            }
        }
    }
}

template <int MODE, bool L, bool S, bool REMAP = true>
static void run(const char *name, const float *x, float *y, long long frames)
{
    const int spans = (int)(frames / SPAN);
    const int grid = 256 * 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) copy_kernel<MODE, L, S, REMAP><<<grid, NT>>>(x, y, frames, spans);
    CK(hipEventRecord(e0));
    const int iters = 30;
    for (int i = 0; i < iters; ++i) copy_kernel<MODE, L, S, REMAP><<<grid, NT>>>(x, y, frames, spans);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    const double bytes = (double)frames * C * 4 * ((L ? 1 : 0) + (S ? 1 : 0));
    printf("%-24s %-12s %.4f ms  %6.0f GB/s\n", name, L && S ? "read+write" : (L ? "read only" : "write only"), ms, bytes / ms / 1e6);
}

int main()
{
    const long long frames = 16LL * 960000 / SPAN * SPAN;           // cfg5's pool: 16 streams of 10 s at 96 kHz
    float *x, *y;
    CK(hipMalloc(&x, frames * C * 4)); CK(hipMalloc(&y, frames * C * 4));
    CK(hipMemset(x, 0, frames * C * 4)); CK(hipMemset(y, 0, frames * C * 4));
    printf("%lld frames of %d channels (%.2f GB each way)\n", frames, C, frames * C * 4 / 1e9);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, true, true>("pairs/2f", x, y, frames); run<1, true, true>("pairs/1f", x, y, frames); run<4, true, true>("pairs/q4", x, y, frames);
        run<2, true, true>("quads/1f", x, y, frames); run<3, true, true>("frames", x, y, frames);
        run<0, true, false>("pairs/2f", x, y, frames); run<1, true, false>("pairs/1f", x, y, frames); run<4, true, false>("pairs/q4", x, y, frames);
        run<2, true, false>("quads/1f", x, y, frames); run<3, true, false>("frames", x, y, frames);
        run<0, false, true>("pairs/2f", x, y, frames); run<1, false, true>("pairs/1f", x, y, frames); run<4, false, true>("pairs/q4", x, y, frames);
        run<2, false, true>("quads/1f", x, y, frames); run<3, false, true>("frames", x, y, frames);
    }
    run<0, true, true, false>("pairs/2f, no XCD remap", x, y, frames);
    run<2, true, true, false>("quads/1f, no XCD remap", x, y, frames);
    return 0;
}
