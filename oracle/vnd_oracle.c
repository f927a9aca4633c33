/* CPU oracle (C restatement) of the velvet-noise sparse tap sum.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.  Never by the product library.
 *
 * Restates ckonst/VNDecorrelate v1.1.0:
 *   - function path  src/vndecorrelate/decorrelation.py:630-660
 *       y[n,c] = sum_k w[c,k] * x[n + i[c,k], c], taps in table order,
 *       float32 recurrence acc = f32(acc + f32(x*w))  (this
 *       sequential no-FMA form is bit-identical to the NumPy slice form)
 *   - class path     src/vndecorrelate/decorrelation.py:393-415
 *       per segment: seg = (+0 -x.. +x..) ; seg *= gain (unless identity) ;
 *       out += seg
 * Parity: pinned - tests/test_oracle_golden.py checks it bit-for-bit against
 * fixtures captured from the reference (oracle/gen_golden.py).
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off, no fast-math).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 2048

/* One stream, interleaved (n, C) float32.  seg_* may be NULL (function path:
 * one implicit segment per channel, no gain).  chan_flags[c]&1 => pass-through. */
static void conv_stream(const float *x, float *y, int64_t n, int32_t C,
                        const int32_t *tap_off, const int32_t *idx, const float *w,
                        const int32_t *seg_off, const int32_t *seg_end,
                        const float *seg_gain, const uint8_t *chan_flags,
                        int apply_gain, int64_t t0, int64_t t1)
{
    float acc[TILE], out[TILE];
    for (int64_t base = t0; base < t1; base += TILE) {
        int64_t len = t1 - base < TILE ? t1 - base : TILE;
        for (int32_t c = 0; c < C; ++c) {
            if (chan_flags && (chan_flags[c] & 1)) {
                for (int64_t j = 0; j < len; ++j) y[(base + j) * C + c] = x[(base + j) * C + c];
                continue;
            }
            int32_t nseg = seg_off ? seg_off[c + 1] - seg_off[c] : 1;
            int32_t k = tap_off[c];
            for (int64_t j = 0; j < len; ++j) out[j] = 0.0f;
            for (int32_t s = 0; s < nseg; ++s) {
                int32_t kend = seg_off ? seg_end[seg_off[c] + s] : tap_off[c + 1];
                for (int64_t j = 0; j < len; ++j) acc[j] = 0.0f;
                for (; k < kend; ++k) {
                    int64_t i = idx[k];
                    float wk = w[k];
                    int64_t lim = n - i - base;          /* j < lim  <=>  base+j+i < n */
                    if (lim > len) lim = len;
                    const float *xs = x + (base + i) * C + c;
                    for (int64_t j = 0; j < lim; ++j) {
                        float p = xs[j * C] * wk;         /* separate rounding: no FMA */
                        acc[j] = acc[j] + p;
                    }
                }
                if (seg_off) {
                    if (apply_gain) {
                        float g = seg_gain[seg_off[c] + s];
                        for (int64_t j = 0; j < len; ++j) acc[j] = acc[j] * g;
                    }
                    for (int64_t j = 0; j < len; ++j) out[j] = out[j] + acc[j];
                } else {
                    for (int64_t j = 0; j < len; ++j) out[j] = acc[j];
                }
            }
            for (int64_t j = 0; j < len; ++j) y[(base + j) * C + c] = out[j];
        }
    }
}

/* Batched entry: x, y are (batch, n, C).  threads <= 1 runs serially; otherwise
 * OpenMP splits (stream, tile) units across `threads` cores. */
int vnd_oracle_convolve_f32(const float *x, float *y, int64_t batch, int64_t n, int32_t C,
                            const int32_t *tap_off, const int32_t *idx, const float *w,
                            const int32_t *seg_off, const int32_t *seg_end,
                            const float *seg_gain, const uint8_t *chan_flags,
                            int apply_gain, int threads)
{
    if (batch < 0 || n < 0 || C <= 0) return 1;
    int64_t tiles = (n + TILE - 1) / TILE;
    int64_t units = batch * tiles;
    if (threads <= 1) {
        for (int64_t b = 0; b < batch; ++b)
            conv_stream(x + b * n * C, y + b * n * C, n, C, tap_off, idx, w, seg_off, seg_end,
                        seg_gain, chan_flags, apply_gain, 0, n);
        return 0;
    }
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t u = 0; u < units; ++u) {
        int64_t b = u / tiles, t = u % tiles;
        int64_t t0 = t * TILE, t1 = t0 + TILE < n ? t0 + TILE : n;
        conv_stream(x + b * n * C, y + b * n * C, n, C, tap_off, idx, w, seg_off, seg_end,
                    seg_gain, chan_flags, apply_gain, t0, t1);
    }
    return 0;
}

int vnd_oracle_abi_version(void) { return 1; }
