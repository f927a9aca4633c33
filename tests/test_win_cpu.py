"""CPU tier: the WINDOW form of the per-table kernel (vnd_win.hpp).  No GPU here:
* the generated tap function vw_taps() - one lane's whole tap sum, with its LDS addressing (ring bases,
  chunk planes, mirror) - is compiled for the HOST and run on an LDS image laid out by this file from the
  documented formulas, against the plain tap sum (the oracle's definition, decorrelation.py:649-658);
* the whole translation unit is cross-compiled for gfx950 and its ISA checked for what the design rests on."""
import ctypes
import zlib
import pathlib
import re
import shutil
import subprocess

import numpy as np
import pytest

REPO = pathlib.Path(__file__).resolve().parents[1]
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
CLANG = '/opt/rocm/lib/llvm/bin/clang++'


@pytest.fixture(scope='module')
def native():
    import __graft_entry__ as entry
    entry.build()
    from vndecorrelate_amd import _native
    _native.load_library()
    return _native


def _table(fir):
    from vndecorrelate_amd.taps import function_path_arrays
    a = function_path_arrays(fir)
    return a.tap_offsets, a.tap_index, a.tap_weight


def _macro(src, name):
    return int(re.search(rf'#define {name} (\d+)', src).group(1))


HOST_SHIM = r'''
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef char vw_lchar;
#define __device__
#define __forceinline__ inline
#define VW_RD(base, imm) (*(const volatile v4f *)((base) + (imm)))
#define VW_FMA(x, w, a) __builtin_elementwise_fma((x), v2f{(w), (w)}, (a))
#define VW_MUL(x, w) ((x) * v2f{(w), (w)})
#define VW_SB
static inline v2f vw_pair(v2f a, v2f b) { return v2f{a.y, b.x}; }
static inline float vw_add1(float a, float b) { return a + b; }
static inline float vw_sub1(float a, float b) { return a - b; }
%(defines)s
#define VW_NBT VW_NB          /* (stereo forms: no tail behind the ring - the quad / octet forms' extra bases are the GPU tests' business) */
%(function)s
static unsigned wrap(unsigned a) { const unsigned b = a - (unsigned)VW_R; return a < b ? a : b; }
extern "C" void run_lane(char *lds, unsigned pl, float *o0, float *o1)
{
    vw_lchar *b[2][VW_NB];
    for (int k = 0; k < VW_NB; ++k) { b[0][k] = lds + wrap(pl + (unsigned)(k * VW_G)) * 16u; b[1][k] = b[0][k] + (VW_M / 4) * VW_PLANE; }
    float a0[VW_M], a1[VW_M];
    vw_taps(b, a0, a1);
    for (int j = 0; j < VW_M; ++j) { o0[j] = a0[j]; o1[j] = a1[j]; }
}
'''


def _host_lane(src, tmp_path, tag, pair=0):
    """pair: the channel pair whose tap function is compiled (vw_taps, vw_taps_1, ...)"""
    defines = '\n'.join(l for l in src.splitlines() if re.match(r'#define VW_(NT|M|R|G|NB|DE|PLANE|LA) ', l))
    name = 'vw_taps' if pair == 0 else f'vw_taps_{pair}'
    fn = src[src.index(f'__device__ __forceinline__ void {name}('):]
    fn = fn[:fn.index('\n}\n') + 3].replace(f'void {name}(', 'void vw_taps(')
    cpp = tmp_path / f'lane_{tag}.cpp'
    cpp.write_text(HOST_SHIM % dict(defines=defines, function=fn))
    so = tmp_path / f'lane_{tag}.so'
    r = subprocess.run([CLANG, '-O1', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', str(cpp), '-o', str(so)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = ctypes.CDLL(str(so))
    lib.run_lane.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def _lds_image(x, own_position, M, R, G, plane):
    """x: (frames, 2) float32, the lane's view: frame 0 = its first own frame.  Ring entry (own + d) mod R holds
    frames [d*M, (d+1)*M); chunk r of the entry at position p is at r*plane + p*16 (+ (M/4)*plane for channel 1);
    positions R .. R+G-1 mirror positions 0 .. G-1."""
    qc = M // 4
    img = np.full(2 * qc * plane, np.nan, np.float32).view(np.uint8)
    for d in range(R):
        p = (own_position + d) % R
        frames = x[d * M:(d + 1) * M]
        if len(frames) < M:
            frames = np.vstack([frames, np.zeros((M - len(frames), 2), np.float32)])
        for c in range(2):
            for r in range(qc):
                for pos in ([p, p + R] if p < G else [p]):
                    at = c * qc * plane + r * plane + pos * 16
                    img[at:at + 16] = frames[4 * r:4 * r + 4, c].copy().view(np.uint8)
    return img


def _want(x, offs, idx, w, M):
    out = np.zeros((M, 2), np.float64)
    for c in range(2):
        for i, wt in zip(idx[offs[c]:offs[c + 1]], w[offs[c]:offs[c + 1]]):
            out[:, c] += np.float64(wt) * x[i:i + M, c]
    return out


def _random_table(rng, taps, span):
    offs, idx, w = [0], [], []
    for c in range(2):
        k = int(rng.integers(1, min(taps, span) + 1))
        ii = np.sort(rng.choice(span, size=k, replace=False))
        idx += list(ii)
        w += list(rng.choice([0.85, -0.85, 0.55, -0.55, 0.35, -0.2, 1.0, -1.0], size=k))
        offs.append(len(idx))
    return np.array(offs, np.int32), np.array(idx, np.int32), np.array(w, np.float32)


@pytest.mark.parametrize('case', ['g48k_k30', 'g48k_k128_u', 'random_a', 'random_b', 'random_c', 'head'])
@pytest.mark.parametrize('M,nt', [(32, 128), (16, 256), (64, 64)])
def test_one_lane_of_the_generated_tap_function(native, golden, tmp_path, case, M, nt):
    rng = np.random.default_rng(zlib.crc32(f'{case}/{M}'.encode()))
    if case.startswith('g48k'):
        offs, idx, w = _table(golden.fir(case))
    elif case == 'head':            # taps 0 and 1, the last offsets of a short filter: both parities at both edges of a run
        offs, idx, w = np.array([0, 3, 6], np.int32), np.array([0, 1, 95, 1, 2, 94], np.int32), np.array([1, -1, .5, .25, 2, -3], np.float32)
    else:
        offs, idx, w = _random_table(rng, 40, int(rng.integers(8, 1500)))
    src, lds_bytes, fmas = native.window_kernel_source(offs, idx, w, 2, M, nt, with_traffic=True)
    R, G, plane = _macro(src, 'VW_R'), _macro(src, 'VW_G'), _macro(src, 'VW_PLANE')
    # the ring: the tile and the halo of the farthest tap - rounded up to a multiple of 16 entries, so that the ring's end falls on
    # a bank period (the wave that straddles it reads conflict-free) - unless those entries would cost a workgroup of residency
    need = nt + (int(idx.max()) + M - 1) // M
    assert need <= R < need + 16 and (R % 16 == 0 or R == need) and _macro(src, 'VW_NB') * G > R - nt
    qc = M // 4                     # planes an odd multiple of 8/QC slots apart: the 8-byte accesses of 16 lanes fill 32 banks once
    assert (plane // 16) >= R + G     # (32-frame runs: 2 mod 4, with the lanes' pair indices swizzled - reads and writes conflict-free)
    assert (plane // 16) % 4 == 2 and '#define VW_LANE_SWIZZLE 1' in src if M == 32 else (plane // 16) % max(2, 16 // qc) == max(1, 8 // qc)
    assert fmas == M * len(idx) and lds_bytes <= 4 * fmas * (M + 4) // M + 64       # never worse than a window per tap
    lib = _host_lane(src, tmp_path, f'{case}_{M}')
    x = rng.uniform(-1, 1, (R * M, 2)).astype(np.float32)
    want = _want(x.astype(np.float64), offs, idx, w, M)
    peak = np.abs(want).max()
    for own in (0, 1, R - 1, R - G, nt, int(rng.integers(0, R))):       # ring positions either side of the wrap
        img = _lds_image(x, own, M, R, G, plane)
        o0, o1 = np.zeros(M, np.float32), np.zeros(M, np.float32)
        lib.run_lane(img.ctypes.data, own, o0.ctypes.data, o1.ctypes.data)
        got = np.stack([o0, o1], 1).astype(np.float64)
        assert np.all(np.isfinite(got)), f'own={own}: a read outside the window (NaN filler)'
        assert np.abs(got - want).max() <= 1e-6 * peak, f'own={own}'


@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_every_channel_pair_of_a_wider_table_gets_its_own_tap_function(native, golden, tmp_path, mode):
    """More than two channels: a workgroup takes one channel PAIR, and the source carries one tap function per pair
    (vw_taps, vw_taps_1, ...: that pair's taps, the same two LDS plane sets), vw_taps_of<PG> to pick one and the
    kernel's dispatch over the pairs."""
    from vndecorrelate_amd.taps import function_path_arrays
    fir = np.ascontiguousarray(golden.fir('g96k_k64_c8')[:, :6])
    arr = function_path_arrays(fir)
    offs, idx, w = arr.tap_offsets, arr.tap_index, arr.tap_weight
    M, nt = 32, 128
    src, lds_bytes, fmas = native.window_kernel_source(offs, idx, w, 2 if mode == 'fast' else 0, M, nt, with_traffic=True)
    assert '#define VW_C 6' in src and fmas == M * len(idx)
    assert all(f'if constexpr (PG == {g}) ' in src and f'case {g}: vw_span<{g}>(a, lds, stream, t_first, ntiles, flags, pace); break;' in src for g in range(3))
    assert 'vw_taps_3(' not in src
    R, G, plane = _macro(src, 'VW_R'), _macro(src, 'VW_G'), _macro(src, 'VW_PLANE')
    need = nt + (int(idx.max()) + M - 1) // M                   # the halo of the farthest tap of ANY pair
    assert need <= R < need + 16 and (R % 16 == 0 or R == need)
    rng = np.random.default_rng(77)
    for pair in range(3):
        lib = _host_lane(src, tmp_path, f'wide_{mode}_{pair}', pair)
        x = rng.uniform(-1, 1, (R * M, 2)).astype(np.float32)
        o2 = np.array([0, offs[2 * pair + 1] - offs[2 * pair], offs[2 * pair + 2] - offs[2 * pair]], np.int32)
        i2, w2 = idx[offs[2 * pair]:offs[2 * pair + 2]], w[offs[2 * pair]:offs[2 * pair + 2]]
        for own in (0, R - 1, nt + 3):
            img = _lds_image(x, own, M, R, G, plane)
            o0, o1 = np.zeros(M, np.float32), np.zeros(M, np.float32)
            lib.run_lane(img.ctypes.data, own, o0.ctypes.data, o1.ctypes.data)
            got = np.stack([o0, o1], 1)
            if mode == 'fast':
                want = _want(x.astype(np.float64), o2, i2, w2, M)
                assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max(), f'pair {pair} own={own}'
            else:                                               # the reference's order: one float32 accumulator, taps in table order
                want = np.zeros((M, 2), np.float32)
                for c in range(2):
                    for i, wt in zip(i2[o2[c]:o2[c + 1]], w2[o2[c]:o2[c + 1]]):
                        want[:, c] = want[:, c] + x[i:i + M, c] * np.float32(wt)
                assert np.array_equal(got, want), f'pair {pair} own={own}'


def _want_exact(x, arr, M):
    """The reference's association in float32, op by op: taps in table order, separately rounded products and sums;
    class path: per segment (sum of -x, then of +x) * gain, segments summed (decorrelation.py:402-414, :656-658)."""
    out = np.zeros((M, 2), np.float32)
    for c in range(2):
        k0, k1 = arr.tap_offsets[c], arr.tap_offsets[c + 1]
        if arr.seg_offsets is None:
            acc = np.zeros(M, np.float32)
            for i, wt in zip(arr.tap_index[k0:k1], arr.tap_weight[k0:k1]):
                acc = acc + x[i:i + M, c] * np.float32(wt)
            out[:, c] = acc
            continue
        total = np.zeros(M, np.float32)
        prev = k0
        for sg in range(arr.seg_offsets[c], arr.seg_offsets[c + 1]):
            sb = np.zeros(M, np.float32)
            for k in range(prev, arr.seg_end[sg]):
                i = arr.tap_index[k]
                sb = sb - x[i:i + M, c] if arr.tap_weight[k] < 0 else sb + x[i:i + M, c]
            if arr.apply_gain:
                sb = sb * np.float32(arr.seg_gain[sg])
            total = total + sb
            prev = arr.seg_end[sg]
        out[:, c] = total
    return out


@pytest.mark.parametrize('case', ['fn_g48k_k30', 'fn_g48k_k128_l', 'fn_unordered', 'class_default', 'class_noenv', 'class_dense'])
@pytest.mark.parametrize('M,nt', [(32, 128), (16, 64), (64, 64)])
def test_one_lane_of_the_exact_tap_function_is_bit_identical(native, golden, tmp_path, case, M, nt):
    """VND_MODE_EXACT in the window form: the generated lane function against the reference's float32 association
    emulated op by op - function-path tables (ascending, and one in scrambled order: a pass per run), class-path
    tables (segments of -1 / +1 taps with gains, identity envelope, duplicates)."""
    from vndecorrelate_amd.taps import TapArrays, class_path_arrays, function_path_arrays
    import vndecorrelate_amd.decorrelation as d
    rng = np.random.default_rng(zlib.crc32(f'{case}/{M}'.encode()))
    if case.startswith('fn_g'):
        arr = function_path_arrays(golden.fir(case[3:]))
    elif case == 'fn_unordered':
        offs, idx, w = _random_table(rng, 24, 900)
        for c in range(2):                           # table order is the caller's: scramble one channel, duplicate an offset
            seg = slice(offs[c], offs[c + 1])
            if c == 0:
                perm = rng.permutation(offs[1] - offs[0])
                idx[seg], w[seg] = idx[seg][perm], w[seg][perm]
        arr = TapArrays(offs, idx, w)
    else:
        kw = dict(class_default=dict(), class_noenv=dict(segment_envelope=()),
                  class_dense=dict(num_impulses=128, duration_seconds=0.03))[case]
        arr = d.VelvetNoise(sample_rate_hz=48000, seed=3, **kw)._tap_arrays()
        assert arr.seg_offsets is not None and arr.chan_flags.sum() == 0
    src = native.window_kernel_source(arr.tap_offsets, arr.tap_index, arr.tap_weight, 0, M, nt, seg_offsets=arr.seg_offsets,
                                      seg_end=arr.seg_end, seg_gain=arr.seg_gain, apply_gain=arr.apply_gain)
    assert '#define VW_EXACT 1' in src and 'VW_FMA(' not in src[src.index('void vw_taps('):src.index('#if VW_EPI')]
    R, G, plane = _macro(src, 'VW_R'), _macro(src, 'VW_G'), _macro(src, 'VW_PLANE')
    lib = _host_lane(src, tmp_path, f'ex_{case}_{M}')
    x = rng.uniform(-1, 1, (R * M, 2)).astype(np.float32)
    x[rng.integers(0, len(x), 40), rng.integers(0, 2, 40)] = 0.0          # exact zeros: the sums must start from +0
    want = _want_exact(x, arr, M)
    for own in (0, R - 1, int(rng.integers(0, R))):
        img = _lds_image(x, own, M, R, G, plane)
        o0, o1 = np.zeros(M, np.float32), np.zeros(M, np.float32)
        lib.run_lane(img.ctypes.data, own, o0.ctypes.data, o1.ctypes.data)
        got = np.stack([o0, o1], 1)
        assert got.tobytes() == want.tobytes(), f'own={own}: {np.abs(got - want).max():.3e}'


@pytest.mark.parametrize('case', ['g48k_k30', 'g48k_k128_l', 'g44k_k30'])
@pytest.mark.parametrize('M,nt', [(32, 128), (16, 64)])
def test_one_lane_of_the_merged_exact_fanout_function_is_bit_identical(native, golden, tmp_path, case, M, nt, monkeypatch):
    """A mono input through a function-path stereo table, VND_MODE_EXACT (decorrelation.py:428-431 then :649-658): ONE ascending read
    stream over the union of both channels' windows, the products f32(x * |w|) shared by the two channels - and still each channel's
    sums in table order, bit for bit the reference's.  Fewer window reads than a pass per channel."""
    from vndecorrelate_amd.taps import function_path_arrays
    rng = np.random.default_rng(zlib.crc32(f'merged/{case}/{M}'.encode()))
    arr = function_path_arrays(golden.fir(case))
    monkeypatch.setenv('VND_WIN_SOURCE_FANOUT', '1')
    src = native.window_kernel_source(arr.tap_offsets, arr.tap_index, arr.tap_weight, 0, M, nt)
    assert '#define VW_EXACT 1' in src and '#define VW_BC 1' in src
    body = src[src.index('void vw_taps('):src.index('#if VW_EPI')]
    assert 'S0[' in body and 'S1[' in body and 'VW_RD(b[1]' not in body          # both channels' sums from the one plane set
    monkeypatch.setenv('VND_WIN_EXACT_MERGED', '0')
    apart = native.window_kernel_source(arr.tap_offsets, arr.tap_index, arr.tap_weight, 0, M, nt)
    reads = lambda text: len(re.findall(r'= VW_RD\(', text[text.index('void vw_taps('):text.index('#if VW_EPI')]))
    products = lambda text: len(re.findall(r'const v2f p\d+ = ', text[text.index('void vw_taps('):text.index('#if VW_EPI')]))
    assert reads(src) < 0.75 * reads(apart) and products(src) < products(apart)
    R, G, plane = _macro(src, 'VW_R'), _macro(src, 'VW_G'), _macro(src, 'VW_PLANE')
    lib = _host_lane(src, tmp_path, f'exm_{case}_{M}')
    mono = rng.uniform(-1, 1, R * M).astype(np.float32)
    mono[rng.integers(0, len(mono), 40)] = 0.0
    x = np.stack([mono, mono], 1)
    want = _want_exact(x, arr, M)
    for own in (0, R - 1, int(rng.integers(0, R))):
        img = _lds_image(x, own, M, R, G, plane)
        o0, o1 = np.zeros(M, np.float32), np.zeros(M, np.float32)
        lib.run_lane(img.ctypes.data, own, o0.ctypes.data, o1.ctypes.data)
        got = np.stack([o0, o1], 1)
        assert got.tobytes() == want.tobytes(), f'own={own}: {np.abs(got - want).max():.3e}'


@pytest.mark.parametrize('mags,order', [(3, 'runs'), (8, 'runs'), (3, 'scrambled'), (9, 'runs'), (20, 'scrambled')])
def test_adds_per_segment_only_for_tables_of_few_gains_in_runs(native, tmp_path, mags, order):
    """The fast mode's adds-per-segment association (decorrelation.py:402-414 as a summation order) is taken for tables with at most 8
    distinct |w| that come in RUNS along the offsets (a segmented envelope: at most 8 changes per channel) - a chain changes its unit
    with one FMA per change; a table of more gains, or of few gains in scrambled order, keeps one FMA per tap.  Either way one lane's
    generated function equals the plain tap sum."""
    rng = np.random.default_rng(900 + mags)
    M, nt = 32, 128
    gains = np.round(rng.uniform(0.05, 1.5, mags), 3).astype(np.float32)
    offs, idx, w = [0], [], []
    for c in range(2):
        k = 40
        ii = np.sort(rng.choice(700, size=k, replace=False))
        idx += list(ii)
        g = rng.choice(gains, size=k) if order == 'scrambled' else gains[(np.arange(k) * min(mags, 9)) // k]
        w += list(g * rng.choice([-1.0, 1.0], size=k))
        offs.append(len(idx))
    offs, idx, w = np.array(offs, np.int32), np.array(idx, np.int32), np.array(w, np.float32)
    src = native.window_kernel_source(offs, idx, w, 2, M, nt)
    body = src[src.index('void vw_taps('):src.index('#if VW_EPI')]
    adds = 'Z2 +' in body or 'Z2 -' in body
    distinct = len(set(np.abs(w).tolist()))
    assert adds == (distinct <= 8 and order == 'runs'), (mags, distinct, order)
    if adds:
        changes = sum(int((np.abs(w[offs[c]:offs[c + 1]])[1:] != np.abs(w[offs[c]:offs[c + 1]])[:-1]).sum()) for c in range(2))
        assert body.count('VW_FMA(') <= changes * M                    # a unit change per run and chain (M / 2 E and P chains each), nothing else multiplies
    else:
        assert body.count('VW_FMA(') >= len(idx) * (M // 2) - len(idx) - 2 * M        # one FMA per (tap, output pair)
    R, G, plane = _macro(src, 'VW_R'), _macro(src, 'VW_G'), _macro(src, 'VW_PLANE')
    lib = _host_lane(src, tmp_path, f'mags{mags}{order}')
    x = rng.uniform(-1, 1, (R * M, 2)).astype(np.float32)
    want = _want(x.astype(np.float64), offs, idx, w, M)
    peak = np.abs(want).max()
    for own in (0, R - 1, 7):
        img = _lds_image(x, own, M, R, G, plane)
        o0, o1 = np.zeros(M, np.float32), np.zeros(M, np.float32)
        lib.run_lane(img.ctypes.data, own, o0.ctypes.data, o1.ctypes.data)
        assert np.abs(np.stack([o0, o1], 1) - want).max() <= 1e-6 * peak


def test_window_reads_fewer_lds_bytes_than_a_read_per_tap(native, golden):
    """The point of the form: bytes read from LDS per (tap, output) product; the pair-read kernel pays 4."""
    per = {}
    for name in ('g48k_k30', 'g48k_k128_u'):
        offs, idx, w = _table(golden.fir(name))
        for M in (16, 32, 64):
            _, b, f = native.window_kernel_source(offs, idx, w, 2, M, 64, with_traffic=True)
            per[name, M] = b / f
    assert per['g48k_k30', 32] < 2.7 and per['g48k_k128_u', 32] < 1.5 and per['g48k_k128_u', 64] < 0.8
    assert per['g48k_k30', 16] > per['g48k_k30', 32] > per['g48k_k30', 64]


@pytest.mark.parametrize('adds', [1, 0])
@pytest.mark.parametrize('gname,M,nt', [('g48k_k30', 32, 256), ('g48k_k128_u', 32, 128)])
def test_window_source_compiles_for_gfx950_with_the_intended_isa(native, golden, tmp_path, gname, M, nt, adds, monkeypatch):
    offs, idx, w = _table(golden.fir(gname))
    monkeypatch.setenv('VND_WIN_ADDS', str(adds))
    src = native.window_kernel_source(offs, idx, w, 2, M, nt)
    f = tmp_path / 'k.hip'
    f.write_text(src)
    out = tmp_path / 'k.s'
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only',
                        '-include', 'hip/hip_runtime.h', '-S', str(f), '-o', str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    assert re.search(r'ScratchSize: 0\b', asm), 'the window kernel must not spill'
    ops = re.findall(r'^\s+([a-z0-9_]+)', asm, re.M)
    count = {o: ops.count(o) for o in set(ops)}
    assert count.get('flat_load_dwordx4', 0) == 0, 'LDS reads fell back to flat loads'
    taps = len(idx)
    # a tap costs M/2 packed operations when its offset is even, M/2 - 1 packed + 2 single ones when odd
    odd = int((idx & 1).sum())
    if adds:
        # the reference's class-path association (decorrelation.py:402-414): a v_pk_add_f32 per (tap, output pair) inside a run of
        # equal |w| - the sign is a source modifier - a v_pk_fma_f32 (sum x gain ratio +- x) where a chain meets the next |w|, and one
        # v_pk_mul_f32 per finished output pair: the table's 4 gains make 3 FMAs per chain of E / P accumulators
        chains = 2 * (M // 2 + M // 2 - 1)
        assert count['v_pk_fma_f32'] <= 3 * chains + 8 and count['v_pk_mul_f32'] == 2 * (M // 2)
        assert taps * (M // 2) - odd - 3 * chains - 16 <= count['v_pk_add_f32'] <= taps * (M // 2) - odd - 3 * chains + 2 * (M // 2)      # (+ the merge, where hipcc packs it)
        assert 'neg_lo:[0,1] neg_hi:[0,1]' in asm or 'neg_lo:[1,0] neg_hi:[1,0]' in asm
    else:
        # one FMA per tap; the first product of an accumulator is a multiply
        packed = count['v_pk_fma_f32'] + count.get('v_pk_mul_f32', 0)
        assert taps * (M // 2) - odd - 16 <= packed <= taps * (M // 2) - odd        # (hipcc may split a few first products)
    assert count['s_barrier'] == 3                        # the prologue's and the two of a tile
    # one 16-byte read per chunk of the union of the windows, each an immediate offset from a base register
    n_reads = len(re.findall(r'q\[\d+\] = VW_RD\(', src))
    assert n_reads <= count['ds_read_b128'] <= n_reads + 2 * (M // 4)
    # ... so the tap phase carries (almost) no address arithmetic
    assert count.get('v_add_u32_e32', 0) < 200


@pytest.mark.parametrize('mode', [2, 0])
def test_split_form_source_keeps_three_waves_per_simd(native, golden, tmp_path, mode, monkeypatch):
    """VND_WIN_SPLIT: a wave computes ONE channel of a stereo table (vw_taps_c0 / vw_taps_c1) - one channel's accumulators per
    lane, so the dense 128-tap table builds for three waves per SIMD (<= 168 registers) without spilling."""
    offs, idx, w = _table(golden.fir('g48k_k128_u'))
    M, nt = 32, 256
    monkeypatch.setenv('VND_WIN_SPLIT', '2')
    src = native.window_kernel_source(offs, idx, w, mode, M, nt)
    assert _macro(src, 'VW_S') == 1 and _macro(src, 'VW_Q') == 0 and _macro(src, 'VW_WAVES_PER_EU') == 3
    assert _macro(src, 'VW_R') == nt // 2 + _macro(src, 'VW_DE')
    assert 3 * 2 * (M // 4) * _macro(src, 'VW_PLANE') <= 160 * 1024           # three workgroups per CU
    assert 'vw_taps_c0(' in src and 'vw_taps_c1(' in src and '#define VW_DISPATCH(pg) vw_span_s(' in src
    f = tmp_path / 'k.hip'
    f.write_text(src)
    out = tmp_path / 'k.s'
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only',
                        '-include', 'hip/hip_runtime.h', '-S', str(f), '-o', str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    assert re.search(r'ScratchSize: 0\b', asm), 'the split form of the dense table must not spill'
    assert int(re.search(r'; NumVgprs: (\d+)', asm).group(1)) <= 168
    ops = re.findall(r'^\s+([a-z0-9_]+)', asm, re.M)
    assert ops.count('s_barrier') == 4                       # the prologue's and the three of a tile
    monkeypatch.delenv('VND_WIN_SPLIT')
    assert _macro(native.window_kernel_source(offs, idx, w, mode, M, nt), 'VW_S') == 0      # off by default


@pytest.mark.parametrize('mode,late,la', [(0, 8, 3), (2, 15, 2)])
def test_split_form_with_64_frame_runs_fits_two_waves_per_simd(native, golden, tmp_path, mode, late, la, monkeypatch):
    """64-frame runs halve the LDS reads per FMA but need 64 (exact: sums) or 128 (fast: E and P) accumulator registers per
    channel: only the split form - one channel per lane - holds them at two waves per SIMD, the fast mode with fifteen of a
    wave's sixteen refill accesses per tile loaded late (VW_LATE) and the per-access constants opaque per tile.  The dense
    128-tap table and (fast mode) the headline's 30-tap table, cross-compiled: no spill."""
    offs, idx, w = _table(golden.fir('g48k_k128_u'))
    M, nt = 64, 256
    monkeypatch.setenv('VND_WIN_SPLIT', '2')
    monkeypatch.setenv('VND_SPEC_LA', str(la))           # (what the launch plan picks for this form: 3 reads ahead exact, 2 fast)
    src = native.window_kernel_source(offs, idx, w, mode, M, nt)
    assert _macro(src, 'VW_S') == 1 and _macro(src, 'VW_M') == 64 and _macro(src, 'VW_LATE') == late and _macro(src, 'VW_LA') == la
    assert _macro(src, 'VW_WAVES_PER_EU') == 2
    assert 2 * 2 * (M // 4) * _macro(src, 'VW_PLANE') <= 160 * 1024           # two workgroups per CU
    f = tmp_path / 'k.hip'
    f.write_text(src)
    out = tmp_path / 'k.s'
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only',
                        '-include', 'hip/hip_runtime.h', '-S', str(f), '-o', str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    assert re.search(r'ScratchSize: 0\b', asm), 'the 64-frame split form of the dense table must not spill'
    assert int(re.search(r'; NumVgprs: (\d+)', asm).group(1)) <= 240            # (256 before the per-access constants left the loop)
    # half the window reads of the 32-frame form for the same sums
    monkeypatch.setenv('VND_WIN_SPLIT', '0')
    monkeypatch.delenv('VND_SPEC_LA')
    plain = native.window_kernel_source(offs, idx, w, mode, 32, 256)
    reads64, reads32 = len(re.findall(r'= VW_RD\(', src)), len(re.findall(r'= VW_RD\(', plain))
    assert reads64 < 1.1 * reads32            # ... per 64 frames instead of per 32


@pytest.mark.parametrize('mode', [2, 0])
@pytest.mark.parametrize('Q', [2, 1])
def test_quads_and_octets_with_a_wave_per_channel(native, golden, tmp_path, mode, Q, monkeypatch):
    """cfg5's table by default: one workgroup of 512 lanes per OCTET (Q = 2: whole 32-byte frames), a WAVE per channel (vw_span_qc,
    vw_taps_<pair>c<channel>) - one channel's accumulators per lane, 32-frame runs on a 2048-frame tile, the 2720-frame halo of
    all eight channels still inside 160 KB; VND_WIN_OCTET=0 (and tables of 4k channels): a QUAD per workgroup of 256 lanes;
    VND_WIN_QUAD=0: the form on channel pairs.  Cross-compiled: no spill, whole 16-byte pieces both ways, one instantiation of
    the span loop per quad / octet (prologue barrier + two per tile: a wave is the only reader of its channel's planes)."""
    offs, idx, w = _table(golden.fir('g96k_k64_c8'))
    M, nt = 32, 256 * Q
    monkeypatch.setenv('VND_WIN_OCTET', '1' if Q == 2 else '0')
    src = native.window_kernel_source(offs, idx, w, mode, M, nt)
    assert _macro(src, 'VW_Q') == Q and _macro(src, 'VW_S') == 0 and _macro(src, 'VW_C') == 8
    # the ring is cut to a multiple of 16 entries (a wave that straddles its end stays in step with the banks); what is cut off - entries
    # that only the tile's last lanes reach - lives in a TAIL behind the mirror, read through per-lane bases of its own
    R, tail, DE, NB = (_macro(src, k) for k in ('VW_R', 'VW_TAIL', 'VW_DE', 'VW_NB'))
    assert R + tail == nt // (4 * Q) + DE and R % 16 == 0 and 0 < tail < 16
    assert _macro(src, 'VW_PLANE') // 16 >= R + _macro(src, 'VW_G') + tail
    tail_reads = re.findall(r'VW_RD\(b\[\d\]\[(\d+)\], (\d+)\)', src)
    used = {int(k) for k, _ in tail_reads if int(k) >= NB}
    assert used and used <= set(range(NB, NB + tail))                                               # one base per entry offset past the ring
    assert all(int(imm) % _macro(src, 'VW_PLANE') == 0 for k, imm in tail_reads if int(k) >= NB)    # ... and only the chunk plane as immediate
    assert 2 * Q * 2 * (M // 4) * _macro(src, 'VW_PLANE') <= 160 * 1024
    for pg in range(4):
        for ch in range(2):
            assert ('vw_taps_c%d(' % ch if pg == 0 else 'vw_taps_%dc%d(' % (pg, ch)) in src
    dispatch = src.split('#define VW_DISPATCH')[1].split('\n')[0]
    assert dispatch.count('vw_span_qc<') == 2 // Q and 'vw_span_qc<0>(a, lds, stream, t_first, ntiles, flags, pace)' in dispatch
    f = tmp_path / 'k.hip'
    f.write_text(src)
    out = tmp_path / 'k.s'
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only',
                        '-include', 'hip/hip_runtime.h', '-S', str(f), '-o', str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    assert re.search(r'ScratchSize: 0\b', asm), 'the quad / octet kernel must not spill'
    ops = re.findall(r'^\s+([a-z0-9_]+)', asm, re.M)
    assert ops.count('flat_load_dwordx4') == 0, 'LDS reads fell back to flat loads'
    assert ops.count('s_barrier') == (2 // Q) * 3 and ops.count('buffer_store_dwordx4') == (2 // Q) * (M // 4)
    assert ops.count('buffer_load_dwordx2') == 0 and ops.count('buffer_store_dwordx2') == 0
    # VND_WIN_QUAD=0: a workgroup per channel PAIR (8-byte pieces)
    monkeypatch.setenv('VND_WIN_QUAD', '0')
    pairs = native.window_kernel_source(offs, idx, w, mode, M, 256)
    assert _macro(pairs, 'VW_Q') == 0 and _macro(pairs, 'VW_R') == 256 + _macro(pairs, 'VW_DE')


@pytest.mark.parametrize('Q', [2, 1])
def test_quads_and_octets_leave_the_normalisers_sums_in_the_store_phase(native, golden, tmp_path, Q, monkeypatch):
    """VelvetNoise.decorrelate on 4k channels (LR mode: the per-channel RMS normaliser alone, decorrelation.py:433-440) in the fast
    mode: the quad / octet kernel built with VW_EPI adds up the squares of a lane's input run (still in its ring entry) and of its
    outputs (in registers) before the store phase overwrites the entry - 8 more 16-byte LDS reads per lane and tile, two float64
    butterflies per wave, one row of 2 C sums per (tile, channel wave).  Cross-compiled: still no spill, the same stores, the
    same barriers, and the float64 adds are there."""
    offs, idx, w = _table(golden.fir('g96k_k64_c8'))
    M, nt = 32, 256 * Q
    monkeypatch.setenv('VND_WIN_OCTET', '1' if Q == 2 else '0')
    plain = native.window_kernel_source(offs, idx, w, 2, M, nt)
    monkeypatch.setenv('VND_WIN_SOURCE_EPI', '1')
    src = native.window_kernel_source(offs, idx, w, 2, M, nt)
    assert _macro(plain, 'VW_EPI') == 0 and _macro(src, 'VW_EPI') == 1 and _macro(src, 'VW_Q') == Q
    asm = {}
    for tag, text in (('plain', plain), ('epi', src)):
        f = tmp_path / f'{tag}.hip'
        f.write_text(text)
        out = tmp_path / f'{tag}.s'
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only',
                            '-include', 'hip/hip_runtime.h', '-S', str(f), '-o', str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        asm[tag] = out.read_text()
        assert re.search(r'ScratchSize: 0\b', asm[tag]), f'{tag}: the quad / octet kernel must not spill'
    ops = {tag: re.findall(r'^\s+([a-z0-9_]+)', text, re.M) for tag, text in asm.items()}
    spans = 2 // Q
    assert ops['epi'].count('s_barrier') == ops['plain'].count('s_barrier') == spans * 3
    assert ops['epi'].count('buffer_store_dwordx4') == ops['plain'].count('buffer_store_dwordx4')
    assert ops['plain'].count('v_add_f64') == 0 and ops['epi'].count('v_add_f64') >= spans * 12          # 2 sums x 6 butterfly steps
    assert ops['epi'].count('ds_bpermute_b32') == spans * 24
    assert ops['epi'].count('ds_read_b128') - ops['plain'].count('ds_read_b128') == spans * (M // 4)   # the lane's own input run
    # the exact mode's build leaves the same two sums per (tile, channel wave) - there they are the per-block predictions the
    # block-parallel NumPy-order sums (rms_par_*) start from, instead of a pass of their own over x and y
    exact = native.window_kernel_source(offs, idx, w, 0, M, nt)
    assert _macro(exact, 'VW_EPI') == 1 and _macro(exact, 'VW_EXACT') == 1
    f = tmp_path / 'exact_epi.hip'
    f.write_text(exact)
    out = tmp_path / 'exact_epi.s'
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only',
                        '-include', 'hip/hip_runtime.h', '-S', str(f), '-o', str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    assert re.search(r'ScratchSize: 0\b', text), 'the exact quad / octet kernel with block sums must not spill'
    assert len(re.findall(r'^\s+v_add_f64', text, re.M)) >= spans * 12


def test_six_channels_ride_two_overlapping_quads(native, golden, tmp_path, monkeypatch):
    """4k + 2 channels (6, 10, ...): the quad form with k full quads and one more that starts at channel C - 4 - two workgroups compute
    and write the shared channel pair with the same bits, every access is still 16 bytes of a frame.  Generated source: two span
    instantiations, the second one's tap functions are those of pairs 1 and 2; cross-compiled without spills."""
    fir = np.ascontiguousarray(golden.fir('g96k_k64_c8')[:, :6])
    offs, idx, w = _table(fir)
    monkeypatch.setenv('VND_WIN_OCTET', '0')
    for mode in (2, 0):
        src = native.window_kernel_source(offs, idx, w, mode, 32, 256)
        assert _macro(src, 'VW_Q') == 1 and _macro(src, 'VW_C') == 6
        dispatch = src.split('#define VW_DISPATCH')[1].split('\n')[0]
        assert dispatch.count('vw_span_qc<') == 2
        body = src.split('vw_taps_of_channel(int pc')[1].split('#define VW_TAPS_OF_CHANNEL')[0]
        second = body.split('QD == 1')[1]
        assert 'vw_taps_1c0(' in second and 'vw_taps_2c1(' in second and 'vw_taps_c0(' not in second
        f = tmp_path / f'k{mode}.hip'
        f.write_text(src)
        out = tmp_path / f'k{mode}.s'
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only',
                            '-include', 'hip/hip_runtime.h', '-S', str(f), '-o', str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert re.search(r'ScratchSize: 0\b', out.read_text())
