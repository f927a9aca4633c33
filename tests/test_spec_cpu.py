"""CPU tier: the per-table (hipRTC) fast kernel's source generator.  No GPU here: the source the
library would compile is fetched through the C ABI, cross-compiled for gfx950 with hipcc, and its
ISA is checked for the properties the design rests on."""
import pathlib
import re
import shutil
import subprocess

import numpy as np
import pytest

REPO = pathlib.Path(__file__).resolve().parents[1]
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


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


def _array(src, name):
    body = re.search(rf'{name}\[1\]\[\d+\] = \{{\{{([^}}]*)\}}', src).group(1).strip(',').split(',')
    return body


def test_source_carries_the_table(native, golden):
    """The generated read schedule covers every (tap, row) exactly once: each unique LDS read
    (plane, (i & ~1) + 2*NT*row) lists the FMAs it feeds."""
    fir = golden.fir('g48k_k30')
    offs, idx, w = _table(fir)
    src = native.spec_kernel_source(offs, idx, w)
    assert _macro(src, 'VS_C') == 2 and _macro(src, 'VS_GROUPS') == 1
    nt, rr = _macro(src, 'VS_NT'), _macro(src, 'VS_RR')
    n_reads = int(re.search(r'VS_RD_N\[1\] = \{(\d+),', src).group(1))
    plane = [int(v) for v in _array(src, 'VS_RD_PLANE')][:n_reads]
    off = [int(v) for v in _array(src, 'VS_RD_OFF')][:n_reads]
    first = [int(v) for v in _array(src, 'VS_RD_FIRST')][:n_reads + 1]
    cs_set = [int(v) for v in _array(src, 'VS_CS_SET')]
    cs_row = [int(v) for v in _array(src, 'VS_CS_ROW')]
    cs_w = [float.fromhex(v.rstrip('f')) for v in _array(src, 'VS_CS_W')]
    assert len(set(zip(plane, off))) == n_reads, 'a read is scheduled twice'
    assert first[0] == 0 and first[-1] == len(idx) * rr and n_reads <= len(idx) * rr
    got = []
    for k in range(n_reads):
        for m in range(first[k], first[k + 1]):
            assert cs_set[m] >> 1 == plane[k]
            got.append((cs_set[m] >> 1, off[k] - 2 * nt * cs_row[m], cs_set[m] & 1, cs_row[m], np.float32(cs_w[m])))
    want = sorted((c, int(i) & ~1, int(i) & 1, j, np.float32(wt)) for c in range(2)
                  for i, wt in zip(idx[offs[c]:offs[c + 1]], w[offs[c]:offs[c + 1]]) for j in range(rr))
    assert sorted(got) == want
    # the span-end chain repeats row 0's odd accumulation order
    for c in range(2):
        chain = [(off[k], np.float32(cs_w[m])) for k in range(n_reads) for m in range(first[k], first[k + 1])
                 if cs_set[m] == 2 * c + 1 and cs_row[m] == 0]
        body = re.search(r'VS_ODD_OFF\[1\]\[2\]\[\d+\] = \{\{(.*?)\},\},\};', src).group(1)
        offs_c = [int(v) for v in body.split('},{')[c].strip('{},').split(',') if v][:len(chain)]
        assert offs_c == [o for o, _ in chain]
    # the ring holds one tile plus the halo plus the slot being refilled
    T = 2 * nt * rr
    assert (_macro(src, 'VS_PP') - 1) * T >= T + int(idx.max()) + 1
    assert _macro(src, 'VS_PP') % _macro(src, 'VS_DD') == 0


def test_rows_share_reads_on_a_dense_table(native, golden):
    """128 taps per channel: taps whose offsets differ by a multiple of the row stride share their reads."""
    fir = golden.fir('g48k_k128_u')
    offs, idx, w = _table(fir)
    src = native.spec_kernel_source(offs, idx, w)
    n_reads = int(re.search(r'VS_RD_N\[1\] = \{(\d+),', src).group(1))
    assert n_reads < 0.9 * len(idx) * _macro(src, 'VS_RR')


def test_scope_checks(native):
    offs = np.array([0, 1, 2, 3], np.int32)
    with pytest.raises(ValueError):            # odd channel count: channel pairs share a workgroup
        native.spec_kernel_source(offs, np.array([1, 2, 3], np.int32), np.ones(3, np.float32))
    with pytest.raises(native.NativeError):    # a halo far beyond LDS
        native.spec_kernel_source(offs[:3], np.array([1, 1 << 20], np.int32), np.ones(2, np.float32))
    with pytest.raises(native.NativeError):
        native.spec_kernel_source(offs[:3], np.array([1, 2], np.int32), np.array([1.0, np.inf], np.float32))


@pytest.mark.parametrize('gname', ['g48k_k30', 'four_channels'])
def test_source_compiles_for_gfx950_with_the_intended_isa(native, golden, tmp_path, gname):
    if gname == 'four_channels':                      # two channel pairs: the dispatch over pairs compiles too
        a, b = golden.fir('g48k_k30'), golden.fir('g44k_k30')
        fir = np.zeros((max(len(a), len(b)), 4), np.float32)
        fir[:len(a), :2], fir[:len(b), 2:] = a, b
    else:
        fir = golden.fir(gname)
    offs, idx, w = _table(fir)
    src = native.spec_kernel_source(offs, idx, w)
    f = tmp_path / 'k.hip'
    f.write_text(src)
    out = tmp_path / 'k.s'
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only',
                        '-include', 'hip/hip_runtime.h', '-S', str(f), '-o', str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    assert re.search(r'ScratchSize: 0\b', asm), 'the specialised kernel must not spill'
    # (a wider signal's lane-quad exchange before the stores costs 8 registers; its ring - a tile plus the halo per channel
    #  pair - keeps a CU at 3 waves per SIMD or fewer anyway: 170 registers each)
    assert int(re.search(r'NumVgprs: (\d+)', asm).group(1)) <= (128 if fir.shape[1] == 2 else 144)
    if fir.shape[1] > 2:
        assert '#define VS_QUAD_STORES 1' in src and len(re.findall(r'quad_perm:\[0,0,1,1\]', asm)) >= 4 * _macro(src, 'VS_RR')
    ops = re.findall(r'^\s+([a-z0-9_]+)', asm, re.M)
    count = {o: ops.count(o) for o in set(ops)}
    assert count.get('flat_load_dwordx2', 0) == 0, 'LDS reads fell back to flat loads'
    groups, dd, rr = fir.shape[1] // 2, _macro(src, 'VS_DD'), _macro(src, 'VS_RR')
    taps = len(idx)
    reads = sum(int(v) for v in re.search(r'VS_RD_N\[\d+\] = \{([^}]*)\}', src).group(1).strip(',').split(','))
    # one packed FMA per (tap, row) and one aligned ds_read_b64 per unique read; ONE tile body per prefetch
    # buffer (the ring's slot phase is a run-time base register, not an unrolled copy of the code)
    # (+ the span-end chain over the odd taps, which hipcc may pack across the two channels)
    assert taps * rr * dd <= count['v_pk_fma_f32'] <= taps * rr * dd + taps
    assert reads * dd <= count['ds_read_b64'] <= reads * dd + 8 * groups * dd
    # no tap read fused into the half-rate two-address forms (the few ds_read2 left are the
    # wave-boundary exchange of the merge)
    assert count.get('ds_read2st64_b64', 0) == 0 and count.get('ds_read2_b64', 0) <= rr * 4 * groups
    assert count['s_barrier'] == groups * (dd + 1) + 1       # one per tile body, the prologue's, the one between units
    # offsets are immediates: no per-tap address arithmetic
    assert count.get('v_add_u32_e32', 0) < 40 * groups


def test_spill_check_reads_the_code_object_itself(native, tmp_path):
    """The window form rejects builds that spill registers (private memory per lane > 0).  The number is read from the
    kernel descriptor inside the ELF - hipFuncGetAttribute reports it too, but not under every tool that wraps the
    runtime (under rocprofv3 a spilling build was let through) - so the check can run here, without a device."""
    src = tmp_path / 'k.hip'
    src.write_text('''
extern "C" __global__ void clean_kernel(float *y) { y[threadIdx.x] = 1.0f; }
extern "C" __global__ void spilling_kernel(float *y, const int *idx)
{
    float a[600];
    for (int i = 0; i < 600; ++i) a[i] = y[i] * 2.0f;
    float s = 0.f;
    for (int i = 0; i < 600; ++i) s += a[idx[i] % 600];
    y[threadIdx.x] = s;
}
''')
    out = tmp_path / 'k.co'
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '--cuda-device-only', '--no-gpu-bundle-output', '-include',
                        'hip/hip_runtime.h', '-c', str(src), '-o', str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    image = out.read_bytes()
    assert image[:4] == b'\x7fELF'
    assert native.code_object_private_bytes(image, 'clean_kernel') == 0
    assert native.code_object_private_bytes(image, 'spilling_kernel') >= 2400          # the 600-float array lives in scratch
    assert native.code_object_private_bytes(image, 'no_such_kernel') == -1
    assert native.code_object_private_bytes(image[:200], 'clean_kernel') == -1          # a truncated image: no answer, no crash
    assert native.code_object_private_bytes(b'not an object at all' * 10, 'clean_kernel') == -1


def test_tuning_variables_exist_only_in_a_tuning_session(tmp_path):
    """A process started WITHOUT VND_TUNING never looks at the tuning variables (geometry overrides, A/B switches): the generated
    kernel is the default one whatever the environment says; with VND_TUNING=1 they are read live (csrc/vnd_spec.hpp: spec_env,
    kTuningNames).  Host switches are not part of this."""
    import json
    import os
    import subprocess
    import sys
    code = ("import sys, json; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from vndecorrelate_amd import _native\n"
            "offs = np.array([0, 3, 6], np.int32); idx = np.array([0, 5, 100, 2, 7, 90], np.int32); w = np.ones(6, np.float32)\n"
            "src = _native.window_kernel_source(offs, idx, w, 2, 32, 256)\n"
            "import re; print(json.dumps({k: int(re.search(r'#define %%s (\\d+)' %% k, src).group(1)) for k in ('VW_G', 'VW_PRIO')}))\n") % str(REPO)
    out = {}
    for tuning in ('', '1'):
        env = {k: v for k, v in os.environ.items() if not k.startswith('VND_')}
        env.update({'VND_WIN_G': '4', 'VND_WIN_PRIO': '3'})
        if tuning:
            env['VND_TUNING'] = tuning
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tuning] = json.loads(r.stdout.strip().splitlines()[-1])
    assert out[''] == {'VW_G': 8, 'VW_PRIO': 1}, out
    assert out['1'] == {'VW_G': 4, 'VW_PRIO': 3}, out


def test_an_unregistered_tuning_name_is_an_error_code(monkeypatch):
    """The library reads tuning variables only under VND_TUNING=1 (this tier sets it) and only names of its registry; one that is
    not there used to abort() the host process in debug builds.  Now: VND_ERR_INVALID with the name in vnd_last_error, the
    process lives, and the next call is clean (vnd_tuning_read is the planner's own read, exposed in the internal header)."""
    import ctypes
    from vndecorrelate_amd import _native
    lib = _native.load_library()
    value = ctypes.c_int32(-1)
    monkeypatch.setenv('VND_WIN_PACE', '0')
    assert lib.vnd_tuning_read(b'VND_WIN_PACE', 1, ctypes.byref(value)) == 0 and value.value == 0          # registered: read live
    monkeypatch.delenv('VND_WIN_PACE')
    assert lib.vnd_tuning_read(b'VND_WIN_PACE', 1, ctypes.byref(value)) == 0 and value.value == 1
    monkeypatch.setenv('VND_NOT_A_TUNING_NAME', '7')
    rc = lib.vnd_tuning_read(b'VND_NOT_A_TUNING_NAME', 3, ctypes.byref(value))
    assert rc == 1 and value.value == 3                                                                  # VND_ERR_INVALID, the fallback
    assert b'VND_NOT_A_TUNING_NAME' in lib.vnd_last_error() and b'kTuningNames' in lib.vnd_last_error()
    assert lib.vnd_tuning_read(b'VND_WIN_PACE', 1, ctypes.byref(value)) == 0                              # nothing sticks
