"""CPU tier: host-side robustness (no GPU compute): the build's staleness check sees every
source of the translation unit, and the FIR table cache is safe under concurrent callers."""
import os
import pathlib
import threading
import time

import numpy as np

REPO = pathlib.Path(__file__).resolve().parents[1]


def test_build_watches_every_native_source(tmp_path, monkeypatch):
    """An edit to ANY header under csrc/ or include/ must rebuild the git-ignored .so
    (it ships to the GPU box as built)."""
    import __graft_entry__ as entry
    ran = []
    monkeypatch.setattr(entry.subprocess, 'run', lambda cmd, **kw: ran.append([str(c) for c in cmd]))
    fake = tmp_path / 'libvnd_amd.so'
    fake.write_bytes(b'')
    monkeypatch.setattr(entry, 'LIB', fake)
    sources = sorted((REPO / 'vndecorrelate_amd' / 'csrc').glob('*.h*')) + sorted((REPO / 'vndecorrelate_amd' / 'csrc').glob('*.inc')) + sorted((REPO / 'include').glob('*.h'))
    assert len(sources) >= 6
    newest = max(s.stat().st_mtime for s in sources)
    for src in sources:
        # library newer than everything except `src`
        os.utime(fake, (newest + 10, newest + 10))
        old = src.stat()
        try:
            os.utime(src, (newest + 20, newest + 20))
            ran.clear()
            try:
                entry.build()
            except Exception:
                pass                       # the import after the (mocked) compile may fail; only the decision matters
            assert any('hipcc' in c[0] for c in ran), f'an edit to {src.name} did not trigger a rebuild'
        finally:
            os.utime(src, (old.st_atime, old.st_mtime))
    os.utime(fake, (newest + 30, newest + 30))
    ran.clear()
    try:
        entry.build()
    except Exception:
        pass
    assert not any('hipcc' in c[0] for c in ran), 'an up-to-date library was rebuilt'


class _FakeTable:
    closed = 0

    def close(self):
        type(self).closed += 1


def test_table_cache_is_thread_safe(monkeypatch):
    """Concurrent lookups build each key at most a few times, always hand every thread a live
    table, and eviction never closes a table somebody may still be using."""
    from vndecorrelate_amd import _native
    import vndecorrelate_amd.decorrelation as d

    built = []

    def fake_create(ctx, offs, idx, w, **kw):
        time.sleep(0.001)
        t = _FakeTable()
        built.append(t)
        return t

    monkeypatch.setattr(_native.TapTable, 'create', staticmethod(fake_create))
    monkeypatch.setattr(_native, 'default_context', lambda: None)
    cache = d._TableCache(capacity=4)

    class Arr:
        tap_offsets = tap_index = tap_weight = np.zeros(1)

        def kwargs(self):
            return {}

    errors = []

    def worker(seed):
        rng = np.random.default_rng(seed)
        for _ in range(300):
            key = int(rng.integers(0, 12))
            try:
                t = cache.get(key, Arr)
                assert isinstance(t, _FakeTable)
            except Exception as e:      # noqa: BLE001
                errors.append(e)

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors
    assert len(cache._items) <= 4
    assert _FakeTable.closed == 0       # dropped, never closed under a possible user
    cache.clear()
    assert len(cache._items) == 0


def test_host_entry_points_take_the_context_lock():
    """Every *_host entry point of the C ABI serialises on the context (source-level check; the
    behaviour is exercised on the GPU by tests/test_gpu_robustness.py)."""
    import re
    # (the library's one translation unit: vnd_amd.hip and the parts it includes)
    text = '\n'.join((REPO / 'vndecorrelate_amd' / 'csrc' / f).read_text() for f in ('vnd_amd.hip', 'vnd_host.hpp', 'vnd_stage.hpp', 'vnd_rccl.hpp', 'vnd_hooks.hpp'))
    header = re.sub(r'/\*.*?\*/', '', (REPO / 'include' / 'vnd_amd.h').read_text(), flags=re.S)
    host_fns = sorted(set(re.findall(r'\b(vnd_[a-z0-9_]+_host)\s*\(', header)))
    assert len(host_fns) >= 7
    # each one either locks itself or forwards to a static helper that does
    locked_helpers = set()
    for m in re.finditer(r'static vnd_status (\w+_host)\(.*?\n\{(.*?)\n\}\n', text, re.S):
        if 'HostLock lock(ctx->host_mutex)' in m.group(2):
            locked_helpers.add(m.group(1))
    for fn in host_fns:
        body = re.search(r'vnd_status ' + fn + r'\(.*?\n\{(.*?)\n\}\n', text, re.S).group(1)
        assert 'HostLock lock(ctx->host_mutex)' in body or any(h + '(' in body for h in locked_helpers), fn
