"""CPU tier: property tests (hypothesis) of the host logic and of the oracle's two
restatements against each other on random tables - cheap breadth beyond the fixtures."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import c_oracle
from oracle import vnd_oracle as O
from vndecorrelate_amd.distributed import shard_range
from vndecorrelate_amd.taps import TapArrays, class_path_arrays, function_path_arrays

SET = settings(max_examples=40, deadline=None, derandomize=True, database=None)


@st.composite
def sparse_fir(draw):
    channels = draw(st.integers(1, 4))
    length = draw(st.integers(1, 300))
    fir = np.zeros((length, channels), np.float32)
    for c in range(channels):
        k = draw(st.integers(0, min(length, 12)))
        idx = draw(st.lists(st.integers(0, length - 1), min_size=k, max_size=k, unique=True))
        for i in idx:
            fir[i, c] = draw(st.sampled_from([0.85, -0.85, 0.55, -0.55, 0.35, -0.2, 1.0, -1.0, 0.125]))
    return fir


@SET
@given(fir=sparse_fir(), n=st.integers(0, 700), seed=st.integers(0, 2**31 - 1))
def test_numpy_and_c_oracles_agree_function_path(fir, n, seed):
    x = np.random.default_rng(seed).uniform(-1, 1, (n, fir.shape[1])).astype(np.float32)
    want = O.convolve_velvet_noise(x, fir)
    arr = function_path_arrays(fir)
    assert np.array_equal(c_oracle.convolve(x, arr.tap_offsets, arr.tap_index, arr.tap_weight), want)
    if n:
        assert np.array_equal(O.convolve_taps_scalar(x, arr.tap_offsets, arr.tap_index, arr.tap_weight), want)
    back = TapArrays.from_bytes(arr.to_bytes())
    assert np.array_equal(back.tap_index, arr.tap_index) and np.array_equal(back.tap_weight, arr.tap_weight)
    # table invariants the kernels rely on: CSR, ascending unique indices per channel, no zero weights
    assert arr.tap_offsets[0] == 0 and arr.tap_offsets[-1] == len(arr.tap_index)
    for c in range(fir.shape[1]):
        sl = arr.tap_index[arr.tap_offsets[c]:arr.tap_offsets[c + 1]]
        assert np.all(np.diff(sl) > 0)
    assert np.all(arr.tap_weight != 0)


@st.composite
def class_table(draw):
    channels = draw(st.integers(1, 3))
    nseg = draw(st.integers(1, 4))
    env = tuple(draw(st.sampled_from([1.0, 0.85, 0.5, 0.25])) for _ in range(nseg))
    chans = []
    for _ in range(channels):
        if draw(st.booleans()) and channels > 1 and len(chans) and any(c is not None for c in chans):
            chans.append(None)
            continue
        chans.append([(draw(st.lists(st.integers(0, 200), max_size=4)), draw(st.lists(st.integers(0, 200), max_size=4)))
                      for _ in range(nseg)])
    return chans, env


@SET
@given(tab=class_table(), n=st.integers(1, 500), seed=st.integers(0, 2**31 - 1))
def test_numpy_and_c_oracles_agree_class_path(tab, n, seed):
    chans, env = tab
    x = np.random.default_rng(seed).uniform(-1, 1, (n, len(chans))).astype(np.float32)
    apply_gain = env != (1.0,)
    arr = class_path_arrays(chans, env, apply_gain)
    want = O.class_convolve(x, chans, env, len(chans))
    got = c_oracle.convolve(x, arr.tap_offsets, arr.tap_index, arr.tap_weight, seg_off=arr.seg_offsets,
                            seg_end=arr.seg_end, seg_gain=arr.seg_gain, chan_flags=arr.chan_flags,
                            apply_gain=arr.apply_gain)
    assert np.array_equal(got, want)
    back = TapArrays.from_bytes(arr.to_bytes())
    assert back.to_bytes() == arr.to_bytes()
    assert arr.seg_offsets[-1] == len(arr.seg_end) and (len(arr.seg_end) == 0 or arr.seg_end[-1] == len(arr.tap_index))


@SET
@given(total=st.integers(0, 5000), world=st.integers(1, 16))
def test_shard_range_is_a_partition(total, world):
    spans = [shard_range(total, world, r) for r in range(world)]
    covered = np.zeros(total, np.int32)
    for start, count in spans:
        covered[start:start + count] += 1
    assert np.all(covered == 1)
    assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _numpy_pairwise_model(sq: np.ndarray) -> np.float32:
    """The order in which NumPy adds a contiguous float32 vector (what rms_pairwise_kernel repeats on the
    device for single-channel tables): 8192-element chunks left to right, each summed pairwise down to
    leaves of at most 128 elements with eight strided accumulators."""
    def leaf(a):
        n = len(a)
        if n < 8:
            r = np.float32(0.0)
            for v in a:
                r = np.float32(r + v)
            return r
        m = n - n % 8
        acc = a[:8].copy()
        for row in a[8:m].reshape(-1, 8):
            acc = (acc + row).astype(np.float32)
        res = np.float32(np.float32(np.float32(acc[0] + acc[1]) + np.float32(acc[2] + acc[3])) +
                         np.float32(np.float32(acc[4] + acc[5]) + np.float32(acc[6] + acc[7])))
        for v in a[m:]:
            res = np.float32(res + v)
        return res

    def tree(a):
        n = len(a)
        if n <= 128:
            return leaf(a)
        half = n // 2
        half -= half % 8
        return np.float32(tree(a[:half]) + tree(a[half:]))

    acc = None
    for i in range(0, len(sq), 8192):
        part = tree(sq[i:i + 8192])
        acc = part if acc is None else np.float32(acc + part)
    return acc


def test_numpy_sums_a_single_channel_pairwise_in_8192_chunks():
    """The summation order the device repeats for (n, 1) signals is this box's NumPy's."""
    rng = np.random.default_rng(11)
    lengths = list(range(1, 20)) + [127, 128, 129, 135, 263, 1000, 8191, 8192, 8193, 16384, 20000, 100001] + \
        [int(v) for v in rng.integers(20, 60000, 25)]
    for n in lengths:
        sq = np.square((rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-3, 2)).astype(np.float32))
        want = np.add.reduce(sq.reshape(n, 1), axis=0)[0]
        assert _numpy_pairwise_model(sq) == want == np.add.reduce(sq), n
        assert np.float32(want / np.float32(n)) == np.mean(sq.reshape(n, 1), axis=0)[0], n
