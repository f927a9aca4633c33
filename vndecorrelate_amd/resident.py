"""Device-resident execution of a ``SignalChain``.

The reference runs a chain stage by stage on NumPy arrays (decorrelation.py:137-144); the
drop-in ``SignalChain`` does the same, so every GPU stage pays an upload and a download.
Here the signal is uploaded once, each stage that has a device form works on HBM buffers
through the ``*_dev`` entry points of the C ABI, and the result comes back once:

* ``VelvetNoise``  -> ``vnd_decorrelate[_fanout]_f32_dev`` (convolution + epilogue; float32)
* ``HaasEffect``   -> ``vnd_haas_f64_dev`` (float64 ``(n + delay, 2)``, bit-identical)
* ``stateless(convolve_velvet_noise, velvet_noise_filters=fir)`` (the README's stateless stage,
  decorrelation.py:104-110, :630-660; also ``convolve_velvet_noise_batched``) -> ``vnd_convolve_f32_dev``
  on a float32 signal with a float32 filter (the operand types NumPy multiplies in float32)
* anything else (``WhiteNoise``, other ``stateless`` callables - the positional quirk form
  ``stateless(convolve_velvet_noise, fir)`` included, which makes the SIGNAL the filter, SURVEY
  Appendix B #7 -, custom normalisers, float64 operands) runs on the host exactly as in the plain
  chain, with one download/upload around it.

Output and workspace buffers are kept per chain and stage (``BufferPool``): a chain called in a loop
on signals of one length allocates nothing after its first call.  ``transfers`` counts the host
round trips of the last ``run`` (tests assert that device stages make none).

torch is used for what it is here for: device buffers and the current stream.
"""
from __future__ import annotations

from functools import partial
from typing import Optional, Sequence

import numpy as np

from . import _native
from .utils.dsp import LayoutMode, rms_normalize, to_float32


def _torch():
    try:
        import torch
    except ImportError as exc:            # pragma: no cover - the image ships torch
        raise RuntimeError('device-resident chains need torch for device buffers') from exc
    if not torch.cuda.is_available():
        raise RuntimeError('device-resident chains need a GPU (torch.cuda.is_available() is False)')
    return torch


transfers = {'to_host': 0, 'to_device': 0}      # of the last run(): downloads / uploads of the signal


class BufferPool:
    """Device buffers of one chain, keyed by (stage index, role): reused while shape and dtype stay the same."""

    def __init__(self):
        self._buffers: dict = {}

    def get(self, torch, key, shape, dtype, device):
        have = self._buffers.get(key)
        if have is None or tuple(have.shape) != tuple(shape) or have.dtype != dtype or have.device != device:
            have = self._buffers[key] = torch.empty(shape, dtype=dtype, device=device)
        return have


def _stateless_convolve_on_device(stage, buf, torch, dec, pool, key):
    """``partial(convolve_velvet_noise[_batched], velvet_noise_filters=fir[, mode=...])`` on a device tensor
    (decorrelation.py:104-110 stores the partial, :142 calls it with the signal), or None when this call has no
    device form: the float64-promoting operand types, a 1-D signal (the reference raises IndexError there),
    channel mismatches (ValueError) - the host function keeps every such behaviour."""
    if not isinstance(stage, partial) or stage.args or stage.func not in (dec.convolve_velvet_noise, dec.convolve_velvet_noise_batched):
        return None
    kw = dict(stage.keywords)
    fir = kw.pop('velvet_noise_filters', None)
    mode = kw.pop('mode', None)
    if kw or fir is None:
        return None
    fir = np.asarray(fir)
    if fir.ndim == 1:
        fir = fir[:, None]
    batched = stage.func is dec.convolve_velvet_noise_batched
    if buf.dtype != torch.float32 or fir.dtype != np.float32 or fir.ndim != 2 or buf.dim() != (3 if batched else 2):
        return None
    channels = buf.shape[-1]
    if channels == 0 or buf.shape[-2] == 0 or fir.shape[1] != channels or (batched and buf.shape[0] == 0):
        return None
    table = dec._fir_tables.get(dec._fir_key(fir, channels), lambda: dec.function_path_arrays(fir, channels))
    x = buf.contiguous()
    y = pool.get(torch, (key, 'y'), tuple(x.shape), torch.float32, x.device)
    batch = x.shape[0] if batched else 1
    table.convolve_device(x.data_ptr(), y.data_ptr(), batch, x.shape[-2], channels,
                          dec._default_mode if mode is None else mode, torch.cuda.current_stream(x.device).cuda_stream)
    return y


def _velvet_on_device(stage, buf, torch, dec, pool=None, key=None):
    """``VelvetNoise.decorrelate`` (decorrelation.py:417-442) on a device tensor, or None when this
    stage's configuration has no device form."""
    if not (stage.normalizer is None or stage.normalizer is rms_normalize):
        return None
    if buf.dim() == 1:
        if stage.num_outs != 2:
            return None
        x = buf.unsqueeze(1)                               # fan-out instead of mono_to_stereo
    elif buf.dim() == 2 and buf.shape[1] == stage.num_outs:
        x = buf
    else:
        return None
    stereo_steps = stage.mode == LayoutMode.MS or stage.width is not None
    if x.shape[0] == 0 or (stereo_steps and stage.num_outs != 2):
        return None
    x = x.to(torch.float32).contiguous()                   # to_float32 (utils/dsp.py:66-68)
    n, channels = x.shape
    pool = pool or BufferPool()
    y = pool.get(torch, (key, 'y'), (n, stage.num_outs), torch.float32, x.device)
    ws_bytes = _native.decorrelate_workspace_bytes(1, n, stage.num_outs)
    work = pool.get(torch, (key, 'work'), (ws_bytes,), torch.uint8, x.device)
    stage._device_table().decorrelate_device(
        x.data_ptr(), y.data_ptr(), 1, n, channels, mode=dec._default_mode, ms_encode=stage.mode == LayoutMode.MS,
        width=stage.width, normalize=dec._normalize_flag(stage.normalizer is not None), workspace_ptr=work.data_ptr(),
        workspace_bytes=ws_bytes, stream=torch.cuda.current_stream(x.device).cuda_stream)
    return y


def _haas_on_device(stage, buf, torch, pool=None, key=None):
    """``HaasEffect.decorrelate`` (decorrelation.py:192-230) on a device tensor, or None."""
    delay = round(stage.delay_time_seconds * stage.sample_rate_hz)
    if delay < 0 or stage.delayed_channel not in (0, 1):
        return None
    if buf.dim() == 1:
        x = buf.unsqueeze(1)
    elif buf.dim() == 2 and buf.shape[1] == 2:
        x = buf
    else:
        return None
    x = x.to(torch.float32).contiguous()
    n, channels = x.shape
    y = (pool or BufferPool()).get(torch, (key, 'y'), (n + delay, 2), torch.float64, x.device)
    if n + delay:
        _native.haas_device(_native.default_context(), x.data_ptr(), y.data_ptr(), 1, n, channels, delay=delay,
                            delayed_channel=stage.delayed_channel, ms_mode=stage.mode == LayoutMode.MS,
                            width=stage.width, stream=torch.cuda.current_stream(x.device).cuda_stream)
    return y


def run(stages: Sequence, input_signal: np.ndarray, pool: Optional[BufferPool] = None) -> np.ndarray:
    """Feed ``input_signal`` through ``stages`` keeping the signal on the device between stages.
    ``pool``: the calling chain's buffers (a stage's output stays valid until that stage runs again:
    the result handed back to the caller is a host copy)."""
    from . import decorrelation as dec
    torch = _torch()
    pool = pool or BufferPool()
    transfers['to_host'] = transfers['to_device'] = 0
    device = torch.device('cuda', _native.default_context().device)
    host = np.asarray(input_signal)
    buf = None                                             # the signal lives in exactly one of host / buf
    # NumPy's sums follow the memory layout; the device repeats the C-contiguous order only, so a
    # normalising first stage on, say, a Fortran-ordered signal stays with NumPy (as in decorrelate)
    odd_layout = host.ndim == 2 and not host.flags.c_contiguous

    def to_device():
        nonlocal buf, host
        if buf is None:
            array = host if host.dtype in (np.float32, np.float64) else to_float32(host)
            buf = torch.from_numpy(np.ascontiguousarray(array)).to(device)
            host = None
            transfers['to_device'] += 1
        return buf

    def to_host():
        nonlocal buf, host
        if host is None:
            host = buf.cpu().numpy()
            buf = None
            transfers['to_host'] += 1
        return host

    for index, stage in enumerate(stages):
        out = None
        if isinstance(stage, dec.VelvetNoise):
            frames = host.shape[0] if buf is None else buf.shape[0]
            if dec._use_device_epilogue(stage.num_outs, stage.normalizer is not None, frames, not odd_layout):
                out = _velvet_on_device(stage, to_device(), torch, dec, pool, index)
        elif isinstance(stage, dec.HaasEffect):
            out = _haas_on_device(stage, to_device(), torch, pool, index)
        elif isinstance(stage, partial) and stage.func in (dec.convolve_velvet_noise, dec.convolve_velvet_noise_batched):
            # (decided on what the signal IS: a float64 signal - Haas output - or one still on the host as int16 keeps
            #  the host function and its promoting kernel)
            current = buf if buf is not None else host
            if current.dtype in (np.float32, torch.float32) and not odd_layout:
                out = _stateless_convolve_on_device(stage, to_device(), torch, dec, pool, index)
        if out is not None:
            buf = out
            odd_layout = False                             # device stages produce C-ordered arrays
        else:
            host = np.asarray(stage(to_host()))
            buf = None
            odd_layout = host.ndim == 2 and not host.flags.c_contiguous
    return to_host()
