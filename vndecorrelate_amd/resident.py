"""Device-resident execution of a ``SignalChain``.

The reference runs a chain stage by stage on NumPy arrays (decorrelation.py:137-144); the
drop-in ``SignalChain`` does the same, so every GPU stage pays an upload and a download.
Here the signal is uploaded once, each stage that has a device form works on HBM buffers
through the ``*_dev`` entry points of the C ABI, and the result comes back once:

* ``VelvetNoise``  -> ``vnd_decorrelate[_fanout]_f32_dev`` (convolution + epilogue; float32)
* ``HaasEffect``   -> ``vnd_haas_f64_dev`` (float64 ``(n + delay, 2)``, bit-identical)
* anything else (``WhiteNoise``, ``stateless`` callables, custom normalisers) runs on the
  host exactly as in the plain chain, with one download/upload around it.

torch is used for what it is here for: device buffers and the current stream.
"""
from __future__ import annotations

from typing import Sequence

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


def _velvet_on_device(stage, buf, torch, dec):
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
    y = torch.empty((n, stage.num_outs), dtype=torch.float32, device=x.device)
    ws_bytes = _native.decorrelate_workspace_bytes(1, n, stage.num_outs)
    work = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    stage._device_table().decorrelate_device(
        x.data_ptr(), y.data_ptr(), 1, n, channels, mode=dec._default_mode, ms_encode=stage.mode == LayoutMode.MS,
        width=stage.width, normalize=dec._normalize_flag(stage.normalizer is not None), workspace_ptr=work.data_ptr(),
        workspace_bytes=ws_bytes, stream=torch.cuda.current_stream(x.device).cuda_stream)
    return y


def _haas_on_device(stage, buf, torch):
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
    y = torch.empty((n + delay, 2), dtype=torch.float64, device=x.device)
    if n + delay:
        _native.haas_device(_native.default_context(), x.data_ptr(), y.data_ptr(), 1, n, channels, delay=delay,
                            delayed_channel=stage.delayed_channel, ms_mode=stage.mode == LayoutMode.MS,
                            width=stage.width, stream=torch.cuda.current_stream(x.device).cuda_stream)
    return y


def run(stages: Sequence, input_signal: np.ndarray) -> np.ndarray:
    """Feed ``input_signal`` through ``stages`` keeping the signal on the device between stages."""
    from . import decorrelation as dec
    torch = _torch()
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
        return buf

    def to_host():
        nonlocal buf, host
        if host is None:
            host = buf.cpu().numpy()
            buf = None
        return host

    for stage in stages:
        out = None
        if isinstance(stage, dec.VelvetNoise):
            frames = host.shape[0] if buf is None else buf.shape[0]
            if dec._use_device_epilogue(stage.num_outs, stage.normalizer is not None, frames, not odd_layout):
                out = _velvet_on_device(stage, to_device(), torch, dec)
        elif isinstance(stage, dec.HaasEffect):
            out = _haas_on_device(stage, to_device(), torch)
        if out is not None:
            buf = out
            odd_layout = False                             # device stages produce C-ordered arrays
        else:
            host = np.asarray(stage(to_host()))
            buf = None
            odd_layout = host.ndim == 2 and not host.flags.c_contiguous
    return to_host()
