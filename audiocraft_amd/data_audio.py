"""Output stage after the generation path -- mirrors `audiocraft.data.audio.audio_write` (reference
audiocraft/data/audio.py:159-231) and the normalisation helpers of `audiocraft.data.audio_utils` (:62-152).

Host-side code (one call per finished clip, not on the decode path).  Where the reference leans on third-party
packages the published algorithm is restated: `torchaudio.functional.loudness` (ITU-R BS.1770-4: K-weighting = a +4 dB
high shelf at 1.5 kHz and a 38 Hz high-pass, 400 ms blocks with 75 % overlap, absolute gate -70 LKFS, relative gate
-10 LU) for the 'loudness' strategy, and ffmpeg's `pcm_s16le` WAV muxing for `format='wav'` (written with the standard
library); mp3 / ogg / flac need an `ffmpeg` binary on PATH exactly like the reference."""
import math
import shutil
import subprocess as sp
import sys
import typing as tp
import wave
from pathlib import Path

import numpy as np
import torch


def _biquad(x: np.ndarray, b: tp.Sequence[float], a: tp.Sequence[float]) -> np.ndarray:
    """torchaudio.functional.biquad = lfilter(..., clamp=True): direct form, output clamped to [-1, 1]."""
    from scipy.signal import lfilter
    b = np.asarray(b, dtype=np.float64) / a[0]
    a = np.asarray(a, dtype=np.float64) / a[0]
    return np.clip(lfilter(b, a, x.astype(np.float64), axis=-1), -1.0, 1.0)


def loudness(wav: torch.Tensor, sample_rate: int) -> float:
    """Integrated loudness in LKFS of wav [C, T] (torchaudio.functional.loudness / transforms.Loudness)."""
    x = wav.detach().float().cpu().numpy()
    if x.ndim == 1:
        x = x[None]
    gate_samples = int(round(0.4 * sample_rate))
    step = int(round(gate_samples * 0.25))
    if x.shape[-1] < gate_samples:
        raise ValueError("loudness needs at least 400 ms of audio")
    # K-weighting: treble shelf (+4 dB, 1500 Hz, Q 1/sqrt(2)), then high-pass (38 Hz, Q 0.5)
    w0 = 2 * math.pi * 1500.0 / sample_rate
    alpha = math.sin(w0) / 2 / (1 / math.sqrt(2))
    A = math.exp(4.0 / 40 * math.log(10))
    t1, t2, t3 = 2 * math.sqrt(A) * alpha, (A - 1) * math.cos(w0), (A + 1) * math.cos(w0)
    x = _biquad(x, [A * ((A + 1) + t2 + t1), -2 * A * ((A - 1) + t3), A * ((A + 1) + t2 - t1)],
                [(A + 1) - t2 + t1, 2 * ((A - 1) - t3), (A + 1) - t2 - t1])
    w0 = 2 * math.pi * 38.0 / sample_rate
    alpha = math.sin(w0) / 2 / 0.5
    x = _biquad(x, [(1 + math.cos(w0)) / 2, -1 - math.cos(w0), (1 + math.cos(w0)) / 2], [1 + alpha, -2 * math.cos(w0), 1 - alpha])
    n_blocks = (x.shape[-1] - gate_samples) // step + 1
    idx = np.arange(gate_samples)[None, :] + step * np.arange(n_blocks)[:, None]
    energy = np.mean(x[:, idx] ** 2, axis=-1)                         # [C, blocks]
    g = np.array([1.0, 1.0, 1.0, 1.41, 1.41])[:energy.shape[0]][:, None]
    block_loudness = -0.691 + 10 * np.log10(np.sum(g * energy, axis=0) + 1e-300)
    gated = block_loudness > -70.0
    if not gated.any():
        return -float('inf')
    e = (energy * gated).sum(-1) / gated.sum()
    gamma_rel = -0.691 + 10 * np.log10(np.sum(g[:, 0] * e)) - 10.0
    gated = gated & (block_loudness > gamma_rel)
    e = (energy * gated).sum(-1) / max(int(gated.sum()), 1)
    return float(-0.691 + 10 * np.log10(np.sum(g[:, 0] * e)))


class _Gain(tp.NamedTuple):
    """What a normalisation strategy decides for one clip: ONE multiplier, ONE symmetric ceiling, applied in one pass."""
    gain: tp.Optional[float] = None      # None: samples are not scaled
    ceiling: tp.Optional[float] = None   # None: samples are not clamped
    soft: bool = False                   # tanh between the gain and the ceiling ('loudness' with its compressor)
    report: bool = False                 # overshoot beyond the ceiling is reported on stderr when asked for


def _db_to_amplitude(db: float) -> float:
    return 10.0 ** (db / 20.0)


def _root_mean_square(x: torch.Tensor) -> float:
    return math.sqrt(x.double().square().mean().item())


def _loudness_gain(wav: torch.Tensor, sample_rate: int, headroom_db: float, energy_floor: float) -> tp.Optional[float]:
    """Multiplier that brings `wav` to -headroom_db LKFS; None for a clip too quiet to measure (left as it is)."""
    if _root_mean_square(wav) < energy_floor:
        return None
    return _db_to_amplitude(-headroom_db - loudness(wav, sample_rate))


def _plan_peak(wav, o) -> _Gain:
    g = _db_to_amplitude(-o['peak_clip_headroom_db']) / wav.abs().max().item()
    return _Gain(gain=g if (o['normalize'] or g < 1) else None)


def _plan_clip(wav, o) -> _Gain:
    return _Gain(ceiling=_db_to_amplitude(-o['peak_clip_headroom_db']))


def _plan_rms(wav, o) -> _Gain:
    g = _db_to_amplitude(-o['rms_headroom_db']) / _root_mean_square(wav.mean(dim=0))   # level of the mono downmix
    return _Gain(gain=g if (o['normalize'] or g < 1) else None, ceiling=1.0, report=True)


def _plan_loudness(wav, o) -> _Gain:
    if o['sample_rate'] is None:
        raise AssertionError("Loudness normalization requires sample rate.")
    g = _loudness_gain(wav, o['sample_rate'], o['loudness_headroom_db'], energy_floor=2e-3)
    return _Gain(gain=g, ceiling=1.0, soft=o['loudness_compressor'] and g is not None, report=True)


def _plan_none(wav, o) -> _Gain:
    if not wav.abs().max().item() < 1:
        raise AssertionError("un-normalised audio must stay inside (-1, 1)")
    return _Gain()


_PLANS: tp.Dict[str, tp.Callable[[torch.Tensor, dict], _Gain]] = {
    'peak': _plan_peak, 'clip': _plan_clip, 'rms': _plan_rms, 'loudness': _plan_loudness, '': _plan_none, 'none': _plan_none}


def _apply_gain(wav: torch.Tensor, plan: _Gain, log_clipping: bool, stem_name: tp.Optional[str]) -> torch.Tensor:
    out = wav if plan.gain is None else wav * plan.gain
    if plan.soft:
        out = torch.tanh(out)
    if plan.ceiling is None:
        return out
    if plan.report and log_clipping:
        over = out.abs() > plan.ceiling
        if bool(over.any()):
            print(f"[normalize_audio] {stem_name or 'clip'}: {over.float().mean().item():.4%} of the samples exceed full scale "
                  f"(largest {out.abs().max().item():.4f}); they are clamped", file=sys.stderr)
    return out.clamp(-plan.ceiling, plan.ceiling)


def _clip_wav(wav: torch.Tensor, log_clipping: bool = False, stem_name: tp.Optional[str] = None) -> None:
    """In-place clamp to full scale (the reference's helper of this name, audio_utils.py:91-101, imported by its tests)."""
    wav.copy_(_apply_gain(wav, _Gain(ceiling=1.0, report=True), log_clipping, stem_name))


def normalize_loudness(wav: torch.Tensor, sample_rate: int, loudness_headroom_db: float = 14,
                       loudness_compressor: bool = False, energy_floor: float = 2e-3) -> torch.Tensor:
    """`wav` scaled to -loudness_headroom_db LKFS (BS.1770-4), optionally soft-limited by tanh; a clip whose RMS is below
    `energy_floor` is returned unchanged.  Behaviour of reference audio_utils.py:62-88 (no clamp at this level)."""
    g = _loudness_gain(wav, sample_rate, loudness_headroom_db, energy_floor)
    out = _apply_gain(wav, _Gain(gain=g, soft=loudness_compressor and g is not None), False, None)
    if not bool(out.isfinite().all()):
        raise AssertionError(f"loudness normalisation produced non-finite samples (gain {g})")
    return out


def normalize_audio(wav: torch.Tensor, normalize: bool = True, strategy: str = 'peak', peak_clip_headroom_db: float = 1,
                    rms_headroom_db: float = 18, loudness_headroom_db: float = 14, loudness_compressor: bool = False,
                    log_clipping: bool = False, sample_rate: tp.Optional[int] = None,
                    stem_name: tp.Optional[str] = None) -> torch.Tensor:
    """Level control in front of the file writer; same strategies and results as reference audio_utils.py:104-152:
    'peak' (largest sample to -peak_clip_headroom_db dBFS), 'clip' (clamp at that level), 'rms' (mono RMS to -rms_headroom_db,
    then clamp to full scale), 'loudness' (BS.1770 loudness to -loudness_headroom_db LKFS, then clamp), '' / 'none' (must
    already fit).  With normalize=False a gain is only applied when it attenuates.  Each strategy reduces to one `_Gain`."""
    if strategy not in _PLANS:
        raise AssertionError(f"Unexpected strategy: '{strategy}'")
    opts = dict(normalize=normalize, peak_clip_headroom_db=peak_clip_headroom_db, rms_headroom_db=rms_headroom_db,
                loudness_headroom_db=loudness_headroom_db, loudness_compressor=loudness_compressor, sample_rate=sample_rate)
    plan = _PLANS[strategy](wav, opts)
    out = _apply_gain(wav, plan, log_clipping, stem_name)
    if strategy == 'loudness' and not bool(out.isfinite().all()):
        raise AssertionError(f"loudness normalisation produced non-finite samples (gain {plan.gain})")
    return out


def f32_pcm(wav: torch.Tensor) -> torch.Tensor:
    """reference audio_utils.py:155-169"""
    if wav.dtype.is_floating_point:
        return wav
    if wav.dtype == torch.int16:
        return wav.float() / 2 ** 15
    if wav.dtype == torch.int32:
        return wav.float() / 2 ** 31
    raise ValueError(f"Unsupported wav dtype: {wav.dtype}")


def i16_pcm(wav: torch.Tensor) -> torch.Tensor:
    """reference audio_utils.py:172-192: float samples in [-1, 1] to int16 -- scaled by 2^15, or by 2^15 - 1 when a sample
    would otherwise land on +32768 (the int16 range is asymmetric); int16 input is returned as it is."""
    if not wav.dtype.is_floating_point:
        assert wav.dtype == torch.int16
        return wav
    assert wav.abs().max() <= 1
    scaled = (wav * 2 ** 15).round()
    if scaled.max() >= 2 ** 15:
        scaled = (wav * (2 ** 15 - 1)).round()
    return scaled.short()


def _write_wav_s16(path: Path, wav: torch.Tensor, sample_rate: int):
    """What `ffmpeg -f f32le ... -f wav -c:a pcm_s16le` produces: interleaved little-endian int16, samples rounded to
    the nearest step and saturated."""
    x = f32_pcm(wav).t().detach().cpu().numpy().astype(np.float64)
    pcm = np.clip(np.rint(x * 32768.0), -32768, 32767).astype('<i2')
    with wave.open(str(path), 'wb') as f:
        f.setnchannels(wav.shape[0])
        f.setsampwidth(2)
        f.setframerate(int(sample_rate))
        f.writeframes(pcm.tobytes())


def _piping_to_ffmpeg(out_path: Path, wav: torch.Tensor, sample_rate: int, flags: tp.List[str]):
    if shutil.which('ffmpeg') is None:
        raise RuntimeError("writing this format needs an `ffmpeg` binary on PATH (as in the reference); use format='wav'")
    command = ['ffmpeg', '-loglevel', 'error', '-y', '-f', 'f32le', '-ar', str(sample_rate), '-ac', str(wav.shape[0]),
               '-i', '-'] + flags + [str(out_path)]
    sp.run(command, input=f32_pcm(wav).t().detach().cpu().numpy().tobytes(), check=True)


def audio_write(stem_name: tp.Union[str, Path], wav: torch.Tensor, sample_rate: int, format: str = 'wav',
                mp3_rate: int = 320, ogg_rate: tp.Optional[int] = None, normalize: bool = True, strategy: str = 'peak',
                peak_clip_headroom_db: float = 1, rms_headroom_db: float = 18, loudness_headroom_db: float = 14,
                loudness_compressor: bool = False, log_clipping: bool = True, make_parent_dir: bool = True,
                add_suffix: bool = True) -> Path:
    """reference data/audio.py:159-231: normalise, then write `<stem_name>.<format>`; returns the path."""
    assert wav.dtype.is_floating_point, "wav is not floating point"
    if wav.dim() == 1:
        wav = wav[None]
    elif wav.dim() > 2:
        raise ValueError("Input wav should be at most 2 dimension.")
    assert wav.isfinite().all()
    wav = normalize_audio(wav, normalize, strategy, peak_clip_headroom_db, rms_headroom_db, loudness_headroom_db,
                          loudness_compressor, log_clipping=log_clipping, sample_rate=sample_rate, stem_name=str(stem_name))
    flags_by_format = {'mp3': ('.mp3', ['-f', 'mp3', '-c:a', 'libmp3lame', '-b:a', f'{mp3_rate}k']),
                       'wav': ('.wav', ['-f', 'wav', '-c:a', 'pcm_s16le']),
                       'ogg': ('.ogg', ['-f', 'ogg', '-c:a', 'libvorbis'] + (['-b:a', f'{ogg_rate}k'] if ogg_rate else [])),
                       'flac': ('.flac', ['-f', 'flac'])}
    if format not in flags_by_format:
        raise RuntimeError(f"Invalid format {format}. Only wav or mp3 are supported.")
    suffix, flags = flags_by_format[format]
    path = Path(str(stem_name) + (suffix if add_suffix else ''))
    if make_parent_dir:
        path.parent.mkdir(exist_ok=True, parents=True)
    try:
        if format == 'wav':
            _write_wav_s16(path, wav, sample_rate)
        else:
            _piping_to_ffmpeg(path, wav, sample_rate, flags)
    except Exception:
        if path.exists():
            path.unlink()   # do not leave half written files around
        raise
    return path
