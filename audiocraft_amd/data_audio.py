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


def normalize_loudness(wav: torch.Tensor, sample_rate: int, loudness_headroom_db: float = 14,
                       loudness_compressor: bool = False, energy_floor: float = 2e-3) -> torch.Tensor:
    """reference audio_utils.py:62-88"""
    energy = wav.pow(2).mean().sqrt().item()
    if energy < energy_floor:
        return wav
    input_loudness_db = loudness(wav, sample_rate)
    gain = 10.0 ** ((-loudness_headroom_db - input_loudness_db) / 20.0)
    output = gain * wav
    if loudness_compressor:
        output = torch.tanh(output)
    assert output.isfinite().all(), (input_loudness_db, energy)
    return output


def _clip_wav(wav: torch.Tensor, log_clipping: bool = False, stem_name: tp.Optional[str] = None) -> None:
    max_scale = wav.abs().max()
    if log_clipping and max_scale > 1:
        clamp_prob = (wav.abs() > 1).float().mean().item()
        print(f"CLIPPING {stem_name or ''} happening with proba (a bit of clipping is okay):", clamp_prob,
              "maximum scale: ", max_scale.item(), file=sys.stderr)
    wav.clamp_(-1, 1)


def normalize_audio(wav: torch.Tensor, normalize: bool = True, strategy: str = 'peak', peak_clip_headroom_db: float = 1,
                    rms_headroom_db: float = 18, loudness_headroom_db: float = 14, loudness_compressor: bool = False,
                    log_clipping: bool = False, sample_rate: tp.Optional[int] = None,
                    stem_name: tp.Optional[str] = None) -> torch.Tensor:
    """reference audio_utils.py:104-152: 'peak' | 'clip' | 'rms' | 'loudness' | '' / 'none'."""
    scale_peak = 10 ** (-peak_clip_headroom_db / 20)
    scale_rms = 10 ** (-rms_headroom_db / 20)
    if strategy == 'peak':
        rescaling = (scale_peak / wav.abs().max())
        if normalize or rescaling < 1:
            wav = wav * rescaling
    elif strategy == 'clip':
        wav = wav.clamp(-scale_peak, scale_peak)
    elif strategy == 'rms':
        mono = wav.mean(dim=0)
        rescaling = scale_rms / mono.pow(2).mean().sqrt()
        if normalize or rescaling < 1:
            wav = wav * rescaling
        _clip_wav(wav, log_clipping=log_clipping, stem_name=stem_name)
    elif strategy == 'loudness':
        assert sample_rate is not None, "Loudness normalization requires sample rate."
        wav = normalize_loudness(wav, sample_rate, loudness_headroom_db, loudness_compressor)
        _clip_wav(wav, log_clipping=log_clipping, stem_name=stem_name)
    else:
        assert wav.abs().max() < 1
        assert strategy == '' or strategy == 'none', f"Unexpected strategy: '{strategy}'"
    return wav


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
