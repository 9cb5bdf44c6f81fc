"""WAV file plumbing either side of the hot path (the reference uses librosa.load / soundfile.write / ffmpeg,
src/mdx.py:257,273,280 and src/my_utils.py:14-16; none of them exist in this image).  scipy only; no arithmetic
beyond sample-format conversion and (if a file is not at the requested rate) polyphase resampling."""
import shutil
import subprocess

import numpy as np
from scipy.io import wavfile
from scipy.signal import resample_poly


def _to_float(data):
    if data.dtype == np.int16:
        return data.astype(np.float32) / 32768.0
    if data.dtype == np.int32:
        return data.astype(np.float32) / 2147483648.0
    if data.dtype == np.uint8:
        return (data.astype(np.float32) - 128.0) / 128.0
    return data.astype(np.float32)


def _ffmpeg_decode(path, sr, mono):
    """Any container / codec ffmpeg reads (main.py hands run_mdx the yt-dlp mp3 or a user's m4a / flac / ogg) -> float32
    (C, N) at `sr`; same command shape as the reference's my_utils.load_audio (src/my_utils.py:14-16)."""
    ch = 1 if mono else 2
    cmd = ["ffmpeg", "-nostdin", "-threads", "0", "-i", path, "-f", "f32le", "-acodec", "pcm_f32le", "-ac", str(ch),
           "-ar", str(int(sr)), "-"]
    out = subprocess.run(cmd, capture_output=True, check=True).stdout
    return np.frombuffer(out, np.float32).reshape(-1, ch).T.copy()


def load_wav(path, sr, mono):
    """-> float32 (channels, N) (or (N,) when mono) at `sr`, like librosa.load(path, mono=mono, sr=sr).
    Plain WAV files are read directly; anything else (or a WAV encoding scipy cannot parse) is decoded through ffmpeg when
    the binary is installed, otherwise a clear error names the file."""
    try:
        file_sr, data = wavfile.read(path)
    except Exception as e:  # not RIFF/WAVE, or an encoding scipy does not know (ADPCM, extensible float64 ...)
        if shutil.which("ffmpeg"):
            x = _ffmpeg_decode(path, sr, mono)
            return (x[0] if mono else x), sr
        raise RuntimeError("cannot read %r: not a WAV file scipy understands (%s) and ffmpeg is not installed to decode "
                           "other containers" % (path, e))
    x = _to_float(data)
    x = x[:, None] if x.ndim == 1 else x
    x = x.T  # (C, N)
    if mono:
        x = x.mean(axis=0, keepdims=True)
    if file_sr != sr:
        g = np.gcd(int(file_sr), int(sr))
        x = resample_poly(x, sr // g, file_sr // g, axis=1).astype(np.float32)
    return (x[0] if mono else x), sr


def write_wav_pcm16(path, data, sr):
    """soundfile.write(path, data (N, C) float, sr) default WAV subtype: 16-bit PCM."""
    y = np.clip(np.asarray(data, dtype=np.float64), -1.0, 32767.0 / 32768.0)
    wavfile.write(path, int(sr), np.rint(y * 32768.0).astype(np.int16))
