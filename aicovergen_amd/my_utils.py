"""load_audio(file, sr) -> float32 mono at `sr` (reference src/my_utils.py:5-21 shells out to ffmpeg; when the
binary is not installed, WAV files are read directly)."""
import shutil
import subprocess

import numpy as np

from . import audio_io


def load_audio(file, sr):
    file = file.strip(" ").strip('"').strip("\n").strip('"').strip(" ")
    try:
        if shutil.which("ffmpeg"):
            cmd = ["ffmpeg", "-nostdin", "-threads", "0", "-i", file, "-f", "f32le", "-acodec", "pcm_f32le", "-ac", "1",
                   "-ar", str(sr), "-"]
            out = subprocess.run(cmd, capture_output=True, check=True).stdout
            return np.frombuffer(out, np.float32).flatten()
        x, _ = audio_io.load_wav(file, sr, mono=True)
        return np.ascontiguousarray(x, dtype=np.float32)
    except Exception as e:
        raise RuntimeError(f"Failed to load audio: {e}")
