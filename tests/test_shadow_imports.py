"""The drop-in boundary: with <repo>/src ahead of the reference's src/ on sys.path, the imports main.py performs
(src/main.py:21-22, src/rvc.py:8-15, src/vc_infer_pipeline.py:324) resolve to this implementation with the reference's
signatures.  Runs in a child interpreter so the shadow names do not leak into the test session."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import inspect, sys
sys.path.insert(0, %r)
from mdx import run_mdx, MDX, MDXModel
from rvc import Config, load_hubert, get_vc, rvc_infer
from vc_infer_pipeline import VC, Pipeline
from rmvpe import RMVPE
from my_utils import load_audio
from infer_pack.models import (SynthesizerTrnMs256NSFsid, SynthesizerTrnMs256NSFsid_nono, SynthesizerTrnMs768NSFsid,
                               SynthesizerTrnMs768NSFsid_nono)
import aicovergen_amd.mdx, aicovergen_amd.rvc, aicovergen_amd.vc_infer_pipeline
assert run_mdx is aicovergen_amd.mdx.run_mdx and VC is aicovergen_amd.vc_infer_pipeline.VC and Pipeline is VC
assert rvc_infer is aicovergen_amd.rvc.rvc_infer
sig = lambda f: list(inspect.signature(f).parameters)
assert sig(run_mdx) == ["model_params", "output_dir", "model_path", "filename", "exclude_main", "exclude_inversion", "suffix",
                        "invert_suffix", "denoise", "keep_orig", "m_threads"], sig(run_mdx)
assert sig(rvc_infer) == ["index_path", "index_rate", "input_path", "output_path", "pitch_change", "f0_method", "cpt", "version",
                          "net_g", "filter_radius", "tgt_sr", "rms_mix_rate", "protect", "crepe_hop_length", "vc",
                          "hubert_model"], sig(rvc_infer)
assert sig(VC.pipeline)[:20] == ["self", "model", "net_g", "sid", "audio", "input_audio_path", "times", "f0_up_key", "f0_method",
                                 "file_index", "index_rate", "if_f0", "filter_radius", "tgt_sr", "resample_sr", "rms_mix_rate",
                                 "version", "protect", "crepe_hop_length", "f0_file"], sig(VC.pipeline)
assert sig(Config.__init__) == ["self", "device", "is_half"] and sig(load_hubert) == ["device", "is_half", "model_path"]
assert sig(get_vc) == ["device", "is_half", "config", "model_path"]
assert sig(MDXModel.__init__)[:5] == ["self", "device", "dim_f", "dim_t", "n_fft"]
assert sig(RMVPE.__init__)[:4] == ["self", "model_path", "is_half", "device"] and sig(load_audio) == ["file", "sr"]
print("shadow imports ok")
"""


def test_imports_through_src_shadows():
    r = subprocess.run([sys.executable, "-c", CHILD % os.path.join(ROOT, "src")], capture_output=True, text=True, cwd="/")
    assert r.returncode == 0 and "shadow imports ok" in r.stdout, r.stdout + r.stderr


def test_launcher_runs_a_main_script_with_shadows_first(tmp_path):
    """src/run_main.py executes the given main.py with <repo>/src ahead of the script's own directory."""
    (tmp_path / "mdx.py").write_text("run_mdx = 'the reference module the shadow must win over'\n")
    (tmp_path / "main.py").write_text(
        "import sys\nfrom mdx import run_mdx\nimport aicovergen_amd.mdx as m\nassert run_mdx is m.run_mdx\n"
        "print('argv', sys.argv[1:])\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "src", "run_main.py"), str(tmp_path / "main.py"), "-i", "song.wav"],
                       capture_output=True, text=True, cwd="/")
    assert r.returncode == 0 and "argv ['-i', 'song.wav']" in r.stdout, r.stdout + r.stderr
