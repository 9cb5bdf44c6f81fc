"""tools/timeline.py on the committed kernel trace of the timed step (profiles/r05_kernel_trace_step.csv.gz): the numbers DESIGN 5 quotes
come out of it, and the three hardware queues are told apart by what they run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_timeline_summary_of_the_committed_trace():
    import timeline
    path = os.path.join(ROOT, "profiles", "r05_kernel_trace_step.csv.gz")
    s = timeline.summary(path)
    assert set(s["queues"].values()) == {"1", "2", "3"} and len(set(s["queues"].values())) == 3
    assert 600 < s["step_ms"] < 760 and s["device_idle_ms"] < 5.0            # the device is never idle
    mdx, hub, syn = (s["phases"][k] for k in ("mdx", "hubert_f0", "synth"))
    assert list(mdx["queues"]) == [s["queues"]["main"]]                        # MDX: one queue, back to back
    assert abs(mdx["kernel_ms_sum"] - (mdx["to_ms"] - mdx["from_ms"])) < 2.0
    assert set(hub["queues"]) == {s["queues"]["main"], s["queues"]["f0"]}
    assert len(s["f0_recurrence_launches_ms"]) == 8                            # the progressive schedule's eight segments
    f0q = hub["queues"][s["queues"]["f0"]]
    assert f0q["busy_ms"] > 0.97 * (f0q["last_ms"] - f0q["first_ms"])         # the f0 chain runs without a gap ...
    assert abs(f0q["last_ms"] - s["hubert_last_attention_ends_ms"]) < 6.0      # ... and ends with HuBERT
    assert len(s["encoder_half_bursts"]) == 4 and s["encoder_half_bursts"][0]["kernels"] == 136
    assert s["main_stream_idle_behind_hubert_ms"] < 10.0                       # only the first chunk's encoder half is exposed
    before = timeline.summary(os.path.join(ROOT, "profiles", "r05_kernel_trace_step_before.csv.gz"))
    assert before["main_stream_idle_behind_hubert_ms"] > 20.0                  # the host order this round replaced
    committed = json.load(open(os.path.join(ROOT, "profiles", "r05_timeline.json")))
    assert committed["step_ms"] == s["step_ms"] and committed["encoder_half_bursts"] == s["encoder_half_bursts"]
