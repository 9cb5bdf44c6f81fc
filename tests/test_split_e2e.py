"""End-to-end parity of the opt-in split precision (AICG_PRECISION=bf16x3): the waveform parity tests of the default fp32 path,
re-run with every layer packed in split precision -- same inputs, same golden files / oracle, same waveform bars (relative RMS
<= 1e-3 on the int16 output for C1, <= 1e-3 on the 66 s synthesizer chunk, <= 1e-4 on a full-size MDX window).  The f0
estimators and the retrieval search stay on the fp32 kernels in this mode (ops.fp32_layers), so the f0 / coarse-bin checks inside
those tests are the fp32 ones unchanged.  What split precision does NOT meet is the <= 1 LSB on >= 99.9 % of the int16 samples
bar of the small golden (tests/test_pipeline.py): that is why fp32 stays the default and the headline."""
import os

import pytest
import torch

from aicovergen_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _split():
    import conftest
    conftest._bind("hip")
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    old = ops.split_precision
    ops.split_precision = True
    yield
    ops.split_precision = old
    torch.cuda.synchronize()


def test_c1_pipeline_split_vs_reference_golden():
    import test_bench_sizes as t
    t.test_c1_pipeline_vs_reference_golden()


def test_66s_chunk_split_vs_oracle():
    import test_bench_sizes as t
    t.test_chunk_hubert_and_synth_vs_oracle(1056160)


def test_mdx_window_split_vs_oracle():
    import test_mdx as t
    t.test_voc_ft_sized_window_matches_oracle()


def test_mdx_16_window_batch_split_vs_oracle():
    import test_bench_sizes as t
    t.test_mdx_16_window_batch_vs_oracle_and_window_counts()


def test_layers_really_split():
    w = torch.randn(64, 64, 3)
    assert ops.PackedConv(w, None, device="cuda:0").split
    with ops.fp32_layers():
        assert not ops.PackedConv(w, None, device="cuda:0").split
