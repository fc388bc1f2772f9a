"""One seed of the randomised GPU-vs-oracle sweep (tools/sweep_parity.py) inside the driver-visible
suite: fused stack over edge geometries N = 1..128, every De / d, gated or not, training or not;
the inner op with every attribute mix; the channel FFN over ragged row counts.  Bounded subsets so
the whole file stays well under a minute on the box."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))

pytestmark = pytest.mark.gpu


def test_sweep_default_seed(gpu, egt_lib, capsys):
    import sweep_parity as SP
    rc = SP.main(seed=2024, n_attn=40, n_ffn=12)
    out = capsys.readouterr().out
    assert rc == 0, out[-4000:]
    assert "FAIL" not in out


def test_sweep_second_seed_subset(gpu, egt_lib, capsys):
    import sweep_parity as SP
    rc = SP.main(seed=7, n_stack=24, n_attn=20, n_ffn=6)
    out = capsys.readouterr().out
    assert rc == 0, out[-4000:]


def test_mfma_inner_op_sweep(gpu, egt_lib, capsys):
    """the producer / consumer MFMA kernels (d in {16,32,64}) over ragged N, odd tile counts and every feature mix they cover"""
    import sweep_mfma as SM
    bad = SM.main(seed=5, count=40)
    out = capsys.readouterr().out
    assert bad == 0 and "FAIL" not in out, out[-4000:]
