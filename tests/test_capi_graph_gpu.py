"""The C-ABI from a host without torch or Python: tests/c/graph_replay.cpp captures egt_seed_advance + egt_stack_fwd +
egt_stack_bwd into a hipGraph (device-resident random-mask seed, EGT_BF_SEED_DEVICE) and checks every replay BITWISE
against eager calls that pass the same seeds as host arguments."""
import subprocess

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,De", [(48, 64), (37, 48), (40, 8)])
def test_c_host_graph_replay_is_bit_identical(N, De, gpu, egt_lib):
    from egt_amd import build
    exe = build.build_c_host()
    r = subprocess.run([exe, str(N), str(De)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "3 hipGraph replays" in r.stdout
