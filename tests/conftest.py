import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "flux-fp8-api_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The oracle is torch on the host cores.  On a big host (the GPU box: 256 hardware threads) torch's default team of one thread per core is
# far past the knee for these shapes -- oracle/cpu_baseline.py's thread scan on that box: one SingleStreamBlock takes 0.021 s on 32 threads,
# 0.099 s on 64, 4.2 s on 256 -- and the GPU suite spent most of its 12.7 minutes there (26 minutes of SYSTEM time: spinning teams).
# Must be set before torch creates its OpenMP runtime; small hosts (the 8-core build container) keep torch's default.
if (os.cpu_count() or 1) >= 64:
    os.environ.setdefault("OMP_NUM_THREADS", "32")
    os.environ.setdefault("MKL_NUM_THREADS", "32")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# ---- scheduling of the full-geometry cases (tests/test_full_geometry_gpu.py, tests/oracle_prefetch.py) ---------------------------------
FULL_DEPTH_CASES = ("c2_19p38_L320", "c1_schnell_bf16_19p38_L512")  # share one 11.9 B-parameter checkpoint: ~3 minutes on ONE core


def _full_geometry_case(item):
    if "test_full_geometry_gpu.py" not in item.nodeid:
        return None
    if item.name.startswith("test_full_depth_19_38"):
        return "c2_19p38_L320"
    cs = getattr(item, "callspec", None)
    name = cs.params.get("name") if cs is not None else None
    return name if isinstance(name, str) and not name.startswith("tiny_") else None


def pytest_collection_modifyitems(config, items):
    """the two full-depth cases run LAST: their checkpoint is synthesised by a background thread while everything else runs"""
    late = [it for it in items if _full_geometry_case(it) in FULL_DEPTH_CASES]
    if late:
        items[:] = [it for it in items if it not in late] + late


def pytest_collection_finish(session):
    """after -m / -k deselection: start the host-side work of every full-geometry case the session will run (GPU sessions only)"""
    import torch

    if not torch.cuda.is_available() or os.environ.get("FLUXMI_TEST_PREFETCH", "1") == "0":
        return
    cases = []
    for it in session.items:
        c = _full_geometry_case(it)
        if c and c not in cases and it.get_closest_marker("gpu") is not None:
            cases.append(c)
    if cases:
        import oracle_prefetch

        oracle_prefetch.schedule([[c for c in cases if c in FULL_DEPTH_CASES], [c for c in cases if c not in FULL_DEPTH_CASES]])
