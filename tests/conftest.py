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


# process-spawning tests: real engines in child processes joined by a process group.  They depend on the most machinery outside the
# kernels (spawn, rendezvous ports, a second HIP context on the device), so they run LAST: under `-x` nothing they do can hide a kernel result.
SPAWNING_TESTS = ("test_two_process_sharded_calibration_over_a_process_group",)
_FILE_RANK = {"test_ops_gpu.py": 0, "test_text_gpu.py": 1, "test_engine_gpu.py": 2, "test_pixels_gpu.py": 3, "test_full_geometry_gpu.py": 4}


def _gpu_rank(item):
    if item.originalname in SPAWNING_TESTS or item.name in SPAWNING_TESTS:
        return 9
    if _full_geometry_case(item) in FULL_DEPTH_CASES:
        return 8  # their checkpoint is synthesised by a background thread while everything else runs
    return _FILE_RANK.get(os.path.basename(item.nodeid.split("::")[0]), 3)


def pytest_collection_modifyitems(config, items):
    """GPU tests: op-level bit-exact tests FIRST (cheap, deterministic, most of the per-row parity evidence), then the text encoders, the
    model-level tests, the full-geometry cases (full depth late: background checkpoint), process-spawning tests LAST.  CPU tests keep
    their place.  The sort is stable, so the order inside a file is the file's."""
    gpu = [it for it in items if it.get_closest_marker("gpu") is not None]
    if gpu:
        gpu_sorted = iter(sorted(gpu, key=_gpu_rank))
        items[:] = [next(gpu_sorted) if it.get_closest_marker("gpu") is not None else it for it in items]


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


_exit_status = [None]


@pytest.hookimpl(trylast=True)
def pytest_sessionfinish(session, exitstatus):
    _exit_status[0] = int(exitstatus)
    if "oracle_prefetch" in sys.modules:
        sys.modules["oracle_prefetch"]._cancel.set()  # workers stop at their next log line while the summary is being written


@pytest.hookimpl(trylast=True)
def pytest_unconfigure(config):
    """no prefetch worker may still be inside torch when the interpreter finalises (oracle_prefetch.cancel_and_join): wait for them; if one
    is in the middle of a minutes-long torch call, leave through os._exit with pytest's own exit status (everything has been reported)."""
    op = sys.modules.get("oracle_prefetch")
    if op is None or op.cancel_and_join(20.0):
        return
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(_exit_status[0] if _exit_status[0] is not None else 1)
