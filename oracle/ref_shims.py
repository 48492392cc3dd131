"""Import shims for running the UNMODIFIED reference (/root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py (in the build container, where
/root/reference exists) to pin the oracle and to generate tests/golden fixtures.  Nothing here is
reachable from the product path and nothing here exists on the GPU box.

Why shims are needed:
  * `loguru` is not installed            -> stub module exposing a no-op `logger`
  * `pydash` is not installed            -> stub module exposing `flatten` (one level, what flux_emphasis.py:380-383,416 uses)
  * float8_quantize.py:19-23 raises unless torch.version.cuda >= 12.4; on a ROCm wheel it is None
                                           -> set torch.version.cuda = "12.4" before importing
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


class _NullLogger:
    def __getattr__(self, name):
        return lambda *a, **k: None


def install():
    import torch

    if "loguru" not in sys.modules:
        m = types.ModuleType("loguru")
        m.logger = _NullLogger()
        sys.modules["loguru"] = m
    if "pydash" not in sys.modules:
        try:
            import pydash  # noqa: F401
        except ImportError:
            m = types.ModuleType("pydash")
            m.flatten = lambda xs: [y for x in xs for y in (x if isinstance(x, (list, tuple)) else [x])]
            sys.modules["pydash"] = m
    if not torch.version.cuda:
        torch.version.cuda = "12.4"
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def import_reference():
    """Returns (float8_quantize, flux_model, util) modules of the reference."""
    install()
    import float8_quantize  # noqa
    import modules.flux_model as flux_model  # noqa
    import util  # noqa

    return float8_quantize, flux_model, util
