"""Shared helpers for the parity tests."""
import torch


def bf16_ord(t: torch.Tensor) -> torch.Tensor:
    """Map bf16 values to integers that are monotonic in the value (for ulp distances)."""
    i = t.contiguous().view(torch.int16).to(torch.int32)
    return torch.where(i < 0, -(i & 0x7FFF), i)


def ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return (bf16_ord(a.cpu()) - bf16_ord(b.cpu())).abs()


def f8_ord(t: torch.Tensor) -> torch.Tensor:
    i = t.contiguous().view(torch.uint8).to(torch.int32)
    return torch.where(i >= 128, -(i & 0x7F), i)


def f8_ulp_diff(a, b):
    return (f8_ord(a.cpu()) - f8_ord(b.cpu())).abs()


def assert_bf16_close(got, ref, max_ulp=1, min_exact=0.99, what=""):
    d = ulp_diff(got, ref)
    exact = (d == 0).float().mean().item()
    assert d.max().item() <= max_ulp and exact >= min_exact, (
        f"{what}: max ulp diff {d.max().item()} (allowed {max_ulp}), exact fraction {exact:.5f} (need {min_exact})"
    )
    return exact


def assert_f8_close(got, ref, max_ulp=1, min_exact=0.99, what=""):
    d = f8_ulp_diff(got, ref)
    exact = (d == 0).float().mean().item()
    assert d.max().item() <= max_ulp and exact >= min_exact, (
        f"{what}: max fp8 ulp diff {d.max().item()} (allowed {max_ulp}), exact fraction {exact:.5f} (need {min_exact})"
    )
    return exact


def round_fp64_to_bf16(x: torch.Tensor) -> torch.Tensor:
    """fp64 -> bf16 with a single rounding (avoids the fp64->fp32->bf16 double rounding)."""
    f = x.to(torch.float32)
    # fix double rounding: if the fp32 value sits exactly on a bf16 tie, nudge towards the fp64 value
    bits = f.view(torch.int32)
    tie = (bits & 0xFFFF) == 0x8000
    nudge = torch.where(x.abs() > f.double().abs(), 1, torch.where(x.abs() < f.double().abs(), -1, 0)).to(torch.int32)
    bits = torch.where(tie, bits + nudge, bits)
    return bits.view(torch.float32).to(torch.bfloat16)
