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


def assert_close_mag(got, ref, mag=0.0, ulps=1.0, min_exact=0.99, what=""):
    """|got - ref| <= `ulps` bf16 ulps evaluated at magnitude max(|ref|, mag)  (mag: scalar or tensor).

    `mag` carries the size of the intermediates the value was computed from: results that are small only
    because of cancellation cannot be reproduced to a ulp of THEIR magnitude by any other summation order.
    A bf16 ulp at magnitude v is at most v * 2^-7."""
    g, r = got.detach().cpu().double(), ref.detach().cpu().double()
    m = mag.detach().cpu().double() if torch.is_tensor(mag) else torch.full_like(r, float(mag))
    tol = ulps * torch.maximum(r.abs(), m) * 2.0 ** -7
    err = (g - r).abs()
    bad = err > tol
    exact = (g == r).double().mean().item()
    assert not bad.any() and exact >= min_exact, (
        f"{what}: {int(bad.sum())} elements beyond {ulps} bf16 ulp (worst err/tol {float((err / tol.clamp_min(1e-300)).max()):.2f}), "
        f"bit-exact fraction {exact:.5f} (need {min_exact})"
    )
    return exact
