"""CPU: the fp32 -> OCP e4m3 conversion the quantiser kernels use (csrc/bya_common.h, f32_to_e4m3: integer round-to-nearest-
even, the 2^14 trick for subnormals), restated in numpy, against torch's float8_e4m3fn converter -- the definition
include/bya.h points to.  (The kernel itself is compared byte for byte on the GPU: tests/test_fp8_gpu.py.)"""
import numpy as np
import torch


def f32_to_e4m3(v):
    u = v.astype(np.float32).view(np.uint32)
    sign = (u >> 24) & np.uint32(0x80)
    a = np.minimum(u & np.uint32(0x7fffffff), np.uint32(0x43e00000))                 # |v| <= 448
    normal = ((a + np.uint32(0x7ffff) + ((a >> 20) & np.uint32(1))) >> 20) - np.uint32(120 << 3)
    sub = (a.view(np.float32) + np.float32(16384.0)).view(np.uint32) - np.uint32(0x46800000)
    return (np.where(a >= np.uint32(0x3c800000), normal, sub) | sign).astype(np.uint8)


def test_integer_conversion_equals_torch_float8_e4m3fn():
    g = torch.Generator().manual_seed(0)
    edge = torch.tensor([448., -448., 447.99, 464.0 - 1e-3, 303.99997, 304.0, 30.999998, 0., -0., 2.0 ** -6, 2.0 ** -9,
                         2.0 ** -10, 1.5 * 2.0 ** -10, 0.99 * 2.0 ** -10, -2.0 ** -10, 0.0156, 0.0155, 1e-8, -1e-8])
    v = torch.cat([torch.randn(1_000_000, generator=g) * 100, torch.randn(300_000, generator=g) * 0.01,
                   torch.randn(200_000, generator=g), edge]).clamp(-448.0, 448.0).float()
    ref = v.to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = f32_to_e4m3(v.numpy())
    assert (got == ref).all(), int((got != ref).sum())
    # every one of the 254 finite e4m3 values is a fixed point
    codes = np.array([c for c in range(256) if (c & 0x7f) != 0x7f], dtype=np.uint8)
    vals = torch.from_numpy(codes).view(torch.float8_e4m3fn).float().numpy()
    assert (f32_to_e4m3(vals) == codes).all()


def test_row_scale_definition():
    """scale = max/448 with the all-zero row mapped to 1, 448/max correctly rounded (tensor / tensor in torch: `448.0 / t` is
    evaluated as reciprocal times 448 and is one ulp off for a third of the rows)."""
    g = torch.Generator().manual_seed(1)
    amax = (torch.randn(100_000, generator=g).abs() * 7 + 1e-3).float()
    exact = (448.0 / amax.double()).float()                                       # the correctly rounded quotient
    assert torch.equal(torch.full_like(amax, 448.0) / amax, exact)
    off = (448.0 / amax != exact).float().mean().item()
    print(f"scalar / tensor differs from the correctly rounded quotient on {100 * off:.1f} % of rows")
