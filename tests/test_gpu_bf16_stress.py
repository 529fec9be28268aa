"""The bf16-plane arithmetic (six bf16 products per fp32 product: csrc/gemm.hip BX, ffn_fused.hip, attention_bx.hip, conv_wino.hip BX) on the
operand distributions trained weights produce and seeded ones never do: heavy tails, one dominant channel, all-positive
post-ReLU rows, operands spanning 80 binades inside one dot product, exact zeros and values whose lower planes vanish.

Every case is measured against float64 in units of  2^-24 * sum_k |a_k| |b_k|  per output (the size of ONE fp32 rounding of the dot product's
absolute mass), with the f32-input MFMA form of the same kernel on the same inputs as the yardstick. What the first run of these tests found
(profiles/r06_bf16_stress.txt, all 63 cases): on heavy-tailed operands the bf16 form is BETTER than the f32 chain (0.55-0.7 x its maximum and mean:
the matrix core adds 16 products before it rounds once); with mixed signs - every product of this path has a mixed-sign weight operand - its MEAN
error is within 1.0-1.2 x the f32 chain's; where EVERY product has the same sign (post-ReLU x post-ReLU, which no layer of the path computes) or
one dot product spans 80 binades the six-product form is 1.2-1.45 x worse in the mean, 1.8 x in the attention over 4096 all-positive keys: the
three dropped products (m l, l m, l l: up to 2 x 2^-24 of a term) and the matrix core's internal accumulation leave a one-sided residue that
cancels between signs and adds up without them (probe: profiles/r05_bf16x_probe.txt, "U x U"). The bars: the MEAN error within 1.25 x the
yardstick for mixed-sign cases and within the stated factor for the one-sided ones; the MAXIMUM (an extreme of 10^5 outputs, +-40 % between
seeds) within 1.75 x; absolute caps in units. Where a kernel has no f32 form behind the C ABI (the fused feed-forward) the yardstick is named
in the test. `python tests/test_gpu_bf16_stress.py` prints the table.
Reference call sites of the products: `lightglue/lightglue.py:120-123, 144-162`, `SuperGlue/models/superglue.py:87-116`, `lightglue/superpoint.py:155-168`.
"""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

UNIT = 2.0 ** -24


def seed_of(*key):
    return zlib.crc32(repr(key).encode()) & 0x7FFFFFFF      # the same operands in every process (hash() of a str is salted)


@pytest.fixture(scope="module")
def ctx():
    from icepy4d_amd import _lib
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    c = _lib.Context(0)
    yield c
    c.close()


def dev(a):
    return torch.as_tensor(a).cuda().contiguous()


def operands(kind, rows, k, g):
    """[rows, k] fp32 operand of the named distribution."""
    x = torch.randn(rows, k, generator=g)
    if kind == "heavy_entries":          # 1 % of the entries 1000 x larger
        x = torch.where(torch.rand(rows, k, generator=g) < 0.01, x * 1e3, x)
    elif kind == "heavy_channel":        # one input channel 1000 x larger in every row
        x[:, k // 3] *= 1e3
    elif kind == "post_relu":            # all-positive rows: no cancellation, the sum is its own absolute mass
        x = F.relu(x) + 0.0
    elif kind == "wide_binades":         # magnitudes 2^-40 .. 2^40 inside one dot product
        e = torch.randint(-40, 41, (rows, k), generator=g).float()
        x = torch.sign(x) * (1 + torch.rand(rows, k, generator=g)) * torch.exp2(e)
    elif kind == "zeros_and_short":      # a third of the rows exactly zero, the rest values with 8 significant bits (middle and low plane vanish)
        x = torch.round(x * 16) / 16
        x = x.bfloat16().float()
        x[::3] = 0.0
    elif kind != "randn":
        raise ValueError(kind)
    return x.contiguous()


KINDS = ["randn", "heavy_entries", "heavy_channel", "post_relu", "wide_binades", "zeros_and_short"]


def units(out, ref, mass):
    """|out - ref| in units of 2^-24 * mass, where the mass is positive."""
    e = (out.double() - ref).abs() / (UNIT * mass.clamp_min(1e-300))
    e = e[mass > 0]
    return e.max().item(), e.mean().item()


def within(bx, f32, floor=0.25, max_ratio=1.75, mean_ratio=1.25):
    return bx[0] <= max_ratio * f32[0] + floor and bx[1] <= mean_ratio * f32[1] + floor


ONE_SIDED = {"post_relu", "wide_binades"}      # every product of a dot product has the same sign / one term carries the sum: no cancellation of the residue


@pytest.mark.parametrize("kind_a", KINDS)
@pytest.mark.parametrize("kind_w", ["randn", "heavy_entries", "post_relu"])
@pytest.mark.parametrize("m,n,k,tile", [(512, 256, 512, 0), (384, 256, 256, 1)])
def test_gemm_bf16_planes_on_hard_operands(ctx, kind_a, kind_w, m, n, k, tile):
    """`im_gemm_nt` mode 2 / 3 (bf16 planes, 64 / 128 tiles) against mode 0 / 1 (f32-input MFMA) on the same hard operands, both against fp64."""
    from icepy4d_amd._lib import ptr, stream_ptr
    g = torch.Generator().manual_seed(seed_of(kind_a, kind_w, m))
    a = operands(kind_a, m, k, g)
    w = operands(kind_w, n, k, g) / k ** 0.5
    b = torch.randn(n, generator=g)
    ref = a.double() @ w.double().t() + b.double()
    mass = a.double().abs() @ w.double().abs().t() + b.double().abs()
    da, dw, db = dev(a), dev(w), dev(b)
    res = {}
    for mode in (tile, tile | 2):
        dc = torch.full((m, n), float("nan"), device="cuda")
        ctx.call("im_gemm_nt", ptr(da), ptr(dw), ptr(db), ptr(dc), m, n, k, 1.0, mode, stream_ptr())
        torch.cuda.synchronize()
        out = dc.cpu()
        assert torch.isfinite(out).all()
        res[mode] = units(out, ref, mass)
    f32, bx = res[tile], res[tile | 2]
    print(f"gemm {kind_a:16s} x {kind_w:14s} K={k} tile={tile}: bf16 planes max {bx[0]:.2f} mean {bx[1]:.3f} | f32 MFMA max {f32[0]:.2f} mean {f32[1]:.3f}  [2^-24 sum|a||w|]")
    one_sided = kind_a == "wide_binades" or (kind_a == "post_relu" and kind_w == "post_relu")
    assert within(bx, f32, mean_ratio=1.5 if one_sided else 1.25, max_ratio=2.0 if one_sided else 1.75), (bx, f32)
    assert bx[0] < 64 and bx[1] < 4, bx   # and never far from one rounding of the mass, whatever the yardstick does
    if kind_a == "zeros_and_short":
        assert (dc.cpu()[::3] == b).all()   # zero rows give the bias exactly


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("act", [0, 1])
def test_ffn_bf16_planes_on_hard_operands(ctx, kind, act):
    """`im_ffn_fused` (both GEMMs on bf16 planes; LayerNorm + GELU or ReLU between them) with hard activations / attention messages against the same
    chain in fp64. No f32 form of this kernel sits behind the C ABI: the yardstick is the chain in plain torch fp32 on the device (fp32 GEMMs,
    fp32 LayerNorm / GELU), whose error against fp64 the kernel must not exceed by more than 1.5 x (it does better on most rows)."""
    from icepy4d_amd._lib import ptr, stream_ptr
    rows = 512
    g = torch.Generator().manual_seed(seed_of(kind, act))
    x = operands(kind, 2 * rows, 256, g).reshape(2, rows, 256)
    att = operands(kind, 2 * rows, 256, g).reshape(2, rows, 256)
    if kind == "wide_binades":            # keep LayerNorm's input finite in fp32: 2^-20 .. 2^20
        x = torch.sign(x) * x.abs().clamp(2.0 ** -20, 2.0 ** 20); att = torch.sign(att) * att.abs().clamp(2.0 ** -20, 2.0 ** 20)
    w0 = operands("heavy_entries", 512, 512, g) / 512 ** 0.5
    b0 = torch.randn(512, generator=g) * 0.1
    lg = 1 + 0.1 * torch.randn(512, generator=g)
    lb = 0.1 * torch.randn(512, generator=g)
    w3 = operands("heavy_entries", 256, 512, g) / 512 ** 0.5
    b3 = torch.randn(256, generator=g) * 0.1

    def chain(dt, device):
        c = lambda t: t.to(device=device, dtype=dt)
        h = torch.cat([c(x), c(att)], -1) @ c(w0).t() + c(b0)
        h = F.gelu(F.layer_norm(h, (512,), c(lg), c(lb), 1e-5)) if act == 0 else F.relu(h)
        return (c(x) + h @ c(w3).t() + c(b3)), h
    ref, h64 = chain(torch.float64, "cuda")
    y32, _ = chain(torch.float32, "cuda")
    # the absolute mass of the second product + the residual: what one fp32 rounding of the output is worth
    mass = (x.double().abs().cuda() + h64.abs() @ w3.double().abs().t().cuda() + b3.double().abs().cuda()).cpu()
    dx, da = dev(x), dev(att)
    hw = [t.contiguous().numpy() for t in (w0, b0, lg, lb, w3, b3)]
    if act == 1:
        hw[2] = hw[3] = None
    ctx.call("im_ffn_fused", act, ptr(dx), ptr(da), *[ptr(t) for t in hw], 2, rows, None, stream_ptr())
    torch.cuda.synchronize()
    out = dx.cpu()
    assert torch.isfinite(out).all()
    bx, y = units(out, ref.cpu(), mass), units(y32.cpu(), ref.cpu(), mass)
    print(f"ffn act={act} {kind:16s}: bf16 planes max {bx[0]:.2f} mean {bx[1]:.3f} | torch fp32 chain max {y[0]:.2f} mean {y[1]:.3f}  [2^-24 mass]")
    assert bx[0] <= 1.5 * y[0] + 1.0 and bx[1] <= 1.5 * y[1] + 0.25, (bx, y)


@pytest.mark.parametrize("kind", ["randn", "heavy_entries", "heavy_channel", "post_relu", "zeros_and_short"])
@pytest.mark.parametrize("n", [512, 4096])
def test_flash_attn_bf16_planes_on_hard_operands(ctx, kind, n):
    """`im_flash_attn` forms 0 and 2 (bf16 planes) against form 1 (f32-input MFMA) with hard K and V (Q stays N(0,1) / 4 so that the softmax is not
    one-hot everywhere; heavy K entries still make it nearly so for some rows), against a float64 softmax, in units of 2^-24 sum_j p_j |v_j|."""
    from icepy4d_amd._lib import ptr, stream_ptr
    heads = 4
    g = torch.Generator().manual_seed(seed_of(kind, n))
    q = torch.randn(2, heads, n, 64, generator=g) * 0.5
    k = operands(kind, 2 * heads * n, 64, g).reshape(2, heads, n, 64)
    v = operands(kind, 2 * heads * n, 64, g).reshape(2, heads, n, 64)
    if kind in ("heavy_entries", "heavy_channel"):
        k = k * 0.05                       # logits up to a few hundred instead of 1e5: the exp2 argument reduction is in play, the sum is not one term
    dn = torch.tensor([n, n], dtype=torch.int32, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    for cross in (0, 1):
        refs, masses = [], []
        for z in range(2):
            y = z ^ 1 if cross else z
            p = torch.softmax(q[z].double().cuda() @ k[y].double().cuda().transpose(-1, -2) * 0.125, -1)
            refs.append((p @ v[y].double().cuda()).transpose(0, 1).reshape(n, heads * 64).cpu())
            masses.append((p @ v[y].double().abs().cuda()).transpose(0, 1).reshape(n, heads * 64).cpu())
        ref, mass = torch.stack(refs), torch.stack(masses)
        res = {}
        for form in (0, 1, 2):
            dout = torch.empty(2, n, heads * 64, device="cuda")
            ctx.call("im_flash_attn", ptr(dq), ptr(dk), ptr(dv), ptr(dout), ptr(dn), n, 2, heads, cross | (form << 1), 0.125, stream_ptr())
            torch.cuda.synchronize()
            out = dout.cpu()
            assert torch.isfinite(out).all()
            res[form] = units(out, ref, mass)
        print(f"attn {kind:16s} n={n} cross={cross}: planes (pre-cut) max {res[0][0]:.2f} mean {res[0][1]:.3f} | planes (in-kernel) max {res[2][0]:.2f} mean {res[2][1]:.3f} "
              f"| f32 MFMA max {res[1][0]:.2f} mean {res[1][1]:.3f}  [2^-24 sum p|v|]")
        # the scores go through exp2 of a difference of logits: an error of one unit of the LOGITS' mass (up to hundreds here) is a relative error of the
        # probabilities, so the yardstick, not an absolute number of units, is the bar
        one_sided = kind == "post_relu"       # P > 0 and V >= 0: the residue of every product has one sign over up to 4096 keys
        lim = dict(floor=1.0, mean_ratio=2.0 if one_sided else 1.25, max_ratio=2.25 if one_sided else 1.75)
        assert within(res[0], res[1], **lim) and within(res[2], res[1], **lim), res
        assert res[0][1] < 12, res              # 12 units = 7e-7 of the value's mass, against the 1e-4 of the path


@pytest.mark.parametrize("kind", ["randn", "heavy_entries", "heavy_channel", "post_relu", "zeros_and_short"])
def test_conv_winograd_bf16_planes_on_hard_operands(ctx, kind, monkeypatch):
    """`im_conv3x3_winograd`: the bf16-plane form (the product) against the f32-input MFMA form (`IM_CONV_F32=1`) - the SAME transform, only the
    sixteen products differ - with hard activations and heavy-tailed weights, against an fp64 direct convolution, in units of 2^-24 of the direct
    form's absolute mass (the Winograd transform itself costs a few units in both forms)."""
    from icepy4d_amd._lib import ptr, stream_ptr
    cin, cout, h, w = 64, 64, 40, 48
    g = torch.Generator().manual_seed(seed_of(kind))
    x = operands(kind, 2 * h * w, cin, g).reshape(2, h, w, cin)
    wt = operands("heavy_entries", cout, cin * 9, g).reshape(cout, cin, 3, 3) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    xc = x.permute(0, 3, 1, 2).double().cuda()
    ref = F.conv2d(xc, wt.double().cuda(), b.double().cuda(), padding=1).cpu()
    mass = (F.conv2d(xc.abs(), wt.double().abs().cuda(), b.double().abs().cuda(), padding=1)).cpu()
    dx = dev(x)
    res = {}
    for form in ("bf16x6", "f32"):
        if form == "f32":
            monkeypatch.setenv("IM_CONV_F32", "1")
        else:
            monkeypatch.delenv("IM_CONV_F32", raising=False)
        dout = torch.full((2, h, w, cout), float("nan"), device="cuda")
        ctx.call("im_conv3x3_winograd", ptr(dx), ptr(wt.contiguous()), ptr(b), ptr(dout), 2, h, w, cin, cout, 0, 0, stream_ptr())
        torch.cuda.synchronize()
        out = dout.cpu().permute(0, 3, 1, 2)
        assert torch.isfinite(out).all()
        res[form] = units(out, ref, mass)
    print(f"conv {kind:16s}: bf16 planes max {res['bf16x6'][0]:.2f} mean {res['bf16x6'][1]:.3f} | f32 MFMA max {res['f32'][0]:.2f} mean {res['f32'][1]:.3f}  [2^-24 mass]")
    assert within(res["bf16x6"], res["f32"], floor=0.5, mean_ratio=1.25, max_ratio=1.5), res


if __name__ == "__main__":       # python tests/test_gpu_bf16_stress.py > profiles/r06_bf16_stress.txt : the observed ratios
    import subprocess, sys
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-m", "gpu", "-q", "-s", "-p", "no:cacheprovider"], capture_output=True, text=True)
    lines = [l.lstrip(".F") for l in r.stdout.splitlines() if " max " in l]
    print("\n".join(lines))
    print(r.stdout.splitlines()[-1])
