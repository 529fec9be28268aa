"""Stage-isolated parity of the MFMA kernels (through the C ABI) against plain torch fp32 references.
fp32 tolerance: the kernels accumulate in a different order than torch-CPU, so results agree to ~1e-6
relative to the dot-product magnitude; the tests allow 2e-5 absolute on O(1) values (north_star: 1e-4)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from icepy4d_amd import _lib
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    c = _lib.Context(0)
    yield c
    c.close()


def dev(a):
    return torch.as_tensor(a).cuda().contiguous()


@pytest.mark.parametrize("m,n,k,big", [(64, 64, 32, 0), (100, 65, 256, 0), (4096, 256, 512, 0), (300, 257, 256, 1), (1024, 768, 256, 1),
                                       (64, 64, 32, 2), (100, 65, 256, 2), (4096, 256, 512, 2), (300, 257, 256, 3), (1024, 768, 256, 3)])
def test_gemm_nt(ctx, m, n, k, big):
    """big: bit 0 = 128 x 128 tiles, bit 1 = the product on the bf16 matrix cores (six bf16 products per fp32 product; same tolerance)."""
    from icepy4d_amd._lib import ptr, stream_ptr
    g = torch.Generator().manual_seed(m + n)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    ref = 0.5 * (a.double() @ w.double().t() + b.double())
    da, dw, db = dev(a), dev(w), dev(b)
    dc = torch.full((m, n), float("nan"), device="cuda")
    ctx.call("im_gemm_nt", ptr(da), ptr(dw), ptr(db), ptr(dc), m, n, k, 0.5, big, stream_ptr())
    torch.cuda.synchronize()
    err = (dc.cpu().double() - ref).abs().max().item()
    assert err < 2e-5, err


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("rows,live", [(64, None), (200, (200, 77)), (4096, (4096, 3001)), (33, (1, 33))])
def test_ffn_fused(ctx, rows, live, act, monkeypatch):
    """ffn.0 on cat([x, att]) -> LayerNorm(512) + GELU (LightGlue, `lightglue.py:144-149, 160-162`) or ReLU (SuperGlue's mlp with
    BatchNorm folded, `superglue.py:51-61, 104-116`) -> second linear -> residual in one kernel, against the same chain in
    float64 torch; ragged live-row counts per image; rows past them must stay untouched."""
    from icepy4d_amd._lib import ptr, stream_ptr
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(2, rows, 256, generator=g)
    att = torch.randn(2, rows, 256, generator=g)
    w0 = torch.randn(512, 512, generator=g) / 512 ** 0.5
    b0 = torch.randn(512, generator=g) * 0.1
    lg = 1 + 0.1 * torch.randn(512, generator=g)
    lb = 0.1 * torch.randn(512, generator=g)
    w3 = torch.randn(256, 512, generator=g) / 512 ** 0.5
    b3 = torch.randn(256, generator=g) * 0.1
    h = torch.cat([x, att], -1).double() @ w0.double().t() + b0.double()
    h = F.gelu(F.layer_norm(h, (512,), lg.double(), lb.double(), 1e-5)) if act == 0 else F.relu(h)
    ref = x.double() + h @ w3.double().t() + b3.double()
    dn = None if live is None else dev(torch.tensor(live, dtype=torch.int32))
    hw = [t.contiguous().numpy() for t in (w0, b0, lg, lb, w3, b3)]
    if act == 1:
        hw[2] = hw[3] = None
    outs = {}
    for split in ("0", "1"):      # one block per CU with full-K planes / two with half-K planes (launch_ffn_fused reads the switch per call; default: by grid size)
        monkeypatch.setenv("IM_FFN_SPLIT", split)
        dx, da = dev(x), dev(att)
        ctx.call("im_ffn_fused", act, ptr(dx), ptr(da), *[ptr(t) for t in hw], 2, rows, ptr(dn), stream_ptr())
        torch.cuda.synchronize()
        out = outs[split] = dx.cpu()
        for z in range(2):
            n = rows if live is None else live[z]
            err = (out[z, :n].double() - ref[z, :n]).abs().max().item()
            assert err < 3e-5, (split, z, err)
            assert torch.equal(out[z, n:], x[z, n:]), split
    # the split form (the contraction in two halves, weight steps of one k chunk) forms every sum in the other form's order
    assert torch.equal(outs["0"], outs["1"])


def test_gemm_asymmetric_layout(ctx):
    """A = I with an asymmetric W catches a transposed C-write (guide: MFMA layout check)."""
    from icepy4d_amd._lib import ptr, stream_ptr
    n = 64
    a = torch.eye(n)
    w = torch.arange(n * n, dtype=torch.float32).reshape(n, n)
    dc = torch.zeros(n, n, device="cuda")
    da, dw = dev(a), dev(w)  # keep alive: a temporary would be freed and its address reused
    for mode in (0, 2):      # f32-input MFMA; bf16 planes (integers up to 4095 are exact in three bf16 values)
        dc.zero_()
        ctx.call("im_gemm_nt", ptr(da), ptr(dw), None, ptr(dc), n, n, n, 1.0, mode, stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(dc.cpu(), w.t()), mode


@pytest.mark.parametrize("relu", [1, 0])
@pytest.mark.parametrize("cin,cout,h,w,pool", [(16, 64, 8, 32, 0), (64, 64, 37, 70, 0), (64, 128, 40, 64, 1), (128, 256, 17, 33, 0), (64, 64, 31, 47, 1)])
def test_conv3x3(ctx, cin, cout, h, w, pool, relu):
    from icepy4d_amd._lib import ptr, stream_ptr
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(2, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    if relu:
        ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    dx = dev(x.permute(0, 2, 3, 1))
    ho, wo = ref.shape[-2:]
    dout = torch.full((2, ho, wo, cout), float("nan"), device="cuda")
    ctx.call("im_conv3x3", ptr(dx), ptr(wt.contiguous()), ptr(b), ptr(dout), 2, h, w, cin, cout, relu, pool, stream_ptr())
    torch.cuda.synchronize()
    err = (dout.cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item()
    assert err < 2e-5, err


@pytest.mark.parametrize("relu", [1, 0])
@pytest.mark.parametrize("cin,cout,h,w,pool", [(16, 64, 8, 32, 0), (64, 64, 37, 70, 0), (64, 128, 40, 64, 1), (128, 256, 17, 33, 0), (64, 64, 31, 47, 1),
                                                   # whole regions in x, a partial last region in y: the epilogue's buffer stores with the rows below
                                                   # the image dropped by the range check (the output is NaN-filled and holds two images)
                                                   (64, 64, 36, 64, 0), (64, 64, 44, 96, 1), (128, 128, 20, 32, 0),
                                                   # maps LOWER than one 8-row region (conv3 / conv4 of a wide, flat tile): the scalar store offset of
                                                   # a row below the image alone exceeds the descriptor's record count there - these must take the
                                                   # compared-store path and leave the second image and the NaN fill around it alone
                                                   (64, 64, 2, 32, 0), (128, 128, 4, 48, 0), (64, 128, 6, 32, 0), (64, 64, 6, 64, 1), (64, 64, 4, 32, 1)])
@pytest.mark.parametrize("form", ["bf16x6", "f32"])
def test_conv3x3_winograd(ctx, cin, cout, h, w, pool, relu, form, monkeypatch):
    """Winograd F(2x2, 3x3) on the matrix cores against an fp64 direct convolution; its rounding error is a few 1e-6
    on O(1) outputs (the direct kernel is ~1e-6), far inside the 1e-4 budget of the path. Both forms of the sixteen products: six bf16
    products per fp32 product on the bf16 matrix cores (the product kernel since round 6) and the f32-input MFMA (`IM_CONV_F32=1`)."""
    from icepy4d_amd._lib import ptr, stream_ptr
    if form == "f32":
        monkeypatch.setenv("IM_CONV_F32", "1")
    else:
        monkeypatch.delenv("IM_CONV_F32", raising=False)
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(2, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    if relu:
        ref = F.relu(ref)
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    dx = dev(x.permute(0, 2, 3, 1))
    ho, wo = ref.shape[-2:]
    big = torch.full((4, ho, wo, cout), float("nan"), device="cuda")            # the two output images with a NaN image in front and behind
    dout = big[1:3]
    ctx.call("im_conv3x3_winograd", ptr(dx), ptr(wt.contiguous()), ptr(b), ptr(dout), 2, h, w, cin, cout, relu, pool, stream_ptr())
    torch.cuda.synchronize()
    err = (dout.cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item()
    assert err < 2e-5, err
    assert torch.isnan(big[0]).all() and torch.isnan(big[3]).all()              # no stray store on either side


@pytest.mark.parametrize("n0,n1,cross", [(128, 128, 0), (300, 257, 0), (300, 257, 1), (1000, 77, 1), (64, 1, 1)])
def test_flash_attn(ctx, n0, n1, cross):
    from icepy4d_amd._lib import ptr, stream_ptr
    nmax, heads = 1024, 4
    g = torch.Generator().manual_seed(n0 * 7 + n1)
    q = torch.randn(2, heads, nmax, 64, generator=g)
    k = torch.randn(2, heads, nmax, 64, generator=g)
    v = torch.randn(2, heads, nmax, 64, generator=g)
    ns = [n0, n1]
    scale = 0.125
    dout = torch.full((2, nmax, heads * 64), float("nan"), device="cuda")
    dn = torch.tensor(ns, dtype=torch.int32, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    ctx.call("im_flash_attn", ptr(dq), ptr(dk), ptr(dv), ptr(dout), ptr(dn), nmax, 2, heads, cross, scale, stream_ptr())
    torch.cuda.synchronize()
    out = dout.cpu()
    for z in range(2):
        y = z ^ 1 if cross else z
        qq, kk, vv = q[z, :, :ns[z]].double(), k[y, :, :ns[y]].double(), v[y, :, :ns[y]].double()
        att = torch.softmax(qq @ kk.transpose(-1, -2) * scale, -1) @ vv  # [h, n, 64]
        ref = att.transpose(0, 1).reshape(ns[z], heads * 64)
        err = (out[z, :ns[z]].double() - ref).abs().max().item()
        assert err < 2e-5, (z, err)
        assert torch.isnan(out[z, ns[z]:]).all()  # rows beyond the live count are untouched


@pytest.mark.parametrize("n", [1024, 4096])
def test_flash_attn_bf16_form_is_as_accurate_as_the_f32_mfma_form(ctx, n):
    """The product path (csrc/attention_bx.hip: fp32 operands as three bf16 values, six bf16 products per fp32 product on the bf16
    matrix cores) against the f32-input MFMA kernel of rounds 1-5 (csrc/attention.hip, `cross` bit 1 of the stage entry) on the same
    inputs, both against a float64 softmax: the two errors must be of one size - the bf16 form's maximum and mean within 1.25 x the f32
    form's. Measured at 4096 keys: maximum 2.9e-7 against 4.5e-7, mean 1.87e-8 against 1.68e-8 (the matrix core adds 16 products before
    it rounds once, which lowers the spread and leaves a slightly larger bias: profiles/r05_bf16x_probe.txt); at 1024 both are lower."""
    from icepy4d_amd._lib import ptr, stream_ptr
    heads = 4
    g = torch.Generator().manual_seed(n)
    q = torch.randn(2, heads, n, 64, generator=g)
    k = torch.randn(2, heads, n, 64, generator=g)
    v = torch.randn(2, heads, n, 64, generator=g)
    dn = torch.tensor([n, n], dtype=torch.int32, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    for cross in (0, 1):
        refs = []
        for z in range(2):
            y = z ^ 1 if cross else z
            att = torch.softmax(q[z].double().cuda() @ k[y].double().cuda().transpose(-1, -2) * 0.125, -1) @ v[y].double().cuda()
            refs.append(att.transpose(0, 1).reshape(n, heads * 64).cpu())
        errs = {}
        outs = {}
        for form in (0, 1, 2):   # 0: bf16 planes cut by the first launch (the product), 1: f32-input MFMA, 2: bf16 planes cut inside the kernel
            dout = torch.empty(2, n, heads * 64, device="cuda")
            ctx.call("im_flash_attn", ptr(dq), ptr(dk), ptr(dv), ptr(dout), ptr(dn), n, 2, heads, cross | (form << 1), 0.125, stream_ptr())
            torch.cuda.synchronize()
            outs[form] = dout.cpu()
            e = torch.stack([(outs[form][z].double() - refs[z]).abs() for z in range(2)])
            errs[form] = (e.max().item(), e.mean().item())
        assert errs[2][0] <= 1.25 * errs[1][0] and errs[2][1] <= 1.25 * errs[1][1], errs
        print(f"n={n} cross={cross}: bf16 form max {errs[0][0]:.3e} mean {errs[0][1]:.3e} | f32 form max {errs[1][0]:.3e} mean {errs[1][1]:.3e}")
        assert errs[0][0] < 2e-5 and errs[1][0] < 2e-5
        assert errs[0][0] <= 1.25 * errs[1][0] and errs[0][1] <= 1.25 * errs[1][1], errs


@pytest.mark.parametrize("n0,n1", [(1, 63), (64, 65), (127, 129), (191, 193), (130, 1), (257, 256)])
def test_flash_attn_key_group_edges(ctx, n0, n1):
    """Key counts around the 64-key tile / 128-key step boundaries of the two-key-group kernel, self and cross; the
    K / V rows beyond the live counts hold NaN (the kernel must not read them: buffer range check) and the output rows
    beyond the live counts must stay untouched."""
    from icepy4d_amd._lib import ptr, stream_ptr
    nmax, heads = 320, 4
    g = torch.Generator().manual_seed(1000 * n0 + n1)
    q = torch.randn(2, heads, nmax, 64, generator=g)
    k = torch.randn(2, heads, nmax, 64, generator=g)
    v = torch.randn(2, heads, nmax, 64, generator=g)
    ns = [n0, n1]
    kp, vp = k.clone(), v.clone()
    for z in range(2):
        kp[z, :, ns[z]:] = float("nan")
        vp[z, :, ns[z]:] = float("nan")
    dn = torch.tensor(ns, dtype=torch.int32, device="cuda")
    dq, dk, dv = dev(q), dev(kp), dev(vp)
    for cross in (0, 1):
        dout = torch.full((2, nmax, heads * 64), float("nan"), device="cuda")
        ctx.call("im_flash_attn", ptr(dq), ptr(dk), ptr(dv), ptr(dout), ptr(dn), nmax, 2, heads, cross, 0.125, stream_ptr())
        torch.cuda.synchronize()
        out = dout.cpu()
        for z in range(2):
            y = z ^ 1 if cross else z
            qq, kk, vv = q[z, :, :ns[z]].double(), k[y, :, :ns[y]].double(), v[y, :, :ns[y]].double()
            ref = (torch.softmax(qq @ kk.transpose(-1, -2) * 0.125, -1) @ vv).transpose(0, 1).reshape(ns[z], heads * 64)
            err = (out[z, :ns[z]].double() - ref).abs().max().item()
            assert err < 2e-5, (cross, z, err)
            assert torch.isnan(out[z, ns[z]:]).all()


def test_flash_attn_large_dynamic_range(ctx):
    """Scores spanning hundreds of units with the row maxima arriving late: exercises the lazily raised reference maximum
    (the rescale path must fire, P must never overflow) against an fp64 softmax."""
    from icepy4d_amd._lib import ptr, stream_ptr
    n, heads = 1024, 4
    g = torch.Generator().manual_seed(77)
    q = torch.randn(2, heads, n, 64, generator=g) * 4.0
    k = torch.randn(2, heads, n, 64, generator=g) * 4.0
    v = torch.randn(2, heads, n, 64, generator=g)
    k[:, :, 900:] *= 3.0          # the largest scores sit in the last tiles
    dn = torch.tensor([n, n], dtype=torch.int32, device="cuda")
    dq, dk, dv = dev(q), dev(k), dev(v)
    dout = torch.empty(2, n, heads * 64, device="cuda")
    ctx.call("im_flash_attn", ptr(dq), ptr(dk), ptr(dv), ptr(dout), ptr(dn), n, 2, heads, 0, 1.0, stream_ptr())
    torch.cuda.synchronize()
    out = dout.cpu()
    assert torch.isfinite(out).all()
    for z in range(2):
        ref = (torch.softmax(q[z].double() @ k[z].double().transpose(-1, -2), -1) @ v[z].double()).transpose(0, 1).reshape(n, heads * 64)
        # scores of magnitude ~600 carry ~1e-4 of fp32 rounding each, which the softmax turns into relative error of P:
        # the yardstick is what plain fp32 arithmetic (torch CPU) loses on the same inputs
        ref32 = (torch.softmax(q[z] @ k[z].transpose(-1, -2), -1) @ v[z]).transpose(0, 1).reshape(n, heads * 64)
        err32 = (ref32.double() - ref).abs().max().item()
        err = (out[z].double() - ref).abs().max().item()
        assert err < max(4 * err32, 1e-4), (z, err, err32)




@pytest.mark.parametrize("nq,nk", [(1500, 4096), (900, 4096), (2048, 2049), (1025, 3000), (4096, 700), (1024, 4096), (2049, 1024), (130, 3967)])
def test_flash_attn_split_kv_regimes_at_4096(ctx, nq, nk):
    """The device-side split-KV of the attention kernel (`attention.hip`: the keys of a 128-query block are cut into 1 / 2 / 4 ranges
    of whole steps for nq > 2048 / nq in (1024, 2048] / nq <= 1024; the last block to arrive merges the parked partials in split order)
    at the widths a 4096-keypoint pair walks through while LightGlue prunes it (`lightglue/lightglue.py:495-510`): image 0 holds nq
    live rows, image 1 nk, self and cross, against an fp64 softmax. K / V rows beyond the live counts hold NaN (never read: buffer
    range check), so do the query rows beyond them, and the output rows beyond the live counts stay untouched. Each shape runs twice
    through the same scratch (the merge counters reset themselves)."""
    from icepy4d_amd._lib import ptr, stream_ptr
    nmax, heads = 4096, 4
    g = torch.Generator().manual_seed(nq * 5 + nk)
    q = torch.randn(2, heads, nmax, 64, generator=g)
    k = torch.randn(2, heads, nmax, 64, generator=g)
    v = torch.randn(2, heads, nmax, 64, generator=g)
    ns = [nq, nk]
    qp, kp, vp = q.clone(), k.clone(), v.clone()
    for z in range(2):
        qp[z, :, ns[z]:] = float("nan"); kp[z, :, ns[z]:] = float("nan"); vp[z, :, ns[z]:] = float("nan")
    dn = torch.tensor(ns, dtype=torch.int32, device="cuda")
    dq, dk, dv = dev(qp), dev(kp), dev(vp)
    for cross in (0, 1):
        outs = []
        for rep in range(2):
            dout = torch.full((2, nmax, heads * 64), float("nan"), device="cuda")
            ctx.call("im_flash_attn", ptr(dq), ptr(dk), ptr(dv), ptr(dout), ptr(dn), nmax, 2, heads, cross, 0.125, stream_ptr())
            torch.cuda.synchronize()
            outs.append(dout.cpu())
        out = outs[0]
        assert torch.equal(outs[0].nan_to_num(7.0), outs[1].nan_to_num(7.0))         # deterministic merge order
        for z in range(2):
            y = z ^ 1 if cross else z
            for hd in range(heads):
                qq, kk, vv = q[z, hd, :ns[z]].double(), k[y, hd, :ns[y]].double(), v[y, hd, :ns[y]].double()
                ref = torch.softmax(qq @ kk.t() * 0.125, -1) @ vv
                err = (out[z, :ns[z], hd * 64:(hd + 1) * 64].double() - ref).abs().max().item()
                assert err < 2e-5, (cross, z, hd, err)
            assert torch.isnan(out[z, ns[z]:]).all()
