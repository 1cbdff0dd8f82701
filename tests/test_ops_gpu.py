"""GPU: each HIP kernel through the C ABI against a plain PyTorch fp32 reference of the same op
(tolerances: fp32 path 1e-4 relative to the tensor's max; bf16 path 2e-2)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from stlpose_amd import capi  # noqa: E402
from stlpose_amd.engine import choose_tile  # noqa: E402

DT = {"fp32": (capi.F32, torch.float32, 2e-4), "bf16": (capi.BF16, torch.bfloat16, 3e-2), "f16": (capi.F16, torch.float16, 4e-3),
      "mixed": (capi.BF16, torch.bfloat16, 3e-2)}   # mixed: GRADIENT tensors bf16 (this entry), forward tensors f16 (FDT)
FDT = {"fp32": (capi.F32, torch.float32), "bf16": (capi.BF16, torch.bfloat16), "mixed": (capi.F16, torch.float16)}   # forward tensors
EPS = 1e-5


def stream():
    return torch.cuda.current_stream().cuda_stream


def nhwc(x, td):
    return x.permute(0, 2, 3, 1).contiguous().to(td)


def from_nhwc(t, B, H, W, Cc):
    return t.view(B, H, W, Cc).float().permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


def stats_of(y_nhwc, Cc):
    """[NSHARD][2C] fp64 sums as the conv epilogue would produce them (all in shard 0)."""
    f = y_nhwc.float().reshape(-1, Cc).double()
    st = torch.zeros(capi.NSHARD, 2, Cc, dtype=torch.float64, device=y_nhwc.device)
    st[0, 0], st[0, 1] = f.sum(0), (f * f).sum(0)
    return st


def bn_src(xt, stats, gamma, beta, n, relu):
    s = capi.Src()
    s.x, s.mode, s.relu = xt.data_ptr(), capi.SRC_BN, int(relu)
    s.stats, s.gamma, s.beta = stats.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    s.inv_count, s.eps = 1.0 / n, EPS
    return s


def test_selftest_lane_maps():
    out = torch.full((4,), -1.0, device="cuda")
    capi.call("stl_selftest_mfma", out.data_ptr(), stream())
    torch.cuda.synchronize()
    assert out.tolist() == [0.0, 0.0, 0.0, 0.0], f"bf16-mfma / f32-mfma / tr-read / f64-atomic errors: {out.tolist()}"


CONV_CASES = [
    # B, H, W, Ci, Co, ks, stride
    (2, 24, 16, 32, 32, 3, 1),
    (3, 12, 9, 64, 64, 3, 1),
    (2, 24, 18, 32, 64, 3, 2),
    (2, 8, 6, 256, 64, 1, 1),
    (2, 16, 12, 48, 48, 3, 1),     # ragged channel chunk (W48 widths)
    (1, 30, 22, 128, 32, 3, 1),    # odd tile edges
    (2, 13, 11, 32, 32, 3, 2),     # odd input size, stride 2
    (2, 12, 9, 96, 192, 1, 1),     # streaming 1x1 kernel, ragged K stage / channel block (W48 widths)
    (3, 11, 7, 192, 96, 1, 1),     # streaming 1x1 kernel, pixel count not a multiple of the tile
]


@pytest.mark.parametrize("tile", ["auto", "explicit"])
@pytest.mark.parametrize("dt", ["fp32", "bf16", "f16"])
@pytest.mark.parametrize("case", CONV_CASES + [(4, 48, 36, 32, 32, 3, 1), (4, 48, 36, 64, 64, 3, 1), (4, 24, 18, 128, 128, 3, 1),
                                               (4, 48, 36, 64, 256, 1, 1), (2, 12, 9, 256, 256, 3, 1)])
def test_conv_forward_plain_and_stats(case, dt, tile):
    code, td, tol = DT[dt]
    B, H, W, Ci, Co, ks, s = case
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, Ci, H, W, device="cuda", generator=g)
    w = torch.randn(Co, Ci, ks, ks, device="cuda", generator=g) / math.sqrt(Ci * ks * ks)
    xt, wt = nhwc(x, td), w.permute(0, 2, 3, 1).contiguous().to(td)
    pad = 1 if ks == 3 else 0
    ref = F.conv2d(xt.float().permute(0, 3, 1, 2), wt.float().permute(0, 3, 1, 2), stride=s, padding=pad)
    Ho, Wo = ref.shape[2:]
    out = torch.full((B * Ho * Wo * Co,), float("nan"), device="cuda", dtype=td)
    st = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device="cuda")
    p = capi.Conv()
    p.shape = -1
    p.dtype, p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co = code, B, H, W, Ci, Ho, Wo, Co
    p.ks, p.stride = ks, s
    p.TH, p.TW = choose_tile(B, Ho, Wo, s, ks, xt.element_size()) if tile == "explicit" else (0, 0)
    p.src.x, p.src.mode = xt.data_ptr(), capi.SRC_PLAIN
    p.w, p.out, p.out_stats = wt.data_ptr(), out.data_ptr(), st.data_ptr()
    capi.call("stl_conv_forward", C.byref(p), stream())
    torch.cuda.synchronize()
    got = from_nhwc(out, B, Ho, Wo, Co)
    assert not torch.isnan(got).any()
    assert relerr(got, ref) < tol
    # statistics are those of the STORED tensor
    sums = st.view(capi.NSHARD, 2, Co).sum(0)
    stored = out.view(-1, Co).double()
    assert torch.allclose(sums[0], stored.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(sums[1], (stored * stored).sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("dt", ["fp32", "bf16", "mixed"])
@pytest.mark.parametrize("case", [(2, 24, 16, 32, 32, 3, 1), (2, 24, 18, 32, 64, 3, 2), (2, 8, 6, 128, 64, 1, 1),
                                  (4, 48, 36, 64, 64, 3, 1), (4, 24, 18, 128, 128, 3, 1), (2, 24, 18, 64, 256, 1, 1),
                                  (2, 24, 18, 96, 72, 3, 1)])
def test_conv_bn_relu_chain_forward_backward(case, dt):
    """x --BN(relu) on load--> conv --> y ; backward: BN-backward on load, ReLU mask + r1/r2 in the
    data-gradient epilogue, weight gradient slabs.  Reference: torch autograd through
    batch_norm(train) -> relu -> conv2d -> batch_norm(train).  mixed: the forward tensors (x0, y, forward weights) are
    f16, the gradients (dt, dx, data-gradient weights) bf16; the backward kernels read the forward tensors as `ydtype`."""
    code, td, tol = DT[dt]
    fcode, ftd = FDT[dt]
    ydt = fcode if fcode != code else 0
    B, H, W, Ci, Co, ks, s = case
    g = torch.Generator(device="cuda").manual_seed(2)
    pad = 1 if ks == 3 else 0
    x0 = torch.randn(B, Ci, H, W, device="cuda", generator=g) * 1.5 + 0.3
    x0t = nhwc(x0, ftd)
    x0r = x0t.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)  # the stored raw tensor
    w = (torch.randn(Co, Ci, ks, ks, device="cuda", generator=g) / math.sqrt(Ci * ks * ks))
    wt = w.permute(0, 2, 3, 1).contiguous().to(ftd)
    wr = wt.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    g1 = (torch.rand(Ci, device="cuda", generator=g) + 0.5).requires_grad_(True)
    b1 = (torch.rand(Ci, device="cuda", generator=g) - 0.5).requires_grad_(True)
    g2 = (torch.rand(Co, device="cuda", generator=g) + 0.5).requires_grad_(True)
    b2 = (torch.rand(Co, device="cuda", generator=g) - 0.5).requires_grad_(True)
    h = F.relu(F.batch_norm(x0r, None, None, g1, b1, True, 0.1, EPS))
    y = F.conv2d(h, wr, stride=s, padding=pad)
    t = F.batch_norm(y, None, None, g2, b2, True, 0.1, EPS)
    Ho, Wo = y.shape[2:]
    dt_up = torch.randn(B, Co, Ho, Wo, device="cuda", generator=g)
    dtt = nhwc(dt_up, td)
    t.backward(dtt.float().permute(0, 3, 1, 2))
    # ---------------- HIP forward
    st1 = stats_of(x0t, Ci)
    yk = torch.empty(B * Ho * Wo * Co, device="cuda", dtype=ftd)
    st2 = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device="cuda")
    p = capi.Conv()
    p.shape = -1
    p.dtype, p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co = fcode, B, H, W, Ci, Ho, Wo, Co
    p.ks, p.stride = ks, s
    p.TH, p.TW = 0, 0
    p.src = bn_src(x0t, st1, g1.detach(), b1.detach(), B * H * W, True)
    p.w, p.out, p.out_stats = wt.data_ptr(), yk.data_ptr(), st2.data_ptr()
    capi.call("stl_conv_forward", C.byref(p), stream())
    torch.cuda.synchronize()
    assert relerr(from_nhwc(yk, B, Ho, Wo, Co), y.detach()) < (4e-3 if dt == "mixed" else tol)
    # ---------------- backward reductions for BN2 (what fuse_backward would have produced)
    ykf = yk.view(-1, Co).double()
    mean2 = ykf.mean(0)
    rstd2 = 1.0 / torch.sqrt(ykf.var(0, unbiased=False) + EPS)
    dtf = dtt.view(-1, Co).double()
    rst2 = torch.zeros(capi.NSHARD, 2, Co, dtype=torch.float64, device="cuda")
    rst2[0, 0] = dtf.sum(0)
    rst2[0, 1] = (dtf * (ykf - mean2) * rstd2).sum(0)
    gs = capi.Src()
    gs.x, gs.y, gs.mode = dtt.data_ptr(), yk.data_ptr(), capi.SRC_BNBWD
    gs.stats, gs.rstats, gs.gamma = st2.data_ptr(), rst2.data_ptr(), g2.detach().data_ptr()
    gs.inv_count, gs.eps = 1.0 / (B * Ho * Wo), EPS
    # ---------------- weight gradient
    wg = capi.Wgrad()
    wg.dtype, wg.B, wg.Hi, wg.Wi, wg.Ci, wg.Ho, wg.Wo, wg.Co = code, B, H, W, Ci, Ho, Wo, Co
    wg.ks, wg.stride, wg.ydtype = ks, s, ydt
    ctile = capi.lib().stl_wgrad_chunk(C.byref(wg))   # 64: wide-channel kernel variant (bf16 1x1 layers with Co, Ci >= 64)
    assert ctile == (64 if (dt in ("bf16", "mixed") and s == 1 and Ci >= 64 and Co >= 64) else 32)
    wg.TH, wg.TW = choose_tile(B, Ho, Wo, s, ks, x0t.element_size(), bn_cols=32, maxhalo=((256 if ks == 3 else 192) if ctile == 64 else 576))
    npt = math.ceil(B * (Ho + 1) / wg.TH) * math.ceil(Wo / wg.TW)
    wg.nsplit = min(3, npt)
    part = torch.full((wg.nsplit * Co * ks * ks * Ci,), float("nan"), device="cuda")
    wg.h, wg.g, wg.partial = p.src, gs, part.data_ptr()
    capi.call("stl_conv_wgrad", C.byref(wg), stream())
    torch.cuda.synchronize()
    dw = part.view(wg.nsplit, Co, ks * ks, Ci).sum(0).view(Co, ks, ks, Ci).permute(0, 3, 1, 2)
    assert not torch.isnan(dw).any()
    assert relerr(dw, wr.grad) < tol * 3
    # ---------------- data gradient with ReLU mask + BN1 reductions
    wb = wt.float().view(Co, ks * ks, Ci).flip(1).permute(2, 1, 0).contiguous().to(td)  # [Ci][8-tap][Co]
    dx = torch.full((B * H * W * Ci,), float("nan"), device="cuda", dtype=td)
    red = torch.zeros(capi.NSHARD * 2 * Ci, dtype=torch.float64, device="cuda")
    d = capi.Conv()
    d.shape = -1
    d.dtype, d.B, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co = code, B, Ho, Wo, Co, H, W, Ci
    d.ks, d.stride, d.stuff, d.ydtype = ks, 1, int(s == 2), ydt
    d.TH, d.TW = 0, 0
    d.src, d.w, d.out = gs, wb.data_ptr(), dx.data_ptr()
    d.mask_y, d.mask_bn, d.red = x0t.data_ptr(), p.src, red.data_ptr()
    capi.call("stl_conv_forward", C.byref(d), stream())
    torch.cuda.synchronize()
    # reference for the masked gradient wrt the BN1 output: dL/dh * (h > 0)
    x0f = x0t.float().view(-1, Ci).double()
    mean1, rstd1 = x0f.mean(0), 1.0 / torch.sqrt(x0f.var(0, unbiased=False) + EPS)
    xhat = (x0f - mean1) * rstd1
    dxk = dx.view(-1, Ci).double()
    assert not torch.isnan(dxk).any()
    # dgamma1 = sum dt1 * xhat, dbeta1 = sum dt1 from torch
    r = red.view(capi.NSHARD, 2, Ci).sum(0)
    assert relerr(r[0], b1.grad.double()) < tol * 3
    assert relerr(r[1], g1.grad.double()) < tol * 3
    # full BN1 backward from the kernel's dt1 must reproduce dL/dx0
    n = B * H * W
    dx0 = (g1.detach().double() * rstd1) * (dxk - r[0] / n - xhat * r[1] / n)
    ref = x0r.grad.permute(0, 2, 3, 1).reshape(-1, Ci).double()
    assert relerr(dx0, ref) < tol * 3


@pytest.mark.parametrize("dt", ["fp32", "bf16", "mixed"])
@pytest.mark.parametrize("case", [(4, 48, 36, 64, 64), (3, 24, 18, 128, 128), (2, 24, 18, 96, 72), (2, 24, 18, 64, 192)])
def test_conv_two_image_block_equals_single_image(case, dt, monkeypatch):
    """Block shape 3 (two LDS images, one barrier per stage, staging spread over the MFMA taps; round 4) against block
    shape 2 (one image) on the same tile: same chunk and tap order, so outputs are BIT-identical -- forward with a
    BN + ReLU source and the data gradient with BatchNorm-backward on load and every epilogue operand.  A 16-block grid
    makes every block walk several tiles (the new-tile path of the staging slots)."""
    code, td, tol = DT[dt]
    fcode, ftd = FDT[dt]
    ydt = fcode if fcode != code else 0
    B, H, W, Ci, Co = case
    monkeypatch.setenv("STL_CONV_GRID_CAP", "16")
    monkeypatch.setenv("STL_CONV_R2", "0")   # the planner's round-4 choice (round 5 plans the two-per-CU kernel for these layers: next test)
    g = torch.Generator(device="cuda").manual_seed(11)
    x0t = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=g) * 1.5 + 0.3, ftd)
    wt = (torch.randn(Co, 3, 3, Ci, device="cuda", generator=g) / math.sqrt(9 * Ci)).to(ftd)
    g1, b1 = torch.rand(Ci, device="cuda", generator=g) + 0.5, torch.rand(Ci, device="cuda", generator=g) - 0.5
    g2 = torch.rand(Co, device="cuda", generator=g) + 0.5
    st1 = stats_of(x0t, Ci)

    def forward(shape):
        yk = torch.full((B * H * W * Co,), float("nan"), device="cuda", dtype=ftd)
        st2 = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device="cuda")
        p = capi.Conv()
        p.dtype, p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co, p.ks, p.stride = fcode, B, H, W, Ci, H, W, Co, 3, 1
        p.TH, p.TW, p.shape = 0, 0, -1
        p.src = bn_src(x0t, st1, g1, b1, B * H * W, True)
        p.w, p.out, p.out_stats = wt.data_ptr(), yk.data_ptr(), st2.data_ptr()
        capi.call("stl_conv_plan", C.byref(p))
        assert p.shape == 3, f"planner chose block shape {p.shape}"
        p.shape = shape
        capi.call("stl_conv_forward", C.byref(p), stream())
        torch.cuda.synchronize()
        return yk, st2, p
    y3, s3, p3 = forward(3)
    y2, s2, _ = forward(2)
    assert not torch.isnan(y3.float()).any()
    assert torch.equal(y3, y2)
    assert torch.allclose(s3, s2, rtol=1e-12, atol=1e-9)
    # data gradient: BNBWD source (dt, y), addend, block-end mask, ReLU mask of the producing BN + its reductions
    dtt = nhwc(torch.randn(B, Co, H, W, device="cuda", generator=g), td)
    ykf, dtf = y3.view(-1, Co).double(), dtt.view(-1, Co).double()
    mean2, rstd2 = ykf.mean(0), 1.0 / torch.sqrt(ykf.var(0, unbiased=False) + EPS)
    rst2 = torch.zeros(capi.NSHARD, 2, Co, dtype=torch.float64, device="cuda")
    rst2[0, 0], rst2[0, 1] = dtf.sum(0), (dtf * (ykf - mean2) * rstd2).sum(0)
    gs = capi.Src()
    gs.x, gs.y, gs.mode = dtt.data_ptr(), y3.data_ptr(), capi.SRC_BNBWD
    gs.stats, gs.rstats, gs.gamma = s3.data_ptr(), rst2.data_ptr(), g2.data_ptr()
    gs.inv_count, gs.eps = 1.0 / (B * H * W), EPS
    wb = wt.float().view(Co, 9, Ci).flip(1).permute(2, 1, 0).contiguous().to(td)
    addend = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=g), td)
    zmask = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=g), ftd)

    def dgrad(shape):
        dx = torch.full((B * H * W * Ci,), float("nan"), device="cuda", dtype=td)
        red = torch.zeros(capi.NSHARD * 2 * Ci, dtype=torch.float64, device="cuda")
        d = capi.Conv()
        d.dtype, d.B, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co, d.ks, d.stride, d.ydtype = code, B, H, W, Co, H, W, Ci, 3, 1, ydt
        d.TH, d.TW, d.shape = 0, 0, -1
        d.src, d.w, d.out = gs, wb.data_ptr(), dx.data_ptr()
        d.mask_y, d.mask_bn, d.red = x0t.data_ptr(), p3.src, red.data_ptr()
        d.addend, d.mask_z = addend.data_ptr(), zmask.data_ptr()
        capi.call("stl_conv_plan", C.byref(d))
        assert d.shape == 3, f"planner chose block shape {d.shape}"
        d.shape = shape
        capi.call("stl_conv_forward", C.byref(d), stream())
        torch.cuda.synchronize()
        return dx, red
    dx3, r3 = dgrad(3)
    dx2, r2 = dgrad(2)
    assert not torch.isnan(dx3.float()).any()
    assert torch.equal(dx3, dx2)
    # the reductions are per-lane fp32 sums (fused multiply-adds in one instantiation, separate ones in the other) -> fp64 atomics
    assert torch.allclose(r3, r2, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("cap", ["16", ""])
@pytest.mark.parametrize("dt", ["bf16", "mixed"])
@pytest.mark.parametrize("case", [(4, 48, 36, 64, 64), (3, 24, 18, 128, 128), (2, 24, 18, 96, 64), (2, 24, 18, 64, 192), (2, 96, 72, 64, 64), (3, 13, 11, 256, 128),
                                  (2, 20, 14, 128, 224), (4, 96, 72, 32, 32), (3, 48, 36, 32, 32), (2, 21, 13, 32, 16)])
def test_conv_two_per_cu_kernel_equals_one_per_cu_block(case, dt, cap, monkeypatch):
    """conv_r2_kernel (round 5: <= 128 VGPRs, <= 76 KB of LDS, filters by LDS-DMA into an unpadded swizzled image, planar halo
    image; block shapes 5 = 256 px and 6 = 128 px) against conv_core_kernel's one-per-CU block (shape 3) on the same launch:
    same chunk and tap order per output pixel, so the outputs are BIT-identical whatever the tile -- forward (BatchNorm + ReLU
    source, statistics) in both pixel counts, the block-end source, and the data gradient with every operand set the planner
    emits.  Statistics / reductions are block-wise fp32 partial sums -> equal to rounding.  cap 16: every block walks several tiles."""
    code, td, tol = DT[dt]
    fcode, ftd = FDT[dt]
    ydt = fcode if fcode != code else 0
    B, H, W, Ci, Co = case
    c32 = Ci == 32 and Co <= 32          # the C <= 32 layers: shape 8 (or its small-map fallback 4) -> the 256 x 32 form, shape 10
    NEW = "31" if c32 else "3"           # STL_CONV_R2: forward + data gradients (+ deep small maps + C <= 32)
    if cap:
        monkeypatch.setenv("STL_CONV_GRID_CAP", cap)
    g = torch.Generator(device="cuda").manual_seed(15)
    x0t = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=g) * 1.5 + 0.3, ftd)
    skip = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=g), ftd)
    wt = (torch.randn(Co, 3, 3, Ci, device="cuda", generator=g) / math.sqrt(9 * Ci)).to(ftd)
    g1, b1 = torch.rand(Ci, device="cuda", generator=g) + 0.5, torch.rand(Ci, device="cuda", generator=g) - 0.5
    g2 = torch.rand(Co, device="cuda", generator=g) + 0.5
    st1 = stats_of(x0t, Ci)
    tot = lambda st: st.view(capi.NSHARD, -1).sum(0)   # noqa: E731  (which shard a block adds to depends on the grid, i.e. on the tile)

    def forward(r2, want, bnadd=False):
        monkeypatch.setenv("STL_CONV_R2", r2)
        yk = torch.full((B * H * W * Co,), float("nan"), device="cuda", dtype=ftd)
        zk = torch.full((B * H * W * Ci,), float("nan"), device="cuda", dtype=ftd)
        st2 = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device="cuda")
        p = capi.Conv()
        p.dtype, p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co, p.ks, p.stride = fcode, B, H, W, Ci, H, W, Co, 3, 1
        p.TH, p.TW, p.shape = 0, 0, -1
        p.src = bn_src(x0t, st1, g1, b1, B * H * W, True)
        if bnadd:
            p.src.mode, p.src.y, p.src_out = capi.SRC_BNADD, skip.data_ptr(), zk.data_ptr()
        p.w, p.out, p.out_stats = wt.data_ptr(), yk.data_ptr(), st2.data_ptr()
        capi.call("stl_conv_plan", C.byref(p))
        assert p.shape in want, f"planner chose block shape {p.shape}, expected one of {want}"
        capi.call("stl_conv_forward", C.byref(p), stream())
        torch.cuda.synchronize()
        assert capi.lib().stl_last_kernel().decode().startswith("conv_r2_kernel") == (p.shape in (5, 6, 10))
        return yk, st2, zk, p
    OLD = (8, 4) if c32 else (3,)
    y3, s3, _, p3 = forward("0", OLD)
    for r2, want in (((NEW, (10,)),) if c32 else (("3", (5,)), ("7", (6,)))):
        y5, s5, _, _ = forward(r2, want)
        assert not torch.isnan(y5.float()).any() and torch.equal(y5, y3), (r2, want)
        assert torch.allclose(tot(s5), tot(s3), rtol=1e-6, atol=1e-4)
    yb3, sb3, zb3, _ = forward("0", OLD, bnadd=True)
    for r2, want in (((NEW, (10,)),) if c32 else (("3", (5,)), ("7", (6,)))):
        yb, sb, zb, _ = forward(r2, want, bnadd=True)
        assert not torch.isnan(yb.float()).any() and torch.equal(yb, yb3) and torch.equal(zb, zb3), (r2, want)
        assert torch.allclose(tot(sb), tot(sb3), rtol=1e-6, atol=1e-4)
    # data gradients: BNBWD source (dt, y) and the four operand sets
    dtt = nhwc(torch.randn(B, Co, H, W, device="cuda", generator=g), td)
    ykf, dtf = y3.view(-1, Co).double(), dtt.view(-1, Co).double()
    mean2, rstd2 = ykf.mean(0), 1.0 / torch.sqrt(ykf.var(0, unbiased=False) + EPS)
    rst2 = torch.zeros(capi.NSHARD, 2, Co, dtype=torch.float64, device="cuda")
    rst2[0, 0], rst2[0, 1] = dtf.sum(0), (dtf * (ykf - mean2) * rstd2).sum(0)
    gs = capi.Src()
    gs.x, gs.y, gs.mode = dtt.data_ptr(), y3.data_ptr(), capi.SRC_BNBWD
    gs.stats, gs.rstats, gs.gamma = s3.data_ptr(), rst2.data_ptr(), g2.data_ptr()
    gs.inv_count, gs.eps = 1.0 / (B * H * W), EPS
    wb = wt.float().view(Co, 9, Ci).flip(1).permute(2, 1, 0).contiguous().to(td)
    addend = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=g), td)
    zmask = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=g), ftd)

    def dgrad(r2, want, ops):
        monkeypatch.setenv("STL_CONV_R2", r2)
        dx = torch.full((B * H * W * Ci,), float("nan"), device="cuda", dtype=td)
        red = torch.zeros(capi.NSHARD * 2 * Ci, dtype=torch.float64, device="cuda")
        d = capi.Conv()
        d.dtype, d.B, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co, d.ks, d.stride, d.ydtype = code, B, H, W, Co, H, W, Ci, 3, 1, ydt
        d.TH, d.TW, d.shape = 0, 0, -1
        d.src, d.w, d.out = gs, wb.data_ptr(), dx.data_ptr()
        if "my" in ops:
            d.mask_y, d.mask_bn, d.red = x0t.data_ptr(), bn_src(x0t, st1, g1, b1, B * H * W, "mz" not in ops), red.data_ptr()
        if "ad" in ops:
            d.addend = addend.data_ptr()
        if "mz" in ops:
            d.mask_z = zmask.data_ptr()
        capi.call("stl_conv_plan", C.byref(d))
        if want == OLDD and d.shape not in OLDD:
            return None, d.shape   # e.g. Co >= 256 on a small map: the wave-specialised kernel's layer, not this test's
        assert d.shape in want, f"planner chose block shape {d.shape}, expected one of {want}"
        capi.call("stl_conv_forward", C.byref(d), stream())
        torch.cuda.synchronize()
        assert capi.lib().stl_last_kernel().decode().startswith("conv_r2_kernel") == (d.shape in (6, 10))
        return dx, red
    d32 = Co == 32 and Ci <= 32          # (the data gradient's roles: its Ci is this conv's Co)
    OLDD = (8, 4) if d32 else (3,)
    for ops in ("my", "my+ad+mz", "ad", "none"):
        if Co % 32 != 0:                 # a 16-channel data-gradient K has no instantiation on either kernel family of this test
            break
        dx3, r3 = dgrad("0", OLDD, ops)
        if dx3 is None:
            assert r3 == 9
            break
        dx6, r6 = dgrad("31" if d32 else "3", (10,) if d32 else (6,), ops)
        nd = int((dx6 != dx3).sum())
        assert not torch.isnan(dx6.float()).any() and nd == 0, (ops, nd, float((dx6.float() - dx3.float()).abs().max()), float(dx3.float().abs().max()))
        assert torch.allclose(tot(r6), tot(r3), rtol=1e-5, atol=1e-3), ops


@pytest.mark.parametrize("dt", ["bf16", "mixed", "fp32"])
@pytest.mark.parametrize("ops", ["my", "my+ad+mz", "ad", "none"])
@pytest.mark.parametrize("case", [(4, 48, 36, 32, 32, 3), (2, 12, 9, 256, 256, 3), (2, 24, 18, 64, 256, 1), (4, 24, 18, 128, 64, 3)])
def test_conv_compile_time_epilogue_sets_equal_runtime_form(case, ops, dt, monkeypatch):
    """The data-gradient kernels specialised for an epilogue operand set (EO: mask_y + reductions / all three / addend only /
    none; round 4) against the same launch through the run-time-checked epilogue (STL_CONV_NO_EO=3): bit-identical dx; the
    forward conv with statistics (EO 8) likewise.  Cases: C <= 32 block, wave-specialised 128 x 32 block, 1x1 kernel, C >= 64 block."""
    code, td, tol = DT[dt]
    fcode, ftd = FDT[dt]
    ydt = fcode if fcode != code else 0
    B, H, W, Ci, Co, ks = case
    g = torch.Generator(device="cuda").manual_seed(12)
    x0t = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=g) * 1.5 + 0.3, ftd)
    wt = (torch.randn(Co, ks, ks, Ci, device="cuda", generator=g) / math.sqrt(ks * ks * Ci)).to(ftd)
    g1, b1 = torch.rand(Ci, device="cuda", generator=g) + 0.5, torch.rand(Ci, device="cuda", generator=g) - 0.5
    g2 = torch.rand(Co, device="cuda", generator=g) + 0.5
    st1 = stats_of(x0t, Ci)
    src1 = bn_src(x0t, st1, g1, b1, B * H * W, True)

    def forward():
        yk = torch.full((B * H * W * Co,), float("nan"), device="cuda", dtype=ftd)
        st2 = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device="cuda")
        p = capi.Conv()
        p.dtype, p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co, p.ks, p.stride = fcode, B, H, W, Ci, H, W, Co, ks, 1
        p.TH, p.TW, p.shape = 0, 0, -1
        p.src, p.w, p.out, p.out_stats = src1, wt.data_ptr(), yk.data_ptr(), st2.data_ptr()
        capi.call("stl_conv_forward", C.byref(p), stream())
        torch.cuda.synchronize()
        return yk, st2
    ya, sa = forward()
    monkeypatch.setenv("STL_CONV_NO_EO", "3")
    yb, sb = forward()
    monkeypatch.delenv("STL_CONV_NO_EO")
    assert not torch.isnan(ya.float()).any() and torch.equal(ya, yb)
    assert torch.allclose(sa, sb, rtol=1e-5, atol=1e-3)
    dtt = nhwc(torch.randn(B, Co, H, W, device="cuda", generator=g), td)
    ykf, dtf = ya.view(-1, Co).double(), dtt.view(-1, Co).double()
    mean2, rstd2 = ykf.mean(0), 1.0 / torch.sqrt(ykf.var(0, unbiased=False) + EPS)
    rst2 = torch.zeros(capi.NSHARD, 2, Co, dtype=torch.float64, device="cuda")
    rst2[0, 0], rst2[0, 1] = dtf.sum(0), (dtf * (ykf - mean2) * rstd2).sum(0)
    gs = capi.Src()
    gs.x, gs.y, gs.mode = dtt.data_ptr(), ya.data_ptr(), capi.SRC_BNBWD
    gs.stats, gs.rstats, gs.gamma = sa.data_ptr(), rst2.data_ptr(), g2.data_ptr()
    gs.inv_count, gs.eps = 1.0 / (B * H * W), EPS
    wb = wt.float().view(Co, ks * ks, Ci).flip(1).permute(2, 1, 0).contiguous().to(td)
    addend = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=g), td)
    zmask = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=g), ftd)

    def dgrad():
        dx = torch.full((B * H * W * Ci,), float("nan"), device="cuda", dtype=td)
        red = torch.zeros(capi.NSHARD * 2 * Ci, dtype=torch.float64, device="cuda")
        d = capi.Conv()
        d.dtype, d.B, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co, d.ks, d.stride, d.ydtype = code, B, H, W, Co, H, W, Ci, ks, 1, ydt
        d.TH, d.TW, d.shape = 0, 0, -1
        d.src, d.w, d.out = gs, wb.data_ptr(), dx.data_ptr()
        if "my" in ops:
            d.mask_y, d.mask_bn, d.red = x0t.data_ptr(), bn_src(x0t, st1, g1, b1, B * H * W, "mz" not in ops), red.data_ptr()
        if "ad" in ops:
            d.addend = addend.data_ptr()
        if "mz" in ops:
            d.mask_z = zmask.data_ptr()
        capi.call("stl_conv_forward", C.byref(d), stream())
        torch.cuda.synchronize()
        return dx, red
    dxa, ra = dgrad()
    monkeypatch.setenv("STL_CONV_NO_EO", "3")
    dxb, rb = dgrad()
    assert not torch.isnan(dxa.float()).any() and torch.equal(dxa, dxb)
    assert torch.allclose(ra, rb, rtol=1e-5, atol=1e-3)
    monkeypatch.delenv("STL_CONV_NO_EO")
    if ops == "none" and ks == 3:   # the bias + ReLU forward form of the VGG feature extractors (EO 48)
        bias = torch.randn(Co, device="cuda", generator=g)

        def vggconv():
            yk = torch.full((B * H * W * Co,), float("nan"), device="cuda", dtype=ftd)
            p = capi.Conv()
            p.dtype, p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co, p.ks, p.stride = fcode, B, H, W, Ci, H, W, Co, 3, 1
            p.TH, p.TW, p.shape = 0, 0, -1
            p.src.x, p.src.mode = x0t.data_ptr(), capi.SRC_PLAIN
            p.w, p.out, p.bias, p.out_relu = wt.data_ptr(), yk.data_ptr(), bias.data_ptr(), 1
            capi.call("stl_conv_forward", C.byref(p), stream())
            torch.cuda.synchronize()
            return yk
        va = vggconv()
        monkeypatch.setenv("STL_CONV_NO_EO", "3")
        vb = vggconv()
        assert not torch.isnan(va.float()).any() and torch.equal(va, vb)
        ref = F.relu(F.conv2d(x0t.float().view(B, H, W, Ci).permute(0, 3, 1, 2), wt.float().permute(0, 3, 1, 2), bias, padding=1))
        assert relerr(from_nhwc(va, B, H, W, Co), ref) < (3e-2 if ftd == torch.bfloat16 else (4e-3 if ftd == torch.float16 else 2e-4))


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_fuse_forward_backward_upsample(dt):
    code, td, tol = DT[dt]
    B, H, W, Cc = 2, 16, 12, 32
    g = torch.Generator(device="cuda").manual_seed(3)
    xp = nhwc(torch.randn(B, Cc, H, W, device="cuda", generator=g), td)
    y1 = nhwc(torch.randn(B, Cc, H, W, device="cuda", generator=g) * 2 + 1, td)
    y2 = nhwc(torch.randn(B, Cc, H // 4, W // 4, device="cuda", generator=g), td)
    ga = [torch.rand(Cc, device="cuda", generator=g) + 0.5 for _ in range(2)]
    be = [torch.rand(Cc, device="cuda", generator=g) - 0.5 for _ in range(2)]
    leaf = [t.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True) for t in (xp, y1, y2)]
    t1 = F.batch_norm(leaf[1], None, None, ga[0], be[0], True, 0.1, EPS)
    t2 = F.interpolate(F.batch_norm(leaf[2], None, None, ga[1], be[1], True, 0.1, EPS), scale_factor=4, mode="nearest")
    z = F.relu(t1 + leaf[0] + t2)
    s1, s2 = stats_of(y1, Cc), stats_of(y2, Cc)
    p = capi.Fuse()
    p.dtype, p.B, p.H, p.W, p.C, p.nterms, p.relu = code, B, H, W, Cc, 3, 1
    p.t[0].src = bn_src(y1, s1, ga[0], be[0], B * H * W, False)
    p.t[1].src.x, p.t[1].src.mode = xp.data_ptr(), capi.SRC_PLAIN
    p.t[2].src = bn_src(y2, s2, ga[1], be[1], B * (H // 4) * (W // 4), False)
    p.t[2].shift = 2
    zk = torch.empty(B * H * W * Cc, device="cuda", dtype=td)
    p.out = zk.data_ptr()
    capi.call("stl_fuse_forward", C.byref(p), stream())
    torch.cuda.synchronize()
    assert relerr(from_nhwc(zk, B, H, W, Cc), z.detach()) < tol
    # backward with two gradient contributions
    d1 = nhwc(torch.randn(B, Cc, H, W, device="cuda", generator=g), td)
    d2 = nhwc(torch.randn(B, Cc, H, W, device="cuda", generator=g), td)
    zs = zk.view(B, H, W, Cc).float().permute(0, 3, 1, 2)
    dsum = (d1.float() + d2.float()).permute(0, 3, 1, 2)
    z.backward(dsum)
    q = capi.FuseBwd()
    q.dtype, q.B, q.H, q.W, q.C, q.ngrads, q.relu, q.nbn = code, B, H, W, Cc, 2, 1, 1
    q.dz[0], q.dz[1], q.z = d1.data_ptr(), d2.data_ptr(), zk.data_ptr()
    du = torch.empty_like(zk)
    q.du = du.data_ptr()
    q.bn[0] = p.t[0].src
    r1 = torch.zeros(capi.NSHARD * 2 * Cc, dtype=torch.float64, device="cuda")
    q.rstats[0] = r1.data_ptr()
    capi.call("stl_fuse_backward", C.byref(q), stream())
    u = capi.UpBwd()
    u.dtype, u.B, u.H, u.W, u.C, u.shift = code, B, H // 4, W // 4, Cc, 2
    dtl = torch.empty(B * (H // 4) * (W // 4) * Cc, device="cuda", dtype=td)
    r2 = torch.zeros(capi.NSHARD * 2 * Cc, dtype=torch.float64, device="cuda")
    u.du, u.dt, u.bn, u.rstats = du.data_ptr(), dtl.data_ptr(), p.t[2].src, r2.data_ptr()
    capi.call("stl_upsample_backward", C.byref(u), stream())
    torch.cuda.synchronize()
    du_ref = dsum * (zs > 0)
    assert relerr(from_nhwc(du, B, H, W, Cc), du_ref) < tol          # also d/d(plain x)
    assert relerr(from_nhwc(du, B, H, W, Cc), leaf[0].grad) < tol
    # BN term reductions -> full BN backward == autograd's dL/dy
    for yt, st_, rr, gam, lf, dtk, n in ((y1, s1, r1, ga[0], leaf[1], du, B * H * W),
                                         (y2, s2, r2, ga[1], leaf[2], dtl, B * (H // 4) * (W // 4))):
        yf = yt.view(-1, Cc).double()
        mean, rstd = yf.mean(0), 1.0 / torch.sqrt(yf.var(0, unbiased=False) + EPS)
        r = rr.view(capi.NSHARD, 2, Cc).sum(0)
        dy = (gam.double() * rstd) * (dtk.view(-1, Cc).double() - r[0] / n - (yf - mean) * rstd * r[1] / n)
        ref = lf.grad.permute(0, 2, 3, 1).reshape(-1, Cc).double()
        assert relerr(dy, ref) < tol * 3


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("Cc,nterms,ngrads,nbn", [(32, 1, 1, 0), (32, 2, 3, 1), (64, 2, 4, 3), (48, 2, 2, 2), (256, 2, 1, 1), (32, 4, 2, 1)])
def test_elementwise_sums_every_operand_count(dt, Cc, nterms, ngrads, nbn):
    """Round 5 rewrote the sum kernels so that every operand of an element is requested up front (absent operands alias operand
    0): stl_fuse_forward with one to four terms (upsampled ones among them), stl_fuse_backward with 1 .. 4 gradient
    contributions and 0 .. 3 BatchNorm inputs, stl_upsample_backward for shifts 1 .. 3 -- each against plain torch."""
    code, td, tol = DT[dt]
    B, H, W = 2, 16, 24
    g = torch.Generator(device="cuda").manual_seed(7 + Cc + nterms)
    terms_nchw, p = [], capi.Fuse()
    p.dtype, p.B, p.H, p.W, p.C, p.nterms, p.relu = code, B, H, W, Cc, nterms, 1
    keep, z = [], 0
    for t in range(nterms):
        sh = 0 if nterms <= 2 else t % 4          # the four-term case: shifts 0, 1, 2, 3 (upsampled BatchNorm terms)
        h, w = H >> sh, W >> sh
        x = nhwc(torch.randn(B, Cc, h, w, device="cuda", generator=g) * (1 + t) + 0.3 * t, td)
        xf = x.float().permute(0, 3, 1, 2)
        if t == 0 and nterms > 1:                 # a plain term
            p.t[t].src.x, p.t[t].src.mode = x.data_ptr(), capi.SRC_PLAIN
            v = xf
        else:
            ga, be = torch.rand(Cc, device="cuda", generator=g) + 0.5, torch.rand(Cc, device="cuda", generator=g) - 0.5
            st = stats_of(x, Cc)
            p.t[t].src = bn_src(x, st, ga, be, B * h * w, False)
            v = F.batch_norm(xf, None, None, ga, be, True, 0.1, EPS)
            keep += [ga, be, st]
        p.t[t].shift = sh
        if sh:
            v = F.interpolate(v, scale_factor=1 << sh, mode="nearest")
        z = z + v
        keep.append(x)
    zk = torch.empty(B * H * W * Cc, device="cuda", dtype=td)
    p.out = zk.data_ptr()
    capi.call("stl_fuse_forward", C.byref(p), stream())
    torch.cuda.synchronize()
    assert relerr(from_nhwc(zk, B, H, W, Cc), F.relu(z)) < tol
    # ---- backward: ngrads contributions, ReLU mask, nbn BatchNorm inputs
    ds = [nhwc(torch.randn(B, Cc, H, W, device="cuda", generator=g), td) for _ in range(ngrads)]
    ys = [nhwc(torch.randn(B, Cc, H, W, device="cuda", generator=g) * 1.5 + 0.2, td) for _ in range(nbn)]
    q = capi.FuseBwd()
    q.dtype, q.B, q.H, q.W, q.C, q.ngrads, q.relu, q.nbn = code, B, H, W, Cc, ngrads, 1, nbn
    for i, d in enumerate(ds):
        q.dz[i] = d.data_ptr()
    q.z = zk.data_ptr()
    du = torch.empty_like(zk)
    q.du = du.data_ptr()
    rs, ones = [], torch.ones(Cc, device="cuda")
    for i, y in enumerate(ys):
        sti = stats_of(y, Cc)
        q.bn[i] = bn_src(y, sti, ones, ones, B * H * W, False)
        r = torch.zeros(capi.NSHARD * 2 * Cc, dtype=torch.float64, device="cuda")
        q.rstats[i] = r.data_ptr()
        rs.append(r)
        keep.append(sti)
    capi.call("stl_fuse_backward", C.byref(q), stream())
    torch.cuda.synchronize()
    dsum = sum(d.float() for d in ds) * (zk.view(B, H, W, Cc).float() > 0)
    assert relerr(du.view(B, H, W, Cc).float(), dsum) < tol
    duf = du.float().view(-1, Cc).double()
    for y, r in zip(ys, rs):
        yf = y.float().view(-1, Cc).double()
        mean, rstd = yf.mean(0), 1.0 / torch.sqrt(yf.var(0, unbiased=False) + EPS)
        got = r.view(capi.NSHARD, 2, Cc).sum(0)
        assert relerr(got[0], duf.sum(0)) < tol and relerr(got[1], (duf * (yf - mean) * rstd).sum(0)) < tol * 3
    # ---- upsample backward, shifts 1 .. 3
    for sh in (1, 2, 3):
        h, w = H >> sh, W >> sh
        yb = nhwc(torch.randn(B, Cc, h, w, device="cuda", generator=g), td)
        stb = stats_of(yb, Cc)
        u = capi.UpBwd()
        u.dtype, u.B, u.H, u.W, u.C, u.shift = code, B, h, w, Cc, sh
        dtl = torch.empty(B * h * w * Cc, device="cuda", dtype=td)
        r2 = torch.zeros(capi.NSHARD * 2 * Cc, dtype=torch.float64, device="cuda")
        u.du, u.dt, u.bn, u.rstats = du.data_ptr(), dtl.data_ptr(), bn_src(yb, stb, ones, ones, B * h * w, False), r2.data_ptr()
        capi.call("stl_upsample_backward", C.byref(u), stream())
        torch.cuda.synchronize()
        ref = F.avg_pool2d(from_nhwc(du, B, H, W, Cc), 1 << sh) * float(1 << (2 * sh))
        assert relerr(from_nhwc(dtl, B, h, w, Cc), ref) < tol
        yf = yb.float().view(-1, Cc).double()
        mean, rstd = yf.mean(0), 1.0 / torch.sqrt(yf.var(0, unbiased=False) + EPS)
        dtf = dtl.float().view(-1, Cc).double()
        got = r2.view(capi.NSHARD, 2, Cc).sum(0)
        assert relerr(got[0], dtf.sum(0)) < tol and relerr(got[1], (dtf * (yf - mean) * rstd).sum(0)) < tol * 3


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_fuse_forward_large_two_term_sum(dt):
    """>= 2^21 vectors, two terms of the output's resolution: fuse_flat_big_kernel (register-resident BatchNorm constants, two vectors
    per thread in flight) -- against torch AND bit for bit against fuse_fwd_kernel (a channel count that does not divide the grid's
    thread count cannot take the big kernel, so the same data is run once more as a narrower tensor through the small one)."""
    code, td, tol = DT[dt]
    B, H, W, Cc = 8, 128, 128, 128
    g = torch.Generator(device="cuda").manual_seed(11)
    y = nhwc(torch.randn(B, Cc, H, W, device="cuda", generator=g) * 2 + 0.5, td)
    x = nhwc(torch.randn(B, Cc, H, W, device="cuda", generator=g), td)
    ga, be = torch.rand(Cc, device="cuda", generator=g) + 0.5, torch.rand(Cc, device="cuda", generator=g) - 0.5
    st = stats_of(y, Cc)

    def run(b):
        p = capi.Fuse()
        p.dtype, p.B, p.H, p.W, p.C, p.nterms, p.relu = code, b, H, W, Cc, 2, 1
        p.t[0].src = bn_src(y, st, ga, be, B * H * W, False)
        p.t[1].src.x, p.t[1].src.mode = x.data_ptr(), capi.SRC_PLAIN
        out = torch.empty(b * H * W * Cc, device="cuda", dtype=td)
        p.out = out.data_ptr()
        capi.call("stl_fuse_forward", C.byref(p), stream())
        torch.cuda.synchronize()
        return out
    big = run(B)          # 8 * 128 * 128 * 16 = 2^21 vectors: the big kernel
    small = run(1)        # the first image alone: 2^18 vectors, the general kernel (same statistics, same constants)
    assert torch.equal(big[: H * W * Cc], small)
    ref = F.relu(F.batch_norm(y.float().permute(0, 3, 1, 2), None, None, ga, be, True, 0.1, EPS) + x.float().permute(0, 3, 1, 2))
    assert relerr(from_nhwc(big, B, H, W, Cc), ref) < tol


def test_head_mse_argmax_golden(golden_dir):
    import os
    g6 = np.load(os.path.join(golden_dir, "g6_mse.npz"))
    for name in ("b4", "b1"):
        o = torch.from_numpy(g6[f"{name}_o"]).cuda().requires_grad_(True)
        from stlpose_amd.loss import PersonMSELoss
        l = PersonMSELoss()(o, torch.from_numpy(g6[f"{name}_t"]).cuda(), torch.from_numpy(g6[f"{name}_w"]).cuda())
        l.backward()
        assert abs(l.item() - float(g6[f"{name}_loss"])) < 1e-5 * abs(float(g6[f"{name}_loss"]))
        np.testing.assert_allclose(o.grad.cpu().numpy(), g6[f"{name}_grad"], rtol=1e-4, atol=1e-9)
    g7 = np.load(os.path.join(golden_dir, "g7_decode.npz"))
    from stlpose_amd import pose_parsing
    p, mv = pose_parsing.get_max_preds_hrnet(g7["hm"])
    assert np.array_equal(p, g7["preds"]) and np.array_equal(mv, g7["maxvals"])   # bit-exact incl. ties
    assert pose_parsing.get_max_preds_hrnet(np.zeros((0, 17, 4, 4), np.float32)) == ([], [])
    fp, fmv, coords = pose_parsing.get_final_preds_hrnet(g7["hm"], g7["center"], g7["scale"])
    np.testing.assert_allclose(fp, g7["final_preds"], rtol=1e-5, atol=2e-3)
    np.testing.assert_allclose(coords, g7["final_coords"], atol=1e-3)
    # flip_merge == 0.5 * (a + shift(flip_back(b)))
    a = torch.randn(3, 17, 16, 12, device="cuda")
    hm = torch.from_numpy(g7["hm"]).cuda()
    from stlpose_amd.inference import _perm
    out = torch.empty_like(a)
    capi.call("stl_flip_merge", a.data_ptr(), hm.data_ptr(), out.data_ptr(), _perm(17, "cuda").data_ptr(), 3, 17, 16, 12, stream())
    fb = torch.from_numpy(g7["flip_back"]).cuda()
    sh = fb.clone()
    sh[..., 1:] = fb[..., :-1]
    assert torch.allclose(out, 0.5 * (a + sh), atol=1e-6)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_head_forward_backward(dt):
    code, td, tol = DT[dt]
    B, H, W, Ci, J = 2, 24, 16, 32, 17
    g = torch.Generator(device="cuda").manual_seed(5)
    x = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=g), td)
    w = (torch.randn(J, Ci, 1, 1, device="cuda", generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(J, device="cuda", generator=g).requires_grad_(True)
    xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ref = F.conv2d(xr, w, b)
    out = torch.empty(B, J, H, W, device="cuda")
    capi.call("stl_head_forward", code, x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, Ci, J, stream())
    torch.cuda.synchronize()
    assert relerr(out, ref.detach()) < 1e-5
    do = torch.randn(B, J, H, W, device="cuda", generator=g)
    ref.backward(do)
    nblk, nel = 7, J * Ci + J
    dx = torch.empty(B * H * W * Ci, device="cuda", dtype=td)
    part = torch.full((nblk * nel,), float("nan"), device="cuda")
    capi.call("stl_head_backward", code, x.data_ptr(), w.data_ptr(), do.data_ptr(), dx.data_ptr(), part.data_ptr(), nblk,
              B, H, W, Ci, J, stream())
    torch.cuda.synchronize()
    s = part.view(nblk, nel).sum(0)
    assert relerr(from_nhwc(dx, B, H, W, Ci), xr.grad) < tol
    assert relerr(s[: J * Ci].view(J, Ci), w.grad.view(J, Ci)) < 1e-4
    assert relerr(s[J * Ci:], b.grad) < 1e-4


def test_optimizers_match_torch():
    n = 10007
    g = torch.Generator(device="cuda").manual_seed(7)
    p0 = torch.randn(n, device="cuda", generator=g)
    for kind in ("adam", "sgd"):
        pk = p0.clone()
        pt = p0.clone().requires_grad_(True)
        if kind == "adam":
            opt = torch.optim.Adam([pt], lr=1e-3)
            hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0.0, 0.0, 0.0, 1.0], device="cuda")
        else:
            opt = torch.optim.SGD([pt], lr=1e-2, momentum=0.9, nesterov=True, weight_decay=5e-4)
            hyper = torch.tensor([1e-2, 0.0, 0.0, 0.0, 5e-4, 0.9, 1.0, 1.0], device="cuda")
        m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        for it in range(3):
            gr = torch.randn(n, device="cuda", generator=g)
            pt.grad = gr.clone()
            opt.step()
            if kind == "adam":
                capi.call("stl_adam_step", pk.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), n, hyper.data_ptr(), step.data_ptr(), None, stream())
            else:
                capi.call("stl_sgd_step", pk.data_ptr(), gr.data_ptr(), m.data_ptr(), n, hyper.data_ptr(), step.data_ptr(), None, stream())
        torch.cuda.synchronize()
        assert int(step.item()) == 3
        assert torch.allclose(pk, pt.detach(), rtol=1e-5, atol=1e-6), kind


def test_error_reporting():
    p = capi.Conv()
    p.shape = -1
    p.dtype, p.ks, p.stride = 0, 5, 1
    with pytest.raises(RuntimeError, match="ks must be 1 or 3"):
        capi.call("stl_conv_forward", C.byref(p), stream())
    # round 3: the address arithmetic works with 24-bit multiplies and scalar-unit division; problems beyond its range must be
    # refused with a message, never mis-addressed (no memory is touched before the check: the pointers stay null)
    q = capi.Conv()
    q.dtype, q.ks, q.stride = capi.BF16, 3, 1
    q.B, q.Hi, q.Wi, q.Ci, q.Ho, q.Wo, q.Co = 8, 1500, 1500, 32, 1500, 1500, 32   # 1.8e7 pixels (>= 2^24), 5.8e8 elements (< 2^31)
    q.TH, q.TW, q.shape = 0, 0, -1
    dummy = torch.zeros(64, device="cuda")   # never dereferenced: the range check precedes the launch
    q.src.x, q.w, q.out = dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr()
    with pytest.raises(RuntimeError, match="not addressable|exceeds the fast-division limits|scalar-division limits"):
        capi.call("stl_conv_forward", C.byref(q), stream())


def test_weight_prep_and_reduce_slabs_tables():
    """Table-driven batched helpers: OIHW master -> kernel layouts, and split-K slab reduction."""
    import ctypes
    g = torch.Generator(device="cuda").manual_seed(9)
    shapes = [(8, 16, 3, 0), (16, 8, 1, 0), (64, 3, 3, 1)]   # Co, Ci, ks, patch
    master = torch.randn(sum(co * ci * k * k for co, ci, k, _ in shapes), device="cuda", generator=g)
    tab = (capi.WPrep * len(shapes))()
    so, wo, blk = 0, 0, 0
    offs = []
    for i, (co, ci, k, patch) in enumerate(shapes):
        cip = 32 if patch else ci
        kk = 1 if patch else k * k
        e = tab[i]
        e.src_off, e.fwd_off, e.Co, e.Ci, e.ks, e.Cip, e.patch, e.blk0 = so, wo, co, ci, k, cip, patch, blk
        fwd = wo
        wo += co * kk * cip
        e.bwd_off = -1 if patch else wo
        if not patch:
            wo += co * kk * cip
        offs.append((so, fwd, e.bwd_off))
        so += co * ci * k * k
        blk += math.ceil(co * ci * k * k / 1024)
    wk = torch.zeros(wo, device="cuda")
    tdev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).cuda()
    capi.call("stl_weight_prep", capi.F32, master.data_ptr(), wk.data_ptr(), tdev.data_ptr(), len(shapes), blk, stream())
    torch.cuda.synchronize()
    for (co, ci, k, patch), (s0, f0, b0) in zip(shapes, offs):
        w = master[s0:s0 + co * ci * k * k].view(co, ci, k, k)
        if patch:
            ref = torch.zeros(co, 32, device="cuda")
            ref[:, :27] = w.permute(0, 2, 3, 1).reshape(co, 27)
            assert torch.equal(wk[f0:f0 + co * 32].view(co, 32), ref)
        else:
            assert torch.equal(wk[f0:f0 + co * k * k * ci].view(co, k * k, ci), w.permute(0, 2, 3, 1).reshape(co, k * k, ci))
            refb = w.permute(1, 2, 3, 0).reshape(ci, k * k, co).flip(1)
            assert torch.equal(wk[b0:b0 + co * k * k * ci].view(ci, k * k, co), refb)
    # slabs: sum over splits, back to OIHW
    st = (capi.Slab * len(shapes))()
    po, go, blk = 0, 0, 0
    parts, exp = [], []
    for i, (co, ci, k, patch) in enumerate(shapes):
        ns = 3 + i
        kk, cik = (1, 32) if patch else (k * k, ci)
        p = torch.randn(ns, co, kk, cik, device="cuda", generator=g)
        parts.append(p.reshape(-1))
        s = p.sum(0)
        exp.append((s[:, 0, :27].view(co, 3, 3, 3).permute(0, 3, 1, 2) if patch else s.view(co, k, k, ci).permute(0, 3, 1, 2)).reshape(-1))
        e = st[i]
        e.part_off, e.grad_off, e.nsplit, e.Co, e.Ci, e.ks, e.Cip, e.patch, e.blk0, e.pad = po, go, ns, co, ci, k, cik, patch, blk, 0
        po += p.numel()
        go += co * ci * k * k
        blk += math.ceil(co * ci * k * k / 1024)
    partials = torch.cat(parts)
    grads = torch.full((go,), float("nan"), device="cuda")
    sdev = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).cuda()
    capi.call("stl_reduce_slabs", partials.data_ptr(), grads.data_ptr(), sdev.data_ptr(), len(shapes), blk, stream())
    torch.cuda.synchronize()
    assert torch.allclose(grads, torch.cat(exp), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(2, 24, 16, 32, 32, 3), (2, 12, 10, 64, 128, 1), (3, 16, 12, 128, 64, 3)])
def test_conv_block_end_backward_in_epilogue(case, dt):
    """mask_z: the backward of a residual block end z = ReLU(BN(y) + skip) done in the epilogue of
    the data gradient that yields the last contribution to dz:
        du = (conv(g, w) + addend) * (z > 0),  r1 = sum du,  r2 = sum du * yhat(y)
    -- what stl_fuse_backward computes from a materialised dz (reference HRnet.py:58-59, autograd)."""
    code, td, tol = DT[dt]
    B, H, W, Ci, Co, ks = case
    gen = torch.Generator(device="cuda").manual_seed(11)
    pad = 1 if ks == 3 else 0
    g0 = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=gen), td)
    w = torch.randn(Co, Ci, ks, ks, device="cuda", generator=gen) / math.sqrt(Ci * ks * ks)
    wt = w.permute(0, 2, 3, 1).contiguous().to(td)
    add = nhwc(torch.randn(B, Co, H, W, device="cuda", generator=gen), td)
    z = nhwc(torch.randn(B, Co, H, W, device="cuda", generator=gen).clamp_min(0), td)
    y = nhwc(torch.randn(B, Co, H, W, device="cuda", generator=gen) * 1.3 + 0.2, td)
    gam = torch.rand(Co, device="cuda", generator=gen) + 0.5
    bet = torch.rand(Co, device="cuda", generator=gen) - 0.5
    sty = stats_of(y, Co)
    out = torch.full((B * H * W * Co,), float("nan"), device="cuda", dtype=td)
    red = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device="cuda")
    d = capi.Conv()
    d.shape = -1
    d.dtype, d.B, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co = code, B, H, W, Ci, H, W, Co
    d.ks, d.stride = ks, 1
    d.src.x, d.src.mode = g0.data_ptr(), capi.SRC_PLAIN
    d.w, d.out, d.addend = wt.data_ptr(), out.data_ptr(), add.data_ptr()
    d.mask_z, d.mask_y = z.data_ptr(), y.data_ptr()
    d.mask_bn = bn_src(y, sty, gam, bet, B * H * W, False)
    d.red = red.data_ptr()
    capi.call("stl_conv_forward", C.byref(d), stream())
    torch.cuda.synchronize()
    f = F.conv2d(from_nhwc(g0, B, H, W, Ci), wt.float().permute(0, 3, 1, 2), padding=pad) + from_nhwc(add, B, H, W, Co)
    ref = torch.where(from_nhwc(z, B, H, W, Co) > 0, f, torch.zeros_like(f))
    got = from_nhwc(out, B, H, W, Co)
    assert not torch.isnan(got).any()
    assert relerr(got, ref) < tol
    # reductions are those of the STORED gradient
    du = out.view(-1, Co).double()
    yf = y.view(-1, Co).double()
    yhat = (yf - yf.mean(0)) / torch.sqrt(yf.var(0, unbiased=False) + EPS)
    r = red.view(capi.NSHARD, 2, Co).sum(0)
    assert relerr(r[0], du.sum(0)) < 1e-4
    assert relerr(r[1], (du * yhat).sum(0)) < 1e-3


@pytest.mark.parametrize("sigma,hm,img", [(2.0, (48, 64), (192, 256)), (3.0, (72, 96), (288, 384))])
def test_gaussian_targets_match_oracle(sigma, hm, img):
    """Device target generation vs the oracle restatement of JointsDataset.generate_target
    (data/JointsDataset.py:230-286): joints inside, on the border, far outside, invisible."""
    from oracle import pose_ref
    from stlpose_amd.targets import generate_targets
    rng = np.random.default_rng(4)
    B, J = 6, 17
    joints = (rng.uniform(-40, 1.15 * max(img), size=(B, J, 2)) * 4).round() / 4     # exactly representable in fp32
    joints[0, 0] = [0.0, 0.0]
    joints[0, 1] = [img[0] - 1, img[1] - 1]
    joints[0, 2] = [-1000.0, 50.0]
    joints[0, 3] = [img[0] + 3 * sigma * 4 + 8, 10.0]
    vis = (rng.uniform(size=(B, J)) < 0.8).astype(np.float32)
    tgt, tw = generate_targets(torch.from_numpy(joints.astype(np.float32)), torch.from_numpy(vis), hm, img, sigma)
    torch.cuda.synchronize()
    assert tgt.shape == (B, J, hm[1], hm[0]) and tw.shape == (B, J, 1)
    for b in range(B):
        rt, rw = pose_ref.gaussian_targets(joints[b], vis[b], hm, img, sigma)
        assert np.array_equal(tw[b].cpu().numpy(), rw), b                     # weights are exact
        assert np.abs(tgt[b].cpu().numpy() - rt).max() < 2e-7, b               # expf vs np.exp (f32): < 2 ulp at 1.0
        assert np.array_equal(tgt[b].cpu().numpy() > 0, rt > 0), b           # identical support


def test_gaussian_targets_match_reference_golden(golden_dir):
    """Device target generation vs the reference's own generate_target outputs (fixture G9)."""
    import os
    from stlpose_amd.targets import generate_targets
    g = np.load(os.path.join(golden_dir, "g9_targets.npz"))
    for tag in ("s2", "s3"):
        sigma, wh, hh, wi, hi = g[f"{tag}_cfg"]
        tgt, tw = generate_targets(torch.from_numpy(g[f"{tag}_joints"]), torch.from_numpy(g[f"{tag}_vis"]), (int(wh), int(hh)),
                                   (int(wi), int(hi)), float(sigma))
        torch.cuda.synchronize()
        assert np.array_equal(tw.cpu().numpy(), g[f"{tag}_tw"])
        assert np.abs(tgt.cpu().numpy() - g[f"{tag}_target"]).max() < 2e-7
        assert np.array_equal(tgt.cpu().numpy() > 0, g[f"{tag}_target"] > 0)


@pytest.mark.parametrize("dt", ["bf16", "mixed"])
@pytest.mark.parametrize("case", [(4, 48, 36, 64, 64, 5), (3, 24, 18, 128, 128, 2), (2, 24, 18, 96, 72, 3), (2, 12, 9, 256, 256, 1), (2, 13, 11, 64, 192, 4)])
def test_wgrad_64_channel_blocks_equal_32_channel_blocks(case, dt, monkeypatch):
    """The 3x3 weight gradient on 64 x 64-channel blocks of 16 waves (round 5: every pixel tile staged once per 64 channels, one
    quadrant and all nine taps per wave) against the 32 x 32-channel blocks on the same tensors, tile and split: each slab
    element is the same sum over the same pixels in the same order (K runs over the tile's pixels inside one wave either way),
    so the slabs are BIT-identical.  Ragged channel counts (96, 72) exercise the half-empty blocks."""
    code, td, tol = DT[dt]
    fcode, ftd = FDT[dt]
    ydt = fcode if fcode != code else 0
    B, H, W, Ci, Co, ns = case
    g = torch.Generator(device="cuda").manual_seed(21)
    x0t = nhwc(torch.randn(B, Ci, H, W, device="cuda", generator=g) * 1.5 + 0.3, ftd)
    yk = nhwc(torch.randn(B, Co, H, W, device="cuda", generator=g), ftd)
    dtt = nhwc(torch.randn(B, Co, H, W, device="cuda", generator=g), td)
    g1, b1 = torch.rand(Ci, device="cuda", generator=g) + 0.5, torch.rand(Ci, device="cuda", generator=g) - 0.5
    g2 = torch.rand(Co, device="cuda", generator=g) + 0.5
    st1, st2 = stats_of(x0t, Ci), stats_of(yk, Co)
    ykf, dtf = yk.view(-1, Co).double(), dtt.view(-1, Co).double()
    mean2, rstd2 = ykf.mean(0), 1.0 / torch.sqrt(ykf.var(0, unbiased=False) + EPS)
    rst2 = torch.zeros(capi.NSHARD, 2, Co, dtype=torch.float64, device="cuda")
    rst2[0, 0], rst2[0, 1] = dtf.sum(0), (dtf * (ykf - mean2) * rstd2).sum(0)
    gs = capi.Src()
    gs.x, gs.y, gs.mode = dtt.data_ptr(), yk.data_ptr(), capi.SRC_BNBWD
    gs.stats, gs.rstats, gs.gamma = st2.data_ptr(), rst2.data_ptr(), g2.data_ptr()
    gs.inv_count, gs.eps = 1.0 / (B * H * W), EPS

    def run(c64):
        monkeypatch.setenv("STL_WGRAD_C64", c64)
        wg = capi.Wgrad()
        wg.dtype, wg.B, wg.Hi, wg.Wi, wg.Ci, wg.Ho, wg.Wo, wg.Co = code, B, H, W, Ci, H, W, Co
        wg.ks, wg.stride, wg.ydtype = 3, 1, ydt
        assert capi.lib().stl_wgrad_chunk(C.byref(wg)) == (64 if c64 == "1" else 32)
        wg.TH, wg.TW = choose_tile(B, H, W, 1, 3, 2, bn_cols=32, maxhalo=256)
        wg.nsplit = min(ns, math.ceil(B * (H + 1) / wg.TH) * math.ceil(W / wg.TW))
        part = torch.full((wg.nsplit * Co * 9 * Ci,), float("nan"), device="cuda")
        wg.h, wg.g, wg.partial = bn_src(x0t, st1, g1, b1, B * H * W, True), gs, part.data_ptr()
        capi.call("stl_conv_wgrad", C.byref(wg), stream())
        torch.cuda.synchronize()
        assert ("64>" in capi.lib().stl_last_kernel().decode()) == (c64 == "1"), capi.lib().stl_last_kernel().decode()
        return part
    a, b = run("1"), run("0")
    assert not torch.isnan(a).any() and torch.equal(a, b)


@pytest.mark.parametrize("ws", ["xcd-grid", "plain-grid"])
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(2, 24, 16, 32, 32, 3, 1, 3), (3, 12, 9, 64, 64, 3, 1, 4), (2, 24, 18, 32, 64, 3, 2, 2), (2, 12, 9, 128, 32, 1, 1, 8)])
def test_wgrad_group_equals_single_launches(case, dt, ws, monkeypatch):
    """stl_conv_wgrad_group (several weight gradients of one shape in ONE launch, grid.x = n * nsplit) against n
    stl_conv_wgrad launches on the same tensors: every block does the same work in the same order -> bit-identical
    slabs.  Sources: BN (+ReLU) on h and BatchNorm-backward on g, like the layers the planner groups."""
    B, H, W, Ci, Co, ks, s, n = case
    monkeypatch.setenv("STL_WGRAD_XCD", "0" if ws == "plain-grid" else "1")   # XCD-aware block order (default) / plain 3-D grid
    code, td, _ = DT[dt]
    pad = 1 if ks == 3 else 0
    Ho, Wo = (H + 2 * pad - ks) // s + 1, (W + 2 * pad - ks) // s + 1
    torch.manual_seed(5)
    keep, members = [], []
    TH, TW = choose_tile(B, Ho, Wo, s, ks, 2 if dt == "bf16" else 4, bn_cols=32, maxhalo=576)
    npt = math.ceil(B * (Ho + 1) / TH) * math.ceil(Wo / TW)
    nsplit = min(3, npt)
    nel = Co * ks * ks * Ci
    for i in range(n):
        h = nhwc(torch.randn(B, Ci, H, W, device="cuda"), td)
        y = nhwc(torch.randn(B, Co, Ho, Wo, device="cuda"), td)
        dtt = nhwc(torch.randn(B, Co, Ho, Wo, device="cuda"), td)
        g1, b1 = torch.rand(Ci, device="cuda") + 0.5, torch.randn(Ci, device="cuda") * 0.1
        g2 = torch.rand(Co, device="cuda") + 0.5
        st1, st2 = stats_of(h, Ci), stats_of(y, Co)
        yf = y.float().reshape(-1, Co).double()
        mean2, rstd2 = yf.mean(0), 1.0 / torch.sqrt(yf.var(0, unbiased=False) + EPS)
        df = dtt.float().reshape(-1, Co).double()
        rst2 = torch.zeros(capi.NSHARD, 2, Co, dtype=torch.float64, device="cuda")
        rst2[0, 0], rst2[0, 1] = df.sum(0), (df * (yf - mean2) * rstd2).sum(0)
        gs = capi.Src()
        gs.x, gs.y, gs.mode = dtt.data_ptr(), y.data_ptr(), capi.SRC_BNBWD
        gs.stats, gs.rstats, gs.gamma = st2.data_ptr(), rst2.data_ptr(), g2.data_ptr()
        gs.inv_count, gs.eps = 1.0 / (B * Ho * Wo), EPS
        wg = capi.Wgrad()
        wg.dtype, wg.B, wg.Hi, wg.Wi, wg.Ci, wg.Ho, wg.Wo, wg.Co = code, B, H, W, Ci, Ho, Wo, Co
        wg.ks, wg.stride, wg.TH, wg.TW, wg.nsplit = ks, s, TH, TW, nsplit
        wg.h = bn_src(h, st1, g1, b1, B * H * W, relu=(i % 2 == 0)) if i != 1 else capi.Src()
        if i == 1:   # one member with a PLAIN forward input (block-end sums are plain tensors)
            wg.h.x, wg.h.mode = h.data_ptr(), capi.SRC_PLAIN
        wg.g = gs
        single = torch.full((nsplit * nel,), float("nan"), device="cuda")
        grouped = torch.full((nsplit * nel,), float("nan"), device="cuda")
        wg.partial = single.data_ptr()
        capi.call("stl_conv_wgrad", C.byref(wg), stream())
        torch.cuda.synchronize()
        wg.partial = grouped.data_ptr()
        keep.append((h, y, dtt, g1, b1, g2, st1, st2, rst2))
        members.append((wg, single, grouped))
    grp = capi.WgradGroup()
    grp.n = n
    for i, (wg, _s, _g) in enumerate(members):
        grp.p[i] = C.pointer(wg)
    capi.call("stl_conv_wgrad_group", C.byref(grp), stream())
    torch.cuda.synchronize()
    for i, (_wg, single, grouped) in enumerate(members):
        assert not torch.isnan(single).any() and torch.equal(single, grouped), f"member {i} differs from its stand-alone launch"
    # a member of another shape is refused
    members[-1][0].Co += 8
    assert capi.lib().stl_conv_wgrad_group(C.byref(grp), stream()) != 0


@pytest.mark.parametrize("dt", ["fp32", "bf16", "f16"])
@pytest.mark.parametrize("case", [(4, 48, 36, 32, 32), (4, 24, 18, 64, 64), (3, 24, 18, 128, 128), (2, 12, 16, 16, 16), (2, 10, 4, 48, 48), (2, 13, 7, 32, 64),
                                  (2, 24, 18, 256, 64, 1), (3, 13, 11, 128, 192, 1), (2, 12, 9, 256, 256)])
def test_conv_block_end_source_bnadd(case, dt):
    """STL_SRC_BNADD: z = ReLU(BN(y) + skip) formed while the conv stages its input (3x3, or the streaming 1x1 kernel: layer1's
    bottleneck units, round 5), and written out once (src_out) -- against torch: conv2d(relu(batch_norm(y) + skip)) and the sum
    itself; and bit-identical to the two-launch form (stl_fuse_forward, then the conv on the plain sum)."""
    B, H, W, Ci, Co = case[:5]
    ks = case[5] if len(case) > 5 else 3
    code, td, tol = DT[dt]
    torch.manual_seed(11)
    y = torch.randn(B, Ci, H, W, device="cuda") * 1.5 + 0.3
    skip = torch.relu(torch.randn(B, Ci, H, W, device="cuda"))
    g, b = torch.rand(Ci, device="cuda") + 0.5, torch.randn(Ci, device="cuda") * 0.2
    w = torch.randn(Co, Ci, ks, ks, device="cuda") * (2.0 / (Ci * ks * ks)) ** 0.5
    yt, st_, wt = nhwc(y, td), nhwc(skip, td), w.permute(0, 2, 3, 1).contiguous().to(td)     # [Co][tap][Ci]
    stats = stats_of(yt, Ci)
    zr = torch.relu(F.batch_norm(from_nhwc(yt, B, H, W, Ci), None, None, g, b, True, 0.0, EPS) + from_nhwc(st_, B, H, W, Ci))
    ref = F.conv2d(from_nhwc(nhwc(zr, td), B, H, W, Ci), wt.float().permute(0, 3, 1, 2), padding=ks // 2)
    out = torch.full((B * H * W * Co,), float("nan"), device="cuda", dtype=td)
    z = torch.full((B * H * W * Ci,), float("nan"), device="cuda", dtype=td)
    p = capi.Conv()
    p.shape = -1
    p.dtype, p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co, p.ks, p.stride = code, B, H, W, Ci, H, W, Co, ks, 1
    capi.call("stl_conv_plan", C.byref(p))
    if capi.lib().stl_conv_bnadd_ok(C.byref(p)) != 1:
        pytest.skip(f"block shape {p.shape} has no block-end variant (the planner keeps the separate sum launch there)")
    p.src = bn_src(yt, stats, g, b, B * H * W, True)
    p.src.mode, p.src.y = capi.SRC_BNADD, st_.data_ptr()
    p.src_out, p.w, p.out = z.data_ptr(), wt.data_ptr(), out.data_ptr()
    ost = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device="cuda")
    p.out_stats = ost.data_ptr()
    capi.call("stl_conv_forward", C.byref(p), stream())
    torch.cuda.synchronize()
    assert not torch.isnan(z.float()).any() and not torch.isnan(out.float()).any()
    assert relerr(from_nhwc(z, B, H, W, Ci), zr) < tol
    assert relerr(from_nhwc(out, B, H, W, Co), ref) < tol
    # two-launch form: the sum kernel, then the same conv on the plain tensor
    f = capi.Fuse()
    f.dtype, f.B, f.H, f.W, f.C, f.nterms, f.relu = code, B, H, W, Ci, 2, 1
    f.t[0].src = bn_src(yt, stats, g, b, B * H * W, False)
    f.t[1].src.x, f.t[1].src.mode = st_.data_ptr(), capi.SRC_PLAIN
    z2 = torch.empty_like(z)
    f.out = z2.data_ptr()
    capi.call("stl_fuse_forward", C.byref(f), stream())
    q = capi.Conv()
    q.shape = -1
    q.dtype, q.B, q.Hi, q.Wi, q.Ci, q.Ho, q.Wo, q.Co, q.ks, q.stride = code, B, H, W, Ci, H, W, Co, ks, 1
    q.TH, q.TW, q.shape = p.TH, p.TW, p.shape
    q.src.x, q.src.mode = z2.data_ptr(), capi.SRC_PLAIN
    out2 = torch.empty_like(out)
    ost2 = torch.zeros_like(ost)
    q.w, q.out, q.out_stats = wt.data_ptr(), out2.data_ptr(), ost2.data_ptr()
    capi.call("stl_conv_forward", C.byref(q), stream())
    torch.cuda.synchronize()
    assert torch.equal(z, z2), "block-end sum differs from the sum kernel's"
    assert torch.equal(out, out2), "conv over the merged source differs from conv over the materialised sum"
    torch.testing.assert_close(ost.view(capi.NSHARD, -1).sum(0), ost2.view(capi.NSHARD, -1).sum(0), rtol=1e-12, atol=1e-9)
