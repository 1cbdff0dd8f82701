"""Diagnostic: fp32-path accuracy of the unit ops against fp64 references (where do the ~1e-3 gradient errors of the
fp32 network come from?)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from stlpose_amd import capi
from stlpose_amd.engine import choose_tile
from tests.test_ops_gpu import nhwc, from_nhwc, stats_of, bn_src, stream, EPS

def rel(a, b): return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))

def run(B, H, W, Ci, Co, ks, s, xmean):
    code, td = capi.F32, torch.float32
    g = torch.Generator(device="cuda").manual_seed(2)
    pad = 1 if ks == 3 else 0
    x0 = torch.randn(B, Ci, H, W, device="cuda", generator=g) * 1.5 + xmean
    x0t = nhwc(x0, td)
    x0r = x0t.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    w = (torch.randn(Co, Ci, ks, ks, device="cuda", generator=g) / math.sqrt(Ci * ks * ks))
    wt = w.permute(0, 2, 3, 1).contiguous().to(td)
    wr = wt.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    g1 = (torch.rand(Ci, device="cuda", generator=g) + 0.5); b1 = (torch.rand(Ci, device="cuda", generator=g) - 0.5)
    g2 = (torch.rand(Co, device="cuda", generator=g) + 0.5); b2 = (torch.rand(Co, device="cuda", generator=g) - 0.5)
    g1d, b1d, g2d, b2d = [t.double().requires_grad_(True) for t in (g1, b1, g2, b2)]
    h = F.relu(F.batch_norm(x0r, None, None, g1d, b1d, True, 0.1, EPS))
    y = F.conv2d(h, wr, stride=s, padding=pad)
    t = F.batch_norm(y, None, None, g2d, b2d, True, 0.1, EPS)
    Ho, Wo = y.shape[2:]
    dt_up = torch.randn(B, Co, Ho, Wo, device="cuda", generator=g)
    dtt = nhwc(dt_up, td)
    t.backward(dtt.double().permute(0, 3, 1, 2))
    # same graph in torch fp32 (what the oracle does)
    x32 = x0t.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True); w32 = wt.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    g1f, b1f, g2f, b2f = [t_.clone().requires_grad_(True) for t_ in (g1, b1, g2, b2)]
    t32 = F.batch_norm(F.conv2d(F.relu(F.batch_norm(x32, None, None, g1f, b1f, True, 0.1, EPS)), w32, stride=s, padding=pad), None, None, g2f, b2f, True, 0.1, EPS)
    t32.backward(dtt.float().permute(0, 3, 1, 2))
    # HIP forward
    st1 = stats_of(x0t, Ci)
    yk = torch.empty(B * Ho * Wo * Co, device="cuda", dtype=td)
    st2 = torch.zeros(capi.NSHARD * 2 * Co, dtype=torch.float64, device="cuda")
    p = capi.Conv(); p.shape = -1
    p.dtype, p.B, p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.Co = code, B, H, W, Ci, Ho, Wo, Co
    p.ks, p.stride, p.TH, p.TW = ks, s, 0, 0
    p.src = bn_src(x0t, st1, g1, b1, B * H * W, True)
    p.w, p.out, p.out_stats = wt.data_ptr(), yk.data_ptr(), st2.data_ptr()
    capi.call("stl_conv_forward", C.byref(p), stream()); torch.cuda.synchronize()
    e_y = rel(from_nhwc(yk, B, Ho, Wo, Co), y.detach())
    ykf = yk.view(-1, Co).double(); mean2 = ykf.mean(0); rstd2 = 1.0 / torch.sqrt(ykf.var(0, unbiased=False) + EPS)
    dtf = dtt.view(-1, Co).double()
    rst2 = torch.zeros(capi.NSHARD, 2, Co, dtype=torch.float64, device="cuda")
    rst2[0, 0] = dtf.sum(0); rst2[0, 1] = (dtf * (ykf - mean2) * rstd2).sum(0)
    gs = capi.Src(); gs.x, gs.y, gs.mode = dtt.data_ptr(), yk.data_ptr(), capi.SRC_BNBWD
    gs.stats, gs.rstats, gs.gamma = st2.data_ptr(), rst2.data_ptr(), g2.data_ptr()
    gs.inv_count, gs.eps = 1.0 / (B * Ho * Wo), EPS
    wg = capi.Wgrad()
    wg.dtype, wg.B, wg.Hi, wg.Wi, wg.Ci, wg.Ho, wg.Wo, wg.Co, wg.ks, wg.stride = code, B, H, W, Ci, Ho, Wo, Co, ks, s
    wg.TH, wg.TW = choose_tile(B, Ho, Wo, s, ks, 4, bn_cols=32, maxhalo=576)
    npt = math.ceil(B * (Ho + 1) / wg.TH) * math.ceil(Wo / wg.TW); wg.nsplit = min(3, npt)
    part = torch.zeros(wg.nsplit * Co * ks * ks * Ci, device="cuda")
    wg.h, wg.g, wg.partial = p.src, gs, part.data_ptr()
    capi.call("stl_conv_wgrad", C.byref(wg), stream()); torch.cuda.synchronize()
    dw = part.view(wg.nsplit, Co, ks * ks, Ci).double().sum(0).view(Co, ks, ks, Ci).permute(0, 3, 1, 2)
    wb = wt.float().view(Co, ks * ks, Ci).flip(1).permute(2, 1, 0).contiguous()
    dx = torch.zeros(B * H * W * Ci, device="cuda"); red = torch.zeros(capi.NSHARD * 2 * Ci, dtype=torch.float64, device="cuda")
    d = capi.Conv(); d.shape = -1
    d.dtype, d.B, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co = code, B, Ho, Wo, Co, H, W, Ci
    d.ks, d.stride, d.stuff, d.TH, d.TW = ks, 1, int(s == 2), 0, 0
    d.src, d.w, d.out = gs, wb.data_ptr(), dx.data_ptr()
    d.mask_y, d.mask_bn, d.red = x0t.data_ptr(), p.src, red.data_ptr()
    capi.call("stl_conv_forward", C.byref(d), stream()); torch.cuda.synchronize()
    x0f = x0t.view(-1, Ci).double(); mean1, rstd1 = x0f.mean(0), 1.0 / torch.sqrt(x0f.var(0, unbiased=False) + EPS)
    xhat = (x0f - mean1) * rstd1
    dxk = dx.view(-1, Ci).double(); r = red.view(capi.NSHARD, 2, Ci).sum(0); n = B * H * W
    dx0 = (g1.double() * rstd1) * (dxk - r[0] / n - xhat * r[1] / n)
    ref = x0r.grad.permute(0, 2, 3, 1).reshape(-1, Ci)
    ref32 = x32.grad.permute(0, 2, 3, 1).reshape(-1, Ci)
    print(f"B{B} {H}x{W} {Ci}->{Co} k{ks}s{s} xmean {xmean}: HIP vs f64: y {e_y:.1e} dw {rel(dw, wr.grad):.1e} dbeta1 {rel(r[0], b1d.grad):.1e} dgamma1 {rel(r[1], g1d.grad):.1e} dx0 {rel(dx0, ref):.1e}"
          f" | torch32 vs f64: dw {rel(w32.grad, wr.grad):.1e} dbeta1 {rel(b1f.grad, b1d.grad):.1e} dgamma1 {rel(g1f.grad, g1d.grad):.1e} dx0 {rel(ref32, ref):.1e}", flush=True)

for xm in (0.3, 5.0):
    run(2, 24, 16, 32, 32, 3, 1, xm)
    run(4, 48, 36, 64, 64, 3, 1, xm)
    run(4, 24, 18, 128, 128, 3, 1, xm)
    run(2, 24, 18, 64, 256, 1, 1, xm)
