p='/root/repo/stlpose_amd/csrc/conv_core.hip'
s=open(p).read()
a=s.index("    // block shape and pixel tile: p.TH/p.TW > 0 force")
b=s.index("    ConvK k;\n    k.p = p;")
new='''    // block shape and pixel tile: planned once by stl_conv_plan (shape >= 0), else searched here
    Plan plan;
    if (p.shape >= 0 && p.shape < 4 && p.TH > 0 && p.TW > 0) {
        STL_CHECK(p.TH * p.TW <= SHAPES[p.shape].px, "conv: tile %dx%d exceeds block shape %d", p.TH, p.TW, p.shape);
        plan = Plan{p.shape, p.TH, p.TW, 0, 0.0};
    } else if (p.TH > 0 && p.TW > 0) {  // explicit 128-pixel tile (tests)
        STL_CHECK(p.TH * p.TW <= 128, "conv: explicit tile %dx%d exceeds 128 pixels", p.TH, p.TW);
        plan = Plan{0, p.TH, p.TW, 0, 0.0};
    } else {
        plan = choose_plan(p, ck);
        STL_CHECK(plan.shape >= 0, "conv: no tile fits LDS for %dx%d ks %d stride %d Ci %d", p.Ho, p.Wo, p.ks, p.stride, p.Ci);
    }
    {
        const Shape shp = SHAPES[plan.shape];
        const int nv = ceil_div(((plan.TH - 1) * p.stride + p.ks) * ((plan.TW - 1) * p.stride + p.ks) * 4, shp.thr);
        STL_CHECK(nv <= (plan.shape == 0 ? 9 : plan.shape == 1 ? 6 : 3), "conv: tile %dx%d has too large a halo for block shape %d", plan.TH, plan.TW, plan.shape);
    }
'''
s=s[:a]+new+s[b:]
s=s.replace('extern "C" int stl_conv_forward(const stl_conv* pp, void* stream) {','''extern "C" int stl_conv_plan(stl_conv* pp) {
    stl_conv& p = *pp;
    STL_CHECK(p.dtype == STL_F32 || p.dtype == STL_BF16, "conv_plan: bad dtype");
    STL_CHECK((p.ks == 1 || p.ks == 3) && (p.stride == 1 || p.stride == 2) && p.Ci > 0 && p.Co > 0, "conv_plan: bad geometry");
    const int ck = p.dtype == STL_BF16 ? 32 : 16;
    Plan plan = choose_plan(p, ck);
    if (const char* e = getenv("STL_CONV_SHAPE")) {  // tuning knob: force a block shape where legal
        const int f = atoi(e);
        if (f >= 0 && f < 4 && !(f != 0 && p.stride == 2)) {
            Plan best{-1, 0, 0, 0, 1e300};
            const int vrows = p.B * (p.Ho + 1);
            const Shape sh = SHAPES[f];
            for (int tw = (p.Wo < 4 ? p.Wo : 4); tw <= p.Wo && tw <= sh.px; ++tw) {
                int th = sh.px / tw;
                if (th > vrows) th = vrows;
                const int hr = (th - 1) * p.stride + p.ks, hc = (tw - 1) * p.stride + p.ks;
                const int nva = ceil_div(hr * hc * 4, sh.thr);
                if ((f == 0 && nva > 9) || (f == 1 && nva > 6) || (f >= 2 && nva > 3)) continue;
                const size_t l = lds_bytes(p, f, th, tw, ck, nullptr);
                if (l > 158 * 1024) continue;
                const double waste = (double)ceil_div(vrows, th) * th * ceil_div(p.Wo, tw) * tw * (double)hr * hc / (th * tw);
                if (waste < best.cost) best = Plan{f, th, tw, l, waste};
            }
            if (best.shape >= 0) plan = best;
        }
    }
    STL_CHECK(plan.shape >= 0, "conv_plan: no tile fits LDS for %dx%d ks %d stride %d Ci %d", p.Ho, p.Wo, p.ks, p.stride, p.Ci);
    p.shape = plan.shape, p.TH = plan.TH, p.TW = plan.TW;
    return 0;
}

extern "C" int stl_conv_forward(const stl_conv* pp, void* stream) {''')
open(p,'w').write(s)
p='/root/repo/stlpose_amd/engine.py'
s=open(p).read()
s=s.replace("        p.TH, p.TW = 0, 0  # block shape and pixel tile chosen by the launcher\n","        p.TH, p.TW, p.shape = 0, 0, -1\n        capi.call(\"stl_conv_plan\", C.byref(p))  # block shape + pixel tile, searched once\n")
s=s.replace("                d.TH, d.TW = 0, 0\n","                d.TH, d.TW, d.shape = 0, 0, -1\n                capi.call(\"stl_conv_plan\", C.byref(d))\n")
open(p,'w').write(s)
for p in ['/root/repo/tests/test_ops_gpu.py','/root/repo/tools/conv_probe.py','/root/repo/tools/conv_probe3.py']:
    s=open(p).read()
    s=s.replace("    p = capi.Conv()\n","    p = capi.Conv()\n    p.shape = -1\n").replace("    d = capi.Conv()\n","    d = capi.Conv()\n    d.shape = -1\n")
    open(p,'w').write(s)
p='/root/repo/tools/conv_probe.py'
s=open(p).read()
s=s.replace('    stream = torch.cuda.current_stream().cuda_stream\n    for _ in range(3):','    capi.call("stl_conv_plan", C.byref(p))\n    stream = torch.cuda.current_stream().cuda_stream\n    for _ in range(3):')
open(p,'w').write(s)
print("ok")
