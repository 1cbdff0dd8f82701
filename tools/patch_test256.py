p='/root/repo/tests/test_ops_gpu.py'
s=open(p).read()
a='''@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(2, 24, 16, 32, 32, 3, 1), (2, 24, 18, 32, 64, 3, 2), (2, 8, 6, 128, 64, 1, 1),
                                  (4, 48, 36, 64, 64, 3, 1), (4, 24, 18, 128, 128, 3, 1)])
def test_conv_bn_relu_chain_forward_backward(case, dt):'''
b='''@pytest.mark.parametrize("wtile", [128, 256])
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(2, 24, 16, 32, 32, 3, 1), (2, 24, 18, 32, 64, 3, 2), (2, 8, 6, 128, 64, 1, 1),
                                  (4, 48, 36, 64, 64, 3, 1), (4, 24, 18, 128, 128, 3, 1)])
def test_conv_bn_relu_chain_forward_backward(case, dt, wtile):'''
assert a in s
s=s.replace(a,b)
a='''    B, H, W, Ci, Co, ks, s = case
    g = torch.Generator(device="cuda").manual_seed(2)
    pad = 1 if ks == 3 else 0
    x0 = torch.randn(B, Ci, H, W, device="cuda", generator=g) * 1.5 + 0.3'''
b='''    B, H, W, Ci, Co, ks, s = case
    if wtile == 256 and (dt != "bf16" or s != 1):
        pytest.skip("256-pixel weight-gradient tiles are bf16, stride 1 only")
    g = torch.Generator(device="cuda").manual_seed(2)
    pad = 1 if ks == 3 else 0
    x0 = torch.randn(B, Ci, H, W, device="cuda", generator=g) * 1.5 + 0.3'''
assert a in s
s=s.replace(a,b)
a='''    wg.TH, wg.TW = choose_tile(B, Ho, Wo, s, ks, x0t.element_size(), bn_cols=32)
    npt = math.ceil(B * (Ho + 1) / wg.TH) * math.ceil(Wo / wg.TW)
    wg.nsplit = min(3, npt)'''
b='''    if wtile == 256:
        wg.TH, wg.TW = choose_tile(B, Ho, Wo, s, ks, x0t.element_size(), bn_cols=32, maxpx=256, maxhalo=384)
        assert wg.TH * wg.TW > 128
    else:
        wg.TH, wg.TW = choose_tile(B, Ho, Wo, s, ks, x0t.element_size(), bn_cols=32)
    npt = math.ceil(B * (Ho + 1) / wg.TH) * math.ceil(Wo / wg.TW)
    wg.nsplit = min(3, npt)'''
assert a in s
s=s.replace(a,b)
open(p,'w').write(s)
