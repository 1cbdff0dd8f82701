def rep(path, a, b, count=1):
    s = open(path).read()
    assert s.count(a) == count, (path, s.count(a), a)
    open(path, 'w').write(s.replace(a, b))

H = '/root/repo/include/stlpose_hip.h'
rep(H, '''    double* red;           /* [NSHARD][2*Co] += (r1,r2) against mask_y or NULL     */
} stl_conv;''', '''    double* red;           /* [NSHARD][2*Co] += (r1,r2) against mask_y or NULL     */
    const void* mask_z;    /* [B,Ho,Wo,Co] dtype or NULL: ReLU output z whose sign gates `out` (out = 0 where
                              z <= 0).  With addend / mask_y (mask_bn.relu = 0) / red this is the backward of a
                              residual block end  z = ReLU(BN(y) + x)  fused into the data gradient that
                              produces the last contribution to dz (HRnet.py:58-59,99-100). */
} stl_conv;''')
rep('/root/repo/stlpose_amd/capi.py', '''("addend", vp), ("mask_y", vp), ("mask_bn", Src), ("red", vp)]''', '''("addend", vp), ("mask_y", vp), ("mask_bn", Src), ("red", vp), ("mask_z", vp)]''')

I = '/root/repo/stlpose_amd/csrc/conv_common.inc'
rep(I, '''    f2v ad[NTW][2], my[NTW][2];
    if (p.addend) {''', '''    f2v ad[NTW][2], my[NTW][2], mz[NTW][2];
    if (p.mask_z) {
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) load4p<T>(p.mask_z, eov[ni], mz[ni][0], mz[ni][1]);
    }
    if (p.addend) {''')
rep(I, '''        if (p.out_relu) {
#pragma unroll
            for (int h = 0; h < 2; ++h) f[h][0] = fmaxf(f[h][0], 0.f), f[h][1] = fmaxf(f[h][1], 0.f);
        }''', '''        if (p.mask_z) {  // gradient of z = ReLU(...): passes where z > 0
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f[h][0] = mz[ni][h][0] > 0.f ? f[h][0] : 0.f;
                f[h][1] = mz[ni][h][1] > 0.f ? f[h][1] : 0.f;
            }
        }
        if (p.out_relu) {
#pragma unroll
            for (int h = 0; h < 2; ++h) f[h][0] = fmaxf(f[h][0], 0.f), f[h][1] = fmaxf(f[h][1], 0.f);
        }''')
