root='/root/repo/stlpose_amd/csrc/'
s=open(root+'conv_core.hip').read()
i0=s.index('    // ---- per-channel constants, phase 1: only ISSUE')
i1=s.index('    STAMP(1);')
new='''    // ---- per-channel constants
    for (int c = tid; c < k.cipad; c += NTHR) {
        float a = 0.f, b = 0.f, cc = 0.f;
        if (c < p.Ci) src_consts(p.src, c, p.Ci, a, b, cc);
        cs[c] = a, cs[k.cipad + c] = b, cs[2 * k.cipad + c] = cc;
    }
    if (p.mask_y) {
        for (int c = tid; c < BCO; c += NTHR) {
            float a = 0.f, b = 0.f, mu = 0.f, rs = 0.f;
            if (n0 + c < p.Co) {
                bn_mean_rstd(p.mask_bn, n0 + c, p.Co, mu, rs);
                a = p.mask_bn.gamma[n0 + c] * rs;
                b = p.mask_bn.beta[n0 + c] - mu * a;
            }
            cm[c] = a, cm[BCO + c] = b, cm[2 * BCO + c] = mu, cm[3 * BCO + c] = rs;
        }
    }

'''
s=s[:i0]+new+s[i1:]
assert s.count('    consts_finish();\n')==1
s=s.replace('    consts_finish();\n','')
open(root+'conv_core.hip','w').write(s)
