p='/root/repo/stlpose_amd/csrc/wgrad.hip'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, (s.count(a), a)
    s=s.replace(a,b)
rep('''    int off_cg, off_ch, off_g, off_h;
};''','''    int off_cg, off_ch, off_g, off_h;
    float r_TW, r_HC, r_tc, r_vp, r_PI;  // reciprocals for fdiv
};

// floor(m / d) for 0 <= m < 2^21 with r = 1/d (d <= 2^10): exact, ~4 instructions instead of ~40
__device__ __forceinline__ int fdiv(int m, float r) { return (int)(((float)m + 0.5f) * r); }''')
rep('''    if (tid < 32) {
        float a = 0.f, b = 0.f, c = 0.f;
        if (co0 + tid < p.Co) src_consts(p.g, co0 + tid, p.Co, a, b, c);
        cgc[tid] = a, cgc[32 + tid] = b, cgc[64 + tid] = c;
    } else if (tid < 64) {
        const int t = tid - 32;
        float a = 0.f, b = 0.f, c = 0.f;
        if (ci0 + t < p.Ci) src_consts(p.h, ci0 + t, p.Ci, a, b, c);
        chc[t] = a, chc[32 + t] = b;
    }

''','')
rep('''            const int ty = m / p.TW;
            g_yx[i] = (ty << 16) | (m - ty * p.TW);''','''            const int ty = fdiv(m, k.r_TW);
            g_yx[i] = (ty << 16) | (m - ty * p.TW);''')
rep('''            const int hp = v / VPX, hr = hp / k.HC;''','''            const int hp = v / VPX, hr = fdiv(hp, k.r_HC);''')
rep('''            const int ty = m / p.TW, tx = m - ty * p.TW;
            rh[s][i]''','''            const int ty = fdiv(m, k.r_TW), tx = m - ty * p.TW;
            rh[s][i]''')
rep('''        const int tr = t / k.tiles_c, tc = t - tr * k.tiles_c;
        const int vr0 = tr * p.TH, c0 = tc * p.TW;
        const int gb0 = vr0 / vpitch, gy0 = vr0 - gb0 * vpitch;''','''        const int tr = fdiv(t, k.r_tc), tc = t - tr * k.tiles_c;
        const int vr0 = tr * p.TH, c0 = tc * p.TW;
        const int gb0 = fdiv(vr0, k.r_vp), gy0 = vr0 - gb0 * vpitch;''')
rep('''        const int hb0 = vrs / k.PI, hy0 = vrs - hb0 * k.PI - k.pad;''','''        const int hb0 = fdiv(vrs, k.r_PI), hy0 = vrs - hb0 * k.PI - k.pad;''')
rep('''    issue(have);
    WSTAMP(2);
    __syncthreads();  // constants visible''','''    issue(have);
    WSTAMP(2);
    // BatchNorm constants: wave 3, after its loads are in flight
    if (tid >= 192 && tid < 224) {
        const int c = tid - 192;
        float a = 0.f, b = 0.f, cc = 0.f;
        if (co0 + c < p.Co) src_consts(p.g, co0 + c, p.Co, a, b, cc);
        cgc[c] = a, cgc[32 + c] = b, cgc[64 + c] = cc;
    } else if (tid >= 224) {
        const int c = tid - 224;
        float a = 0.f, b = 0.f, cc = 0.f;
        if (ci0 + c < p.Ci) src_consts(p.h, ci0 + c, p.Ci, a, b, cc);
        chc[c] = a, chc[32 + c] = b;
    }
    __syncthreads();  // constants visible''')
rep('''    k.npt = ceil_div(p.B * (p.Ho + 1), p.TH) * k.tiles_c;''','''    k.npt = ceil_div(p.B * (p.Ho + 1), p.TH) * k.tiles_c;
    k.r_TW = 1.0f / p.TW, k.r_HC = 1.0f / k.HC, k.r_tc = 1.0f / k.tiles_c, k.r_vp = 1.0f / (p.Ho + 1), k.r_PI = 1.0f / k.PI;
    STL_CHECK((int64_t)k.npt < (1 << 21) && (int64_t)p.B * k.PI < (1 << 21), "wgrad: too many tiles");''')
open(p,'w').write(s)
