import re
root='/root/repo/stlpose_amd/csrc/'
def load(f): return open(root+f).read()
def save(f,s): open(root+f,'w').write(s)
def rep(s,a,b,count=1):
    assert s.count(a)==count, (s.count(a), a)
    return s.replace(a,b)

# ---------------- common.cuh: fdiv + restructured raw helpers
s=load('common.cuh')
i0=s.index('// Two-phase variant: src_raw_load only ISSUES')
i1=s.index('// block-wide helpers ------')
new='''// floor(m / d) for 0 <= m, m * d < 2^21, with r = 1/d: exact, ~4 instructions instead of ~40
__device__ __forceinline__ int fdiv(int m, float r) { return (int)(((float)m + 0.5f) * r); }

// Two-phase variants: *_raw_load only ISSUES the loads of one channel's statistics (so that they
// queue ahead of a burst of tile loads), *_raw_finish does the arithmetic of bn_mean_rstd / src_consts.
struct SrcRaw {
    double st[2 * STL_NSHARD], rs[2 * STL_NSHARD];
    float g, b, rm, rv;
};
__device__ __forceinline__ void bn_raw_load(const stl_src& s, int c, int C, SrcRaw& r) {
    r.g = s.gamma[c];
    if (s.stats) {
#pragma unroll
        for (int k = 0; k < STL_NSHARD; ++k) {
            r.st[2 * k] = s.stats[(size_t)k * 2 * C + c];
            r.st[2 * k + 1] = s.stats[(size_t)k * 2 * C + C + c];
        }
    } else {
        r.rm = s.rmean[c], r.rv = s.rvar[c];
    }
}
__device__ __forceinline__ void bn_raw_finish(const stl_src& s, const SrcRaw& r, float& mean, float& rstd) {
    if (s.stats) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < STL_NSHARD; ++k) s0 += r.st[2 * k], s1 += r.st[2 * k + 1];
        double m = s0 * (double)s.inv_count;
        double var = s1 * (double)s.inv_count - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)s.eps));
    } else {
        mean = r.rm;
        rstd = (float)(1.0 / sqrt((double)r.rv + (double)s.eps));
    }
}
__device__ __forceinline__ void src_raw_load(const stl_src& s, int c, int C, SrcRaw& r) {
    if (s.mode == STL_SRC_PLAIN) return;
    bn_raw_load(s, c, C, r);
    if (s.mode == STL_SRC_BN) {
        r.b = s.beta[c];
    } else {
#pragma unroll
        for (int k = 0; k < STL_NSHARD; ++k) {
            r.rs[2 * k] = s.rstats[(size_t)k * 2 * C + c];
            r.rs[2 * k + 1] = s.rstats[(size_t)k * 2 * C + C + c];
        }
    }
}
__device__ __forceinline__ void src_raw_finish(const stl_src& s, const SrcRaw& r, float& ca, float& cb, float& cc) {
    if (s.mode == STL_SRC_PLAIN) {
        ca = 1.f, cb = 0.f, cc = 0.f;
        return;
    }
    float mean, rstd;
    bn_raw_finish(s, r, mean, rstd);
    if (s.mode == STL_SRC_BN) {
        ca = r.g * rstd;
        cb = r.b - mean * ca;
        cc = 0.f;
    } else {
        double r1 = 0.0, r2 = 0.0;
#pragma unroll
        for (int k = 0; k < STL_NSHARD; ++k) r1 += r.rs[2 * k], r2 += r.rs[2 * k + 1];
        const float c1 = (float)(r1 * (double)s.inv_count);
        const float c2 = (float)(r2 * (double)s.inv_count);
        const float al = r.g * rstd;
        ca = al;
        cb = -al * rstd * c2;
        cc = al * (mean * rstd * c2 - c1);
    }
}

// sum over the 16 lanes of a DPP row (every lane gets the total): 4 VALU ops, no LDS traffic
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm 1,0,3,2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm 2,3,0,1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
    return v;
}

'''
s=s[:i0]+new+s[i1:]
save('common.cuh',s)

# ---------------- wgrad.hip: drop its local fdiv
s=load('wgrad.hip')
s=rep(s,'''
// floor(m / d) for 0 <= m < 2^21 with r = 1/d (d <= 2^10): exact, ~4 instructions instead of ~40
__device__ __forceinline__ int fdiv(int m, float r) { return (int)(((float)m + 0.5f) * r); }''','')
save('wgrad.hip',s)

# ---------------- conv_common.inc: DPP reduction
s=load('conv_common.inc')
s=rep(s,'''                float a = s0[ni][h][e], b = s1[ni][h][e];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) a += __shfl_xor(a, o), b += __shfl_xor(b, o);
                s0[ni][h][e] = a, s1[ni][h][e] = b;''','''                s0[ni][h][e] = row16_sum(s0[ni][h][e]), s1[ni][h][e] = row16_sum(s1[ni][h][e]);''')
save('conv_common.inc',s)

# ---------------- conv_core.hip
s=load('conv_core.hip')
s=rep(s,'''    int TH, TW;
    int dbg;
};''','''    int TH, TW;
    int dbg;
    float r_HC, r_TW, r_tc, r_PI, r_vp;  // reciprocals for fdiv
};''')
s=rep(s,'''    // ---- per-channel constants
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
''','''    // ---- per-channel constants, phase 1: only ISSUE the statistics loads of item `tid` (items:
    // cipad source channels, then the BCO channels of the ReLU-mask BatchNorm); the arithmetic runs
    // after the first tile's loads have been issued, so all of the prologue's loads fly together
    SrcRaw raw;
    const int nitem = k.cipad + (p.mask_y ? BCO : 0);
    const bool it_src = tid < k.cipad;
    const int it_c = it_src ? tid : n0 + (tid - k.cipad);
    const bool it_ok = it_src ? (tid < p.Ci) : (tid < nitem && it_c < p.Co);
    if (it_ok) {
        if (it_src) {
            src_raw_load(p.src, it_c, p.Ci, raw);
        } else {
            bn_raw_load(p.mask_bn, it_c, p.Co, raw);
            raw.b = p.mask_bn.beta[it_c];
        }
    }
    auto consts_finish = [&]() {
        if (tid < nitem) {
            if (it_src) {
                float a = 0.f, b = 0.f, cc = 0.f;
                if (it_ok) src_raw_finish(p.src, raw, a, b, cc);
                cs[tid] = a, cs[k.cipad + tid] = b, cs[2 * k.cipad + tid] = cc;
            } else {
                float a = 0.f, b = 0.f, mu = 0.f, rs = 0.f;
                if (it_ok) {
                    bn_raw_finish(p.mask_bn, raw, mu, rs);
                    a = raw.g * rs;
                    b = raw.b - mu * a;
                }
                const int c = tid - k.cipad;
                cm[c] = a, cm[BCO + c] = b, cm[2 * BCO + c] = mu, cm[3 * BCO + c] = rs;
            }
        }
        for (int item = tid + NTHR; item < nitem; item += NTHR) {  // more items than threads: rare
            if (item < k.cipad) {
                float a = 0.f, b = 0.f, cc = 0.f;
                if (item < p.Ci) src_consts(p.src, item, p.Ci, a, b, cc);
                cs[item] = a, cs[k.cipad + item] = b, cs[2 * k.cipad + item] = cc;
            } else {
                const int c = item - k.cipad;
                float a = 0.f, b = 0.f, mu = 0.f, rs = 0.f;
                if (n0 + c < p.Co) {
                    bn_mean_rstd(p.mask_bn, n0 + c, p.Co, mu, rs);
                    a = p.mask_bn.gamma[n0 + c] * rs;
                    b = p.mask_bn.beta[n0 + c] - mu * a;
                }
                cm[c] = a, cm[BCO + c] = b, cm[2 * BCO + c] = mu, cm[3 * BCO + c] = rs;
            }
        }
    };
''')
s=rep(s,'''            const int hp = v >> 2, hr = hp / k.HC;
            a_rc[i] = (hr << 16) | (hp - hr * k.HC);
        } else {''','''            const int hp = v >> 2, hr = fdiv(hp, k.r_HC);
            a_rc[i] = (hr << 16) | (hp - hr * k.HC);
        } else {''')
s=rep(s,'''            const int ty = m / k.TW;
            e_yx[mi] = (ty << 16) | (m - ty * k.TW);
        } else {
            m = 0;
        }
        const int ty = m / k.TW, tx = m - ty * k.TW;
        xoff[mi] = ((ty * k.seff) * k.HC + tx * k.seff) * PSA + g * 16;
    }
    const int woff = (wn * NTW * 16 + r16) * ROWB + g * 16;''','''            const int ty = fdiv(m, k.r_TW);
            e_yx[mi] = (ty << 16) | (m - ty * k.TW);
        } else {
            m = 0;
        }
        const int ty = fdiv(m, k.r_TW), tx = m - ty * k.TW;
        xoff[mi] = ((ty * k.seff) * k.HC + tx * k.seff) * PSA + g * 16;
    }
    const int woff = (wn * NTW * 16 + r16) * ROWB + g * 16;''')
s=rep(s,'''        const int tr = t / k.tiles_c, tc = t - tr * k.tiles_c;
        const int vrs = tr * k.TH * k.seff, cb = tc * k.TW * k.seff - k.pad;
        const int b0 = vrs / k.PI, y0 = vrs - b0 * k.PI - k.pad;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            go[i] = -1;''','''        const int tr = fdiv(t, k.r_tc), tc = t - tr * k.tiles_c;
        const int vrs = tr * k.TH * k.seff, cb = tc * k.TW * k.seff - k.pad;
        const int b0 = fdiv(vrs, k.r_PI), y0 = vrs - b0 * k.PI - k.pad;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            go[i] = -1;''')
# resident filters: loads early into rb[], LDS writes after the first tile's issue
s=rep(s,'''    if (k.wres) {  // whole K fits one chunk: filters stay resident in LDS for all tiles
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            if (tid + i * NTHR < BCO * TAPS * 4) {
                V16 val = zero16();
                if (b_g[i] >= 0 && (((tid + i * NTHR) & 3) * KV) < p.Ci) val = ldg16((const char*)p.w + (size_t)b_g[i] * sizeof(T));
                *reinterpret_cast<V16*>(sB + b_l[i]) = val;
            }
        }
    }
''','''    if (k.wres) {  // whole K fits one chunk: filters stay resident in LDS for all tiles (rb[] is free then)
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const bool ok = b_g[i] >= 0 && (((tid + i * NTHR) & 3) * KV) < p.Ci;
            rb[i] = ldg16((const char*)p.w + (size_t)(ok ? b_g[i] : 0) * sizeof(T));
        }
    }
''')
s=rep(s,'''    if (have) tile_setup(t, a_go);
    issue(a_go, 0, have);
    __syncthreads();  // constants + resident filters visible
    STAMP(4);''','''    if (have) tile_setup(t, a_go);
    issue(a_go, 0, have);
    consts_finish();
    if (k.wres) {
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const bool ok = b_g[i] >= 0 && (((tid + i * NTHR) & 3) * KV) < p.Ci;
            V16 val = rb[i];
            mask16(val, ok);
            if (tid + i * NTHR < BCO * TAPS * 4) *reinterpret_cast<V16*>(sB + b_l[i]) = val;
        }
    }
    __syncthreads();  // constants + resident filters visible
    STAMP(4);''')
s=rep(s,'''    while (have) {
        const int tr = t / k.tiles_c, tc = t - tr * k.tiles_c;
        const int vr0 = tr * k.TH, c0 = tc * k.TW;
        write_lds(a_go, ch0 * CK);''','''    while (have) {
        const int tr = fdiv(t, k.r_tc), tc = t - tr * k.tiles_c;
        const int vr0 = tr * k.TH, c0 = tc * k.TW;
        write_lds(a_go, ch0 * CK);''')
s=rep(s,'''            const int eb0 = vr0 / vpitch, ey0 = vr0 - eb0 * vpitch;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                bool pok = e_yx[mi] >= 0;
                int oy = ey0 + (e_yx[mi] >> 16), b = eb0;
                const int c = c0 + (e_yx[mi] & 0xffff);
                if (pok) {
                    while (oy >= vpitch) oy -= vpitch, ++b;
                    pok = (b < p.B) && (oy < p.Ho) && (c < p.Wo);
                }
                const size_t pix = pok ? (((size_t)b * p.Ho + oy) * p.Wo + c) * p.Co : 0;
                epilogue_tile<T, NTW, BCO>(p, acc[mi], cm, pok, pix, n0, wn * NTW * 16, g, s0, s1);''','''            const int eb0 = fdiv(vr0, k.r_vp), ey0 = vr0 - eb0 * vpitch;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                bool pok = e_yx[mi] >= 0;
                int oy = ey0 + (e_yx[mi] >> 16), b = eb0;
                const int c = c0 + (e_yx[mi] & 0xffff);
                if (pok) {
                    while (oy >= vpitch) oy -= vpitch, ++b;
                    pok = (b < p.B) && (oy < p.Ho) && (c < p.Wo);
                }
                const size_t pix = pok ? (((size_t)b * p.Ho + oy) * p.Wo + c) * p.Co : 0;
                epilogue_tile<T, NTW, BCO>(p, acc[mi], cm, pok, pix, n0, wn * NTW * 16, g, s0, s1);''')
s=rep(s,'''    k.npt = ceil_div(vrows, plan.TH) * k.tiles_c;
    const Shape sh = SHAPES[plan.shape];''','''    k.npt = ceil_div(vrows, plan.TH) * k.tiles_c;
    k.r_HC = 1.0f / k.HC, k.r_TW = 1.0f / k.TW, k.r_tc = 1.0f / k.tiles_c, k.r_PI = 1.0f / k.PI, k.r_vp = 1.0f / (p.Ho + 1);
    STL_CHECK(k.npt < (1 << 21) && (int64_t)vrows * 2 * k.PI < (1 << 21) && k.HP < 4096, "conv: index range exceeds the fast-division limits");
    const Shape sh = SHAPES[plan.shape];''')
save('conv_core.hip',s)

# ---------------- conv_ws.inc
s=load('conv_ws.inc')
s=rep(s,'''                const int hp = v >> 2, hr = hp / k.HC;''','''                const int hp = v >> 2, hr = fdiv(hp, k.r_HC);''')
s=rep(s,'''            const int tr = tt / k.tiles_c, tc = tt - tr * k.tiles_c;
            const int vrs = tr * k.TH * k.seff, cb = tc * k.TW * k.seff - k.pad;
            const int b0 = vrs / k.PI, y0 = vrs - b0 * k.PI - k.pad;''','''            const int tr = fdiv(tt, k.r_tc), tc = tt - tr * k.tiles_c;
            const int vrs = tr * k.TH * k.seff, cb = tc * k.TW * k.seff - k.pad;
            const int b0 = fdiv(vrs, k.r_PI), y0 = vrs - b0 * k.PI - k.pad;''')
s=rep(s,'''                const int ty = m / k.TW;
                e_yx[mi] = (ty << 16) | (m - ty * k.TW);''','''                const int ty = fdiv(m, k.r_TW);
                e_yx[mi] = (ty << 16) | (m - ty * k.TW);''')
s=rep(s,'''            const int ty = m / k.TW, tx = m - ty * k.TW;''','''            const int ty = fdiv(m, k.r_TW), tx = m - ty * k.TW;''')
s=rep(s,'''                const int tr = t / k.tiles_c, tc = t - tr * k.tiles_c;
                const int vr0 = tr * k.TH, c0 = tc * k.TW;
                const int eb0 = vr0 / vpitch, ey0 = vr0 - eb0 * vpitch;''','''                const int tr = fdiv(t, k.r_tc), tc = t - tr * k.tiles_c;
                const int vr0 = tr * k.TH, c0 = tc * k.TW;
                const int eb0 = fdiv(vr0, k.r_vp), ey0 = vr0 - eb0 * vpitch;''')
save('conv_ws.inc',s)
print("ok")
