p = '/root/repo/stlpose_amd/csrc/conv_core.hip'
s = open(p).read()
s = s.replace('''    auto issue = [&](const int* go, int k0) {
        const bool chok = (k0 + a_part * KV) < p.Ci;''', '''    auto issue = [&](const int* go, int k0, bool en) {
        const bool chok = en && (k0 + a_part * KV) < p.Ci;''')
s = s.replace('''                const bool ok = b_g[i] >= 0 && (k0 + ((tid + i * NTHR) & 3) * KV) < p.Ci;
                rb[i] = ldg16((const char*)p.w + (size_t)(ok ? b_g[i] + k0 : 0) * sizeof(T));''', '''                const bool ok = en && b_g[i] >= 0 && (k0 + ((tid + i * NTHR) & 3) * KV) < p.Ci;
                rb[i] = ldg16((const char*)p.w + (size_t)(ok ? b_g[i] + k0 : 0) * sizeof(T));''')
a = s.index("    int it = lx;\n    int t = xcd * T8 + it;")
b = s.index("    // ---- flush statistics: lanes of one 16-lane group")
new = '''    int it = lx;
    int t = xcd * T8 + it;
    int ch0 = 0;
    bool have = (it < T8) && (t < k.npt);
    if (have) tile_setup(t, a_go);
    issue(a_go, 0, have);
    __syncthreads();  // constants + resident filters visible

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // flat loop over stages (tile, chunk); exactly ONE issue() site inside the loop so that the
    // staging registers need no PHI copies (which would force a vmcnt(0) before the MFMAs)
    while (have) {
        const int tr = t / k.tiles_c, tc = t - tr * k.tiles_c;
        const int vr0 = tr * k.TH, c0 = tc * k.TW;
        write_lds(a_go, ch0 * CK);
        __syncthreads();
        const bool last_chunk = (ch0 + 1 == k.nchunks);
        int itn = it, tn = t, chn = ch0 + 1;
        bool have_n = true;
        if (last_chunk) {
            itn = it + nx, tn = xcd * T8 + itn, chn = 0;
            have_n = (itn < T8) && (tn < k.npt);
            if (have_n) tile_setup(tn, a_go);
        }
        issue(a_go, chn * CK, have_n);  // next stage's loads land during the MFMAs below
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int toff = ((tap / KS) * k.HC + (tap % KS)) * PSA;
            V16 wf[NTW], xf[MT];
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni) wf[ni] = *reinterpret_cast<const V16*>(sB + woff + ni * 16 * ROWB + tap * 64);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) xf[mi] = *reinterpret_cast<const V16*>(sA + xoff[mi] + toff);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NTW; ++ni) mma16<T>(acc[mi][ni], wf[ni], xf[mi]);
        }
        __syncthreads();  // everyone is done with sA/sB of this stage
        if (last_chunk) {
            // ---- epilogue straight from the accumulators: lane = pixel r16, 4 channels per tile
            const int eb0 = vr0 / vpitch, ey0 = vr0 - eb0 * vpitch;
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
#pragma unroll
                for (int ni = 0; ni < NTW; ++ni) {
                    const int cl = (wn * NTW + ni) * 16 + 4 * g;  // channel within the block's BCO
                    const int co = n0 + cl;
                    const bool ok = pok && co < p.Co;
                    const size_t eo = ok ? pix + co : 0;  // invalid lanes read element 0, store nothing
                    float f[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) f[r] = acc[mi][ni][r];
                    if (p.bias && ok) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] += p.bias[co + r];
                    }
                    if (p.addend) {
                        float ad[4];
                        load4<T>(p.addend, eo, ad);
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] += ad[r];
                    }
                    float yh[4] = {0.f, 0.f, 0.f, 0.f};
                    if (p.mask_y) {
                        float my[4];
                        load4<T>(p.mask_y, eo, my);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (p.mask_bn.relu && !(cm[cl + r] * my[r] + cm[BCO + cl + r] > 0.f)) f[r] = 0.f;
                            yh[r] = (my[r] - cm[2 * BCO + cl + r]) * cm[3 * BCO + cl + r];
                        }
                    }
                    if (p.out_relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] = fmaxf(f[r], 0.f);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) f[r] = ok ? round_to<T>(f[r]) : 0.f;
                    if (p.out_stats) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) s0[ni][r] += f[r], s1[ni][r] += f[r] * f[r];
                    } else if (p.red) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) s0[ni][r] += f[r], s1[ni][r] += f[r] * yh[r];
                    }
                    if (ok) store4<T>(p.out, eo, f);
                    acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        it = itn, t = tn, ch0 = chn, have = have_n;
    }
'''
s = s[:a] + new + s[b:]
open(p, 'w').write(s)
print("patched")
