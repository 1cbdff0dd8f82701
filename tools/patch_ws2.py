p='/root/repo/stlpose_amd/csrc/conv_ws.inc'
s=open(p).read()
# (1) resident multi-chunk weights
s=s.replace('''    if (k.wres) {  // whole K in one chunk: filters resident for all tiles (staged by all 512 threads)
        for (int v = tid; v < BCO * TAPS * 4; v += 512) {
            const int n = v / (TAPS * 4), r = v - n * (TAPS * 4), tap = r >> 2, part = r & 3;
            V16 val = zero16();
            if (n0 + n < p.Co && part * KV < p.Ci)
                val = ldg16((const char*)p.w + ((size_t)((n0 + n) * TAPS + tap) * p.Ci + part * KV) * sizeof(T));
            *reinterpret_cast<V16*>(sB + n * ROWB + tap * 64 + part * 16) = val;
        }
    }''','''    if (k.wres) {  // the block's whole filter slab (all K chunks) stays resident in LDS for all its tiles
        for (int v = tid; v < k.nchunks * BCO * TAPS * 4; v += 512) {
            const int c = v / (BCO * TAPS * 4), vv = v - c * (BCO * TAPS * 4);
            const int n = vv / (TAPS * 4), r = vv - n * (TAPS * 4), tap = r >> 2, part = r & 3;
            const int ch = c * CK + part * KV;
            V16 val = zero16();
            if (n0 + n < p.Co && ch < p.Ci)
                val = ldg16((const char*)p.w + ((size_t)((n0 + n) * TAPS + tap) * p.Ci + ch) * sizeof(T));
            *reinterpret_cast<V16*>(sB + c * (BCO * ROWB) + n * ROWB + tap * 64 + part * 16) = val;
        }
    }''')
s=s.replace('''            const char* cA = sA + buf * k.sz_a;
            const char* cB = sB + buf * k.sz_b;''','''            const char* cA = sA + buf * k.sz_a;
            const char* cB = k.wres ? sB + ch0 * (BCO * ROWB) : sB + buf * k.sz_b;''')
# (2)+(3) epilogue: widened stores for bf16
a=s.index("                    if (p.addend) {\n#pragma unroll\n                        for (int ni = 0; ni < NTW; ++ni) load4<T>(p.addend, eov[ni], ad[ni]);")
b=s.index("            WSTAMP(32 + dbi * 4 + 2);")
new='''                    if (p.addend) {
#pragma unroll
                        for (int ni = 0; ni < NTW; ++ni) load4<T>(p.addend, eov[ni], ad[ni]);
                    }
                    if (p.mask_y) {
#pragma unroll
                        for (int ni = 0; ni < NTW; ++ni) load4<T>(p.mask_y, eov[ni], my[ni]);
                    }
                    float fo[NTW][4];
#pragma unroll
                    for (int ni = 0; ni < NTW; ++ni) {
                        const int cl = ni * 16 + 4 * g;
                        const int co = n0 + cl;
                        const bool ok = okv[ni];
                        float f[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] = acc[mi][ni][r];
                        if (p.bias) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) f[r] += p.bias[ok ? co + r : 0];
                        }
                        if (p.addend) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) f[r] += ad[ni][r];
                        }
                        float yh[4] = {0.f, 0.f, 0.f, 0.f};
                        if (p.mask_y) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (p.mask_bn.relu && !(cm[cl + r] * my[ni][r] + cm[BCO + cl + r] > 0.f)) f[r] = 0.f;
                                yh[r] = (my[ni][r] - cm[2 * BCO + cl + r]) * cm[3 * BCO + cl + r];
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
#pragma unroll
                        for (int r = 0; r < 4; ++r) fo[ni][r] = f[r];
                        acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    if constexpr (sizeof(T) == 2) {
                        // widen the stores: lanes g and g^1 (16 lanes apart) exchange halves of two
                        // neighbouring channel tiles with v_permlane16_swap, so every lane stores 16
                        // contiguous bytes and a pixel gets 64 contiguous bytes per instruction
#pragma unroll
                        for (int ni = 0; ni < NTW; ni += 2) {
                            uint32_t ax = f32_to_bf16(fo[ni][0]) | (f32_to_bf16(fo[ni][1]) << 16);
                            uint32_t ay = f32_to_bf16(fo[ni][2]) | (f32_to_bf16(fo[ni][3]) << 16);
                            uint32_t bx = f32_to_bf16(fo[ni + 1][0]) | (f32_to_bf16(fo[ni + 1][1]) << 16);
                            uint32_t by = f32_to_bf16(fo[ni + 1][2]) | (f32_to_bf16(fo[ni + 1][3]) << 16);
                            const auto sx = __builtin_amdgcn_permlane16_swap(ax, bx, false, false);
                            const auto sy = __builtin_amdgcn_permlane16_swap(ay, by, false, false);
                            V16 v;
                            v.w[0] = sx[0], v.w[1] = sy[0], v.w[2] = sx[1], v.w[3] = sy[1];
                            const int nt = ni + (g & 1);                 // channel tile this lane now stores
                            const int co = n0 + nt * 16 + 8 * (g >> 1);  // its 8 consecutive channels
                            if (pok && co < p.Co) stg16((char*)p.out + (pix + co) * 2, v);
                        }
                    } else {
#pragma unroll
                        for (int ni = 0; ni < NTW; ++ni)
                            if (okv[ni]) store4<T>(p.out, eov[ni], fo[ni]);
                    }
                }
            }
'''
s=s[:a]+new+s[b:]
open(p,'w').write(s)

p='/root/repo/stlpose_amd/csrc/conv_core.hip'
s=open(p).read()
# lds_bytes: resident weights for ws when they fit
s=s.replace('''    const int off_b = off;
    const int sz_b = bco * (taps * 64 + 32);
    off += sz_b * ((ws && nchunks > 1) ? 2 : 1);''','''    const int off_b = off;
    const int sz_b = bco * (taps * 64 + 32);
    // wave-specialised kernel: keep the whole filter slab (all chunks) resident when it fits
    const bool resident = nchunks == 1 || (ws && (size_t)off + (size_t)sz_b * nchunks <= 150 * 1024);
    off += resident ? sz_b * nchunks : sz_b * (ws ? 2 : 1);''')
s=s.replace('''        out->sz_a = ws ? sz_a : 0, out->sz_b = (ws && nchunks > 1) ? sz_b : 0;''','''        out->sz_a = ws ? sz_a : 0, out->sz_b = (ws && !resident) ? sz_b : 0;
        out->wres = resident ? 1 : 0;''')
s=s.replace("    k.wres = k.nchunks == 1;\n","")
# tile search: also shorter tiles
s=s.replace('''        for (int tw = (p.Wo < 4 ? p.Wo : 4); tw <= p.Wo && tw <= sh.px; ++tw) {
            int th = sh.px / tw;
            if (th > vrows) th = vrows;
            if (th < 1) continue;
            const int hr = (th - 1) * p.stride + p.ks, hc = (tw - 1) * p.stride + p.ks;
            const int nva = ceil_div(hr * hc * 4, sh.lthr);
            if (nva > sh.nva_max) continue;''','''        for (int tw = (p.Wo < 4 ? p.Wo : 4); tw <= p.Wo && tw <= sh.px; ++tw)
          for (int frac = 4; frac >= 1; --frac) {
            int th = (sh.px / tw) * frac / 4;
            if (th > vrows) th = vrows;
            if (th < 1) continue;
            const int hr = (th - 1) * p.stride + p.ks, hc = (tw - 1) * p.stride + p.ks;
            const int nva = ceil_div(hr * hc * 4, sh.lthr);
            if (nva > sh.nva_max) continue;''')
open(p,'w').write(s)
print("ok")
