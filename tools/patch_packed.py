import re
# ---------------- conv_ws.inc
p='/root/repo/stlpose_amd/csrc/conv_ws.inc'
s=open(p).read()
# loader transform
a=s.index("                const bool ok = a_go[i] >= 0 && ch < p.Ci;\n                V16 val = ra[i];")
b=s.index("                const int v = ltid + i * 256;\n                *reinterpret_cast<V16*>(dA + (v >> 2) * PSA + (v & 3) * 16) = val;")
s=s[:a]+'''                const bool ok = a_go[i] >= 0 && ch < p.Ci;
                const int chc = ok ? ch : 0;
                V16 val = ra[i];
                if (Q)
                    val = xform_bnbwd<T>(val, rq[i], cs + chc, cs + k.cipad + chc, cs + 2 * k.cipad + chc);
                else if (p.src.mode != STL_SRC_PLAIN)
                    val = xform_bn<T>(val, cs + chc, cs + k.cipad + chc, relu_lo);
                mask16(val, ok);  // zero padding applies AFTER the transform
'''+s[b:]
s=s.replace('''        auto write_lds = [&](int buf, int k0) {
            char* dA = sA + buf * k.sz_a;''','''        const float relu_lo = p.src.relu ? 0.f : -INFINITY;
        auto write_lds = [&](int buf, int k0) {
            char* dA = sA + buf * k.sz_a;''')
s=s.replace('''                    const bool ok = b_g[i] >= 0 && (k0 + ((ltid + i * 256) & 3) * KV) < p.Ci;
                    const uint32_t keep = ok ? 0xFFFFFFFFu : 0u;
                    V16 val = rb[i];
                    val.w[0] &= keep, val.w[1] &= keep, val.w[2] &= keep, val.w[3] &= keep;
                    if (ltid + i * 256 < BCO * TAPS * 4) *reinterpret_cast<V16*>(dB + b_l[i]) = val;''','''                    const bool ok = b_g[i] >= 0 && (k0 + ((ltid + i * 256) & 3) * KV) < p.Ci;
                    V16 val = rb[i];
                    mask16(val, ok);
                    if (ltid + i * 256 < BCO * TAPS * 4) *reinterpret_cast<V16*>(dB + b_l[i]) = val;''')
# compute: stats arrays + epilogue
s=s.replace('''        float s0[NTW][4], s1[NTW][4];
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) s0[ni][r] = s1[ni][r] = 0.f;''','''        f2v s0[NTW][2], s1[NTW][2];
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
            for (int h = 0; h < 2; ++h) s0[ni][h] = f2v{0.f, 0.f}, s1[ni][h] = f2v{0.f, 0.f};''')
a=s.index("                    const size_t pix = pok ? (((size_t)b * p.Ho + oy) * p.Wo + c) * p.Co : 0;\n                    float ad[NTW][4], my[NTW][4];")
b=s.index("            WSTAMP(32 + dbi * 4 + 2);")
s=s[:a]+'''                    const size_t pix = pok ? (((size_t)b * p.Ho + oy) * p.Wo + c) * p.Co : 0;
                    epilogue_tile<T, NTW, BCO>(p, acc[mi], cm, pok, pix, n0, 0, g, s0, s1);
                }
            }
'''+s[b:]
a=s.index("        if (dst) {\n#pragma unroll\n            for (int ni = 0; ni < NTW; ++ni)\n#pragma unroll\n                for (int r = 0; r < 4; ++r) {\n#pragma unroll\n                    for (int o = 1; o < 16; o <<= 1) {")
b=s.index("        __syncthreads();  // (D) all LDS image reads are over")
s=s[:a]+"        if (dst) xor_reduce_stats<NTW>(s0, s1);\n"+s[b:]
s=s.replace('''                for (int r = 0; r < 4; ++r) {
                    const int cl = ni * 16 + 4 * g + r;
                    red[(cw * 2 + 0) * BCO + cl] = s0[ni][r];
                    red[(cw * 2 + 1) * BCO + cl] = s1[ni][r];
                }''','''                for (int r = 0; r < 4; ++r) {
                    const int cl = ni * 16 + 4 * g + r;
                    red[(cw * 2 + 0) * BCO + cl] = s0[ni][r >> 1][r & 1];
                    red[(cw * 2 + 1) * BCO + cl] = s1[ni][r >> 1][r & 1];
                }''')
open(p,'w').write(s)

# ---------------- conv_core.hip (uniform kernel)
p='/root/repo/stlpose_amd/csrc/conv_core.hip'
s=open(p).read()
s=s.replace('#include "conv_ws.inc"','#include "conv_ws.inc"',1)
# include common helpers right after the load4/store4 helpers: put before the first kernel template
s=s.replace("// WM x WN waves, MT pixel tiles and NTW channel tiles per wave; NVA staging vectors per thread for","#include \"conv_common.inc\"\n\n// WM x WN waves, MT pixel tiles and NTW channel tiles per wave; NVA staging vectors per thread for",1)
a=s.index("            const bool ok = go[i] >= 0 && ch < p.Ci;\n            V16 val = ra[i];")
b=s.index("            const int v = tid + i * NTHR;\n            *reinterpret_cast<V16*>(sA + (v >> 2) * PSA + (v & 3) * 16) = val;")
s=s[:a]+'''            const bool ok = go[i] >= 0 && ch < p.Ci;
            const int chc = ok ? ch : 0;
            V16 val = ra[i];
            if (Q)
                val = xform_bnbwd<T>(val, rq[i], cs + chc, cs + k.cipad + chc, cs + 2 * k.cipad + chc);
            else if (p.src.mode != STL_SRC_PLAIN)
                val = xform_bn<T>(val, cs + chc, cs + k.cipad + chc, relu_lo);
            mask16(val, ok);  // zero padding applies AFTER the transform
'''+s[b:]
s=s.replace("    auto write_lds = [&](const int* go, int k0) {\n        const int ch = k0 + a_part * KV;","    const float relu_lo = p.src.relu ? 0.f : -INFINITY;\n    auto write_lds = [&](const int* go, int k0) {\n        const int ch = k0 + a_part * KV;",1)
s=s.replace('''                const bool ok = b_g[i] >= 0 && (k0 + ((tid + i * NTHR) & 3) * KV) < p.Ci;
                const uint32_t keep = ok ? 0xFFFFFFFFu : 0u;
                V16 val = rb[i];
                val.w[0] &= keep, val.w[1] &= keep, val.w[2] &= keep, val.w[3] &= keep;
                if (tid + i * NTHR < BCO * TAPS * 4) *reinterpret_cast<V16*>(sB + b_l[i]) = val;''','''                const bool ok = b_g[i] >= 0 && (k0 + ((tid + i * NTHR) & 3) * KV) < p.Ci;
                V16 val = rb[i];
                mask16(val, ok);
                if (tid + i * NTHR < BCO * TAPS * 4) *reinterpret_cast<V16*>(sB + b_l[i]) = val;''')
s=s.replace('''    float s0[NTW][4], s1[NTW][4];
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) s0[ni][r] = s1[ni][r] = 0.f;''','''    f2v s0[NTW][2], s1[NTW][2];
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
        for (int h = 0; h < 2; ++h) s0[ni][h] = f2v{0.f, 0.f}, s1[ni][h] = f2v{0.f, 0.f};''')
a=s.index("                const size_t pix = pok ? (((size_t)b * p.Ho + oy) * p.Wo + c) * p.Co : 0;\n                // issue every load of this pixel tile first")
b=s.index("        if (last_chunk) STAMP(9);")
s=s[:a]+'''                const size_t pix = pok ? (((size_t)b * p.Ho + oy) * p.Wo + c) * p.Co : 0;
                epilogue_tile<T, NTW, BCO>(p, acc[mi], cm, pok, pix, n0, wn * NTW * 16, g, s0, s1);
            }
        }
'''+s[b:]
a=s.index("    if (dst) {\n#pragma unroll\n        for (int ni = 0; ni < NTW; ++ni)\n#pragma unroll\n            for (int r = 0; r < 4; ++r) {\n#pragma unroll\n                for (int o = 1; o < 16; o <<= 1) {")
b=s.index("        float* red = reinterpret_cast<float*>(smem + k.off_red);  // [WM][2][BCO]")
s=s[:a]+"    if (dst) {\n        xor_reduce_stats<NTW>(s0, s1);\n"+s[b:]
s=s.replace('''                    red[(wm * 2 + 0) * BCO + cl] = s0[ni][r];
                    red[(wm * 2 + 1) * BCO + cl] = s1[ni][r];''','''                    red[(wm * 2 + 0) * BCO + cl] = s0[ni][r >> 1][r & 1];
                    red[(wm * 2 + 1) * BCO + cl] = s1[ni][r >> 1][r & 1];''')
open(p,'w').write(s)
print('ok')
