p='/root/repo/stlpose_amd/csrc/conv_ws.inc'
s=open(p).read()
a=s.index("#pragma unroll\n            for (int tap = 0; tap < TAPS; ++tap) {\n                const int toff = ((tap / KS) * k.HC + (tap % KS)) * PSA;\n                V16 wf[NTW];")
b=s.index("            WSTAMP(32 + dbi * 4 + 1);")
s=s[:a]+'''            // fragment reads of tap t+1 are issued before the MFMAs of tap t (static double buffer)
            V16 wf[2][NTW], xf[2][MT];
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni) wf[0][ni] = *reinterpret_cast<const V16*>(cB + woff + ni * 16 * ROWB);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) xf[0][mi] = *reinterpret_cast<const V16*>(cA + xoff[mi]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                if (tap + 1 < TAPS) {
                    const int toff = (((tap + 1) / KS) * k.HC + ((tap + 1) % KS)) * PSA;
#pragma unroll
                    for (int ni = 0; ni < NTW; ++ni)
                        wf[(tap + 1) & 1][ni] = *reinterpret_cast<const V16*>(cB + woff + ni * 16 * ROWB + (tap + 1) * 64);
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) xf[(tap + 1) & 1][mi] = *reinterpret_cast<const V16*>(cA + xoff[mi] + toff);
                }
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NTW; ++ni) mma16<T>(acc[mi][ni], wf[tap & 1][ni], xf[tap & 1][mi]);
            }
'''+s[b:]
open(p,'w').write(s)
p='/root/repo/stlpose_amd/csrc/conv_core.hip'
s=open(p).read()
a=s.index("#pragma unroll\n        for (int tap = 0; tap < TAPS; ++tap) {\n            const int toff = ((tap / KS) * k.HC + (tap % KS)) * PSA;\n            V16 wf[NTW], xf[MT];")
b=s.index("        __syncthreads();  // everyone is done with sA/sB of this stage")
s=s[:a]+'''        {  // fragment reads of tap t+1 are issued before the MFMAs of tap t (static double buffer)
            V16 wf[2][NTW], xf[2][MT];
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni) wf[0][ni] = *reinterpret_cast<const V16*>(sB + woff + ni * 16 * ROWB);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) xf[0][mi] = *reinterpret_cast<const V16*>(sA + xoff[mi]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                if (tap + 1 < TAPS) {
                    const int toff = (((tap + 1) / KS) * k.HC + ((tap + 1) % KS)) * PSA;
#pragma unroll
                    for (int ni = 0; ni < NTW; ++ni)
                        wf[(tap + 1) & 1][ni] = *reinterpret_cast<const V16*>(sB + woff + ni * 16 * ROWB + (tap + 1) * 64);
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) xf[(tap + 1) & 1][mi] = *reinterpret_cast<const V16*>(sA + xoff[mi] + toff);
                }
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NTW; ++ni) mma16<T>(acc[mi][ni], wf[tap & 1][ni], xf[tap & 1][mi]);
            }
        }
'''+s[b:]
open(p,'w').write(s)
print('ok')
