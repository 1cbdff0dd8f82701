p='/root/repo/stlpose_amd/csrc/wgrad.hip'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, (s.count(a), a)
    s=s.replace(a,b)
i0=s.index('    // ---- reduce the 4 waves through LDS, write this block\'s slab')
i1=s.index('    WSTAMP(8);\n}')
new='''    // ---- reduce the 4 waves through two LDS regions (fixed order (w0+w2)+(w1+w3): deterministic),
    // then every wave writes one (mt, nt) quadrant of the block's slab.
    // region layout: [tile = (mt*2+nt)*TAPS+tap][lane] f32x4 -> conflict-free 16-byte accesses
    f32x4* red = reinterpret_cast<f32x4*>(smem + k.off_g);
    constexpr int RT = 4 * TAPS * 64;  // f32x4 per region
    {
        f32x4* mine = red + (wave & 1) * RT + lane;
        if (wave >= 2) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int tap = 0; tap < TAPS; ++tap) mine[((mt * 2 + nt) * TAPS + tap) * 64] = acc[mt][nt][tap];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int tap = 0; tap < TAPS; ++tap) {
                        f32x4* d = mine + ((mt * 2 + nt) * TAPS + tap) * 64;
                        *d = acc[mt][nt][tap] + *d;
                    }
        }
        __syncthreads();
    }
    WSTAMP(7);
    {
        float* slab = p.partial + (size_t)blockIdx.x * p.Co * TAPS * p.Ci;
        const int mt = wave >> 1, nt = wave & 1;
        const int ci = ci0 + nt * 16 + (lane & 15), co = co0 + mt * 16 + 4 * g;
        const f32x4* src = red + (wave * TAPS) * 64 + lane;
        float* dst = slab + (size_t)co * TAPS * p.Ci + ci;
        const bool ciok = ci < p.Ci;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const f32x4 v = src[tap * 64] + src[RT + tap * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ciok && co + r < p.Co) dst[((size_t)r * TAPS + tap) * p.Ci] = v[r];
        }
    }
'''
s=s[:i0]+new+s[i1:]
# LDS size of the reduction: two regions
rep('const int szRed = 4 * k.taps * 256 * 4;','const int szRed = 2 * 4 * k.taps * 64 * 16;  // two regions of [4*taps][64] f32x4')
open(p,'w').write(s)
