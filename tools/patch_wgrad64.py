p='/root/repo/stlpose_amd/csrc/wgrad.hip'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, (s.count(a), a)
    s=s.replace(a,b)

kernel = r'''
// ------------------------------------------------------------------------------------------------
// 64 x 64 variant (bf16, stride 1, Co >= 64 and Ci >= 64): the block owns 64 output x 64 input
// channels, each wave a 32 x 32 quadrant of them for ALL pixels of the tile (M/N split instead of
// the K split above).  Per staged byte it does twice the MFMA work of the 32 x 32 kernel (the
// activations are re-read Co/64 + Ci/64 times instead of Co/32 + Ci/32), a tile carries 4x the
// MFMAs per barrier pair, and there is no cross-wave reduction: every wave writes its own quadrant
// of the slab straight from the accumulators.  LDS pixel stride 160 B (128 B data + 32 B pad: the
// four rows of a transposed 16-lane read fall into disjoint bank groups).
template <typename T, int KS, int NVH, bool GQ, int TPX>
__global__ __launch_bounds__(256) void wgrad64_kernel(const WgK k) {
    static_assert(sizeof(T) == 2, "bf16 only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KV = 8, TAPS = KS * KS, KSTEP = 32, NR = 2;
    constexpr int VPX = 8;                  // 16-byte vectors per pixel (64 channels)
    constexpr int NVG = TPX * VPX / 256;    // g staging vectors per thread
    constexpr int NKS = TPX / KSTEP;        // K steps per tile, all done by every wave
    const stl_wgrad& p = k.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
    float* cgc = reinterpret_cast<float*>(smem + k.off_cg);  // [3][64]
    float* chc = reinterpret_cast<float*>(smem + k.off_ch);  // [2][64]
    char* sG = smem + k.off_g;
    char* sH = smem + k.off_h;
    WSTAMP(0);

    const int tilepx = p.TH * p.TW;
    const int vpitch = p.Ho + 1;
    const int nks = (tilepx + KSTEP - 1) / KSTEP;

    // ---- loop-invariant staging descriptors
    int g_yx[NVG];
    const int g_part = tid % VPX;
#pragma unroll
    for (int i = 0; i < NVG; ++i) {
        const int m = (tid + i * 256) / VPX;
        g_yx[i] = -1;
        if (m < tilepx) {
            const int ty = fdiv(m, k.r_TW);
            g_yx[i] = (ty << 16) | (m - ty * p.TW);
        }
    }
    int h_rc[NVH];
#pragma unroll
    for (int i = 0; i < NVH; ++i) {
        const int v = tid + i * 256;
        h_rc[i] = -1;
        if (v < k.HP * VPX) {
            const int hp = v / VPX, hr = fdiv(hp, k.r_HC);
            h_rc[i] = (hr << 16) | (hp - hr * k.HC);
        }
    }
    const bool g_chok = (co0 + g_part * KV) < p.Co, h_chok = (ci0 + g_part * KV) < p.Ci;

    // ---- MFMA-side offsets: pixel m = 32 s + 8 g + 4 i + ((lane & 15) >> 2) of K step s
    int rg0[NR], rh[NKS][NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) rg0[i] = (8 * g + 4 * i + ((lane & 15) >> 2)) * k.psg;
#pragma unroll
    for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            int m = s * KSTEP + 8 * g + 4 * i + ((lane & 15) >> 2);
            if (m >= tilepx) m = 0;  // the G rows beyond the tile are zero
            const int ty = fdiv(m, k.r_TW), tx = m - ty * p.TW;
            rh[s][i] = ((ty * p.stride) * k.HC + tx * p.stride) * k.psh;
        }

    V16 rgv[NVG], rgq[GQ ? NVG : 1], rhv[NVH];
    int g_go[NVG], h_go[NVH];

    auto setup = [&](int t) {
        const int tr = fdiv(t, k.r_tc), tc = t - tr * k.tiles_c;
        const int vr0 = tr * p.TH, c0 = tc * p.TW;
        const int gb0 = fdiv(vr0, k.r_vp), gy0 = vr0 - gb0 * vpitch;
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            g_go[i] = -1;
            if (g_yx[i] >= 0 && g_chok) {
                int oy = gy0 + (g_yx[i] >> 16), b = gb0;
                const int c = c0 + (g_yx[i] & 0xffff);
                while (oy >= vpitch) oy -= vpitch, ++b;
                if (b < p.B && oy < p.Ho && c < p.Wo) g_go[i] = ((b * p.Ho + oy) * p.Wo + c) * p.Co + co0 + g_part * KV;
            }
        }
        const int vrs = vr0 * p.stride, cb = c0 * p.stride - k.pad;
        const int hb0 = fdiv(vrs, k.r_PI), hy0 = vrs - hb0 * k.PI - k.pad;
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            h_go[i] = -1;
            if (h_rc[i] >= 0 && h_chok) {
                int iy = hy0 + (h_rc[i] >> 16), b = hb0;
                const int ix = cb + (h_rc[i] & 0xffff);
                if (iy >= 0 && ix >= 0 && ix < p.Wi) {
                    while (iy >= k.PI) iy -= k.PI, ++b;
                    if (b < p.B && iy < p.Hi) h_go[i] = ((b * p.Hi + iy) * p.Wi + ix) * p.Ci + ci0 + g_part * KV;
                }
            }
        }
    };
    auto issue = [&](bool en) {  // unconditional loads, clamped addresses
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            const size_t off = (en && g_go[i] >= 0) ? (size_t)g_go[i] : 0;
            rgv[i] = ldg16((const char*)p.g.x + off * sizeof(T));
            if (GQ) rgq[i] = ldg16((const char*)p.g.y + off * sizeof(T));
        }
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            const size_t off = (en && h_go[i] >= 0) ? (size_t)h_go[i] : 0;
            rhv[i] = ldg16((const char*)p.h.x + off * sizeof(T));
        }
    };
    const float relu_lo = p.h.relu ? 0.f : -INFINITY;
    auto write_lds = [&]() {
        const int cl = g_part * KV;
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            V16 val = rgv[i];
            if (GQ) val = xform_bnbwd<T>(val, rgq[i], cgc + cl, cgc + 64 + cl, cgc + 128 + cl);
            mask16(val, g_go[i] >= 0);
            const int v = tid + i * 256;
            *reinterpret_cast<V16*>(sG + (v / VPX) * k.psg + g_part * 16) = val;
        }
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            if (h_rc[i] < 0) continue;
            V16 val = rhv[i];
            if (p.h.mode == STL_SRC_BN) val = xform_bn<T>(val, chc + cl, chc + 64 + cl, relu_lo);
            mask16(val, h_go[i] >= 0);
            const int v = tid + i * 256;
            *reinterpret_cast<V16*>(sH + (v / VPX) * k.psh + g_part * 16) = val;
        }
    };

    f32x4 acc[2][2][TAPS];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[a][b][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int t = blockIdx.x;
    bool have = t < k.npt;
    WSTAMP(1);
    // BatchNorm constants (wave 3: the 64 channels of g, wave 2: those of h): statistics loads go out
    // ahead of the first tile's loads, the arithmetic runs while those are in flight
    SrcRaw raw;
    const bool cw = wave >= 2, cg = wave == 3;
    const bool cok = cw && (cg ? co0 + lane < p.Co : ci0 + lane < p.Ci);
    if (cok) {
        if (cg) src_raw_load(p.g, co0 + lane, p.Co, raw);
        else src_raw_load(p.h, ci0 + lane, p.Ci, raw);
    }
    if (have) setup(t);
    issue(have);
    WSTAMP(2);
    if (cw) {
        float a = 0.f, b = 0.f, cc = 0.f;
        if (cok) {
            if (cg) src_raw_finish(p.g, raw, a, b, cc);
            else src_raw_finish(p.h, raw, a, b, cc);
        }
        if (cg) cgc[lane] = a, cgc[64 + lane] = b, cgc[128 + lane] = cc;
        else chc[lane] = a, chc[64 + lane] = b;
    }
    __syncthreads();  // constants visible
    WSTAMP(3);
    bool first = true;

    const int acol = (wm * 32) * (int)sizeof(T), bcol = (wn * 32) * (int)sizeof(T);
    while (have) {
        write_lds();
        __syncthreads();
        if (first) WSTAMP(4);
        const int tn = t + gridDim.x;
        const bool have_n = tn < k.npt;
        if (have_n) setup(tn);
        issue(have_n);  // next tile's loads fly during the MFMAs
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            if (s < nks) {
                int rg[NR];
#pragma unroll
                for (int i = 0; i < NR; ++i) rg[i] = rg0[i] + s * KSTEP * k.psg;
                V16 a[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) a[mt] = frag_tr<T>(sG, rg, acol + mt * 16 * (int)sizeof(T), lane);
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    const int toff = ((tap / KS) * k.HC + (tap % KS)) * k.psh;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const V16 b = frag_tr<T>(sH + toff, rh[s], bcol + nt * 16 * (int)sizeof(T), lane);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) mma16<T>(acc[mt][nt][tap], a[mt], b);
                    }
                }
            }
        }
        __syncthreads();
        if (first) WSTAMP(5);
        first = false;
        t = tn, have = have_n;
    }
    WSTAMP(6);
    // ---- every wave writes its quadrant: acc[mt][nt][tap][r] = dw[co = 4g + r][ci = lane & 15]
    {
        float* slab = p.partial + (size_t)blockIdx.x * p.Co * TAPS * p.Ci;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int co = co0 + wm * 32 + mt * 16 + 4 * g, ci = ci0 + wn * 32 + nt * 16 + (lane & 15);
                float* dst = slab + (size_t)co * TAPS * p.Ci + ci;
                const bool ciok = ci < p.Ci;
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ciok && co + r < p.Co) dst[((size_t)r * TAPS + tap) * p.Ci] = acc[mt][nt][tap][r];
            }
    }
    WSTAMP(8);
}

template <typename T, int KS, int NVH, bool GQ, int TPX>
int launch64(const WgK& k, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad64_kernel<T, KS, NVH, GQ, TPX>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((wgrad64_kernel<T, KS, NVH, GQ, TPX>), grid, dim3(256), lds, st, k);
    STL_LAUNCH_CHECK("conv_wgrad64");
    return 0;
}

template <int KS>
int dispatch64(const WgK& k, dim3 grid, size_t lds, hipStream_t st) {
    const int nvh = ceil_div(k.HP * 8, 256);
    const bool gq = k.p.g.mode == STL_SRC_BNBWD;
    if (k.p.TH * k.p.TW > 128) {
        if (nvh <= 11) return gq ? launch64<__bf16, KS, 11, true, 256>(k, grid, lds, st) : launch64<__bf16, KS, 11, false, 256>(k, grid, lds, st);
    } else {
        if (nvh <= 6) return gq ? launch64<__bf16, KS, 6, true, 128>(k, grid, lds, st) : launch64<__bf16, KS, 6, false, 128>(k, grid, lds, st);
    }
    return stl_set_error("wgrad64: halo of %d pixels is too large for a %d-pixel tile", k.HP, k.p.TH * k.p.TW);
}

// channel tile (64 or 32) of the kernel variant stl_conv_wgrad picks for this problem
int wgrad_chunk(const stl_wgrad& p) {
    if (getenv("STL_WGRAD_NO64")) return 32;
    return (p.dtype == STL_BF16 && p.stride == 1 && p.Co >= 64 && p.Ci >= 64) ? 64 : 32;
}
'''
rep('''template <typename T, int KS>
int dispatch(const WgK& k, dim3 grid, size_t lds, hipStream_t st) {''', kernel + '''
template <typename T, int KS>
int dispatch(const WgK& k, dim3 grid, size_t lds, hipStream_t st) {''')

rep('''extern "C" int stl_conv_wgrad(const stl_wgrad* pp, void* stream) {''','''extern "C" int stl_wgrad_chunk(const stl_wgrad* pp) { return wgrad_chunk(*pp); }

extern "C" int stl_conv_wgrad(const stl_wgrad* pp, void* stream) {''')
rep('''    const int esz = p.dtype == STL_BF16 ? 2 : 4;
    k.psg = k.psh = 32 * esz + 16;''','''    hipStream_t st = (hipStream_t)stream;
    if (wgrad_chunk(p) == 64) {
        k.psg = k.psh = 160;
        k.off_cg = 0, k.off_ch = 3 * 64 * 4, k.off_g = 2048;  // consts: g [3][64] at 0, h [2][64] at 768
        k.off_h = k.off_g + (p.TH * p.TW > 128 ? 256 : 128) * k.psg;
        const size_t lds64 = (size_t)k.off_h + (size_t)k.HP * k.psh;
        STL_CHECK(lds64 <= 160 * 1024, "wgrad64: tile needs %zu B of LDS (>160 KiB)", lds64);
        STL_CHECK(p.nsplit <= k.npt || p.nsplit == 1, "wgrad: nsplit %d > tiles %d", p.nsplit, k.npt);
        dim3 grid64(p.nsplit, ceil_div(p.Co, 64), ceil_div(p.Ci, 64));
        return p.ks == 3 ? dispatch64<3>(k, grid64, lds64, st) : dispatch64<1>(k, grid64, lds64, st);
    }
    const int esz = p.dtype == STL_BF16 ? 2 : 4;
    k.psg = k.psh = 32 * esz + 16;''')
rep('''    dim3 grid(p.nsplit, ceil_div(p.Co, 32), ceil_div(p.Ci, 32));
    hipStream_t st = (hipStream_t)stream;''','''    dim3 grid(p.nsplit, ceil_div(p.Co, 32), ceil_div(p.Ci, 32));''')
open(p,'w').write(s)

p='/root/repo/include/stlpose_hip.h'
s=open(p).read()
rep('''int stl_conv_wgrad(const stl_wgrad* p, void* stream);''','''int stl_conv_wgrad(const stl_wgrad* p, void* stream);
/* Channel tile (32 or 64) of the kernel variant stl_conv_wgrad uses for this problem: the grid is
 * nsplit x ceil(Co/tile) x ceil(Ci/tile) blocks, which is what a caller sizes nsplit against. */
int stl_wgrad_chunk(const stl_wgrad* p);''')
open(p,'w').write(s)

p='/root/repo/stlpose_amd/capi.py'
s=open(p).read()
import re
m=re.search(r'    "stl_conv_wgrad": \[(.*?)\],\n', s)
assert m, "sig"
s=s.replace(m.group(0), m.group(0)+'    "stl_wgrad_chunk": [%s],\n' % m.group(1).split(',')[0].strip())
open(p,'w').write(s)
print(m.group(0))
