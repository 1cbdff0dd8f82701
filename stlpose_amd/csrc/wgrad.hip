// Weight gradient of the implicit-GEMM convolution on gfx950.
//   dw[co][tap][ci] = sum over output pixels of  g[pixel][co] * h[pixel*stride + tap - pad][ci]
// GEMM view: M = 32 output channels, N = 32 input channels per tap, K = output pixels.
// Both operands are K-major in memory (NHWC), so fragments are read TRANSPOSED from LDS:
// ds_read_b64_tr_b16 for bf16, plain 4-byte reads for fp32 (one element per lane per MFMA).
// Each of the 4 waves takes a quarter of the tile's pixels (its own K range) and keeps all
// 2x2xTAPS 16x16 accumulators in registers across the block's tiles; the block reduces its
// waves through LDS once and writes one fp32 slab [Co][taps][Ci] (deterministic split-K;
// slabs are summed by stl_reduce_slabs).
// Pipeline: the global loads of tile t+1 are issued (unconditionally, clamped addresses) before
// the MFMAs of tile t and written to LDS after them; index arithmetic is hoisted out of the loop.
#include "common.cuh"

namespace {

#include "conv_common.inc"

struct WgK {
    stl_wgrad p;
    int tiles_c, npt, HR, HC, HP, PI, pad, taps;
    int psg, psh;  // LDS bytes per pixel (32 channels + 16 B pad)
    int off_cg, off_ch, off_g, off_h;
};

template <typename T>
__device__ __forceinline__ V16 frag_tr(const char* base, const int* rowoff, int colbyte, int lane);
// bf16: rows rowoff[0..1] are this lane's two 4-pixel groups (already including q); the read
// returns, for column (lane&15), the 4 pixels of the group.
template <>
__device__ __forceinline__ V16 frag_tr<__bf16>(const char* base, const int* rowoff, int colbyte, int lane) {
    V16 v;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const char* ptr = base + rowoff[h] + colbyte + (lane & 3) * 8;
        s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(uintptr_t)(uint32_t)(uintptr_t)ptr);
        const uint64_t bits = __builtin_bit_cast(uint64_t, r);
        v.w[2 * h] = (uint32_t)bits;
        v.w[2 * h + 1] = (uint32_t)(bits >> 32);
    }
    return v;
}
template <>
__device__ __forceinline__ V16 frag_tr<float>(const char* base, const int* rowoff, int colbyte, int lane) {
    V16 v;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        v.w[s] = *reinterpret_cast<const uint32_t*>(base + rowoff[s] + colbyte + (lane & 15) * 4);
    return v;
}

// NVH: h (input halo) staging vectors per thread; GQ: g is BNBWD (second tensor on load)
template <typename T, int KS, int NVH, bool GQ>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgK k) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KV = ET<T>::KV, TAPS = KS * KS;
    constexpr int KSTEP = 4 * KV;               // pixels per MFMA K step (32 bf16 / 16 f32)
    constexpr int NR = sizeof(T) == 2 ? 2 : 4;  // row offsets a lane needs per fragment
    constexpr int VPX = 32 / KV;                // 16-byte vectors per pixel (32 channels)
    constexpr int NVG = 128 * VPX / 256;        // g staging vectors per thread
    const stl_wgrad& p = k.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.z * 32;
    float* cgc = reinterpret_cast<float*>(smem + k.off_cg);  // [3][32]
    float* chc = reinterpret_cast<float*>(smem + k.off_ch);  // [2][32]
    char* sG = smem + k.off_g;
    char* sH = smem + k.off_h;

    if (tid < 32) {
        float a = 0.f, b = 0.f, c = 0.f;
        if (co0 + tid < p.Co) src_consts(p.g, co0 + tid, p.Co, a, b, c);
        cgc[tid] = a, cgc[32 + tid] = b, cgc[64 + tid] = c;
    } else if (tid < 64) {
        const int t = tid - 32;
        float a = 0.f, b = 0.f, c = 0.f;
        if (ci0 + t < p.Ci) src_consts(p.h, ci0 + t, p.Ci, a, b, c);
        chc[t] = a, chc[32 + t] = b;
    }

    const int tilepx = p.TH * p.TW;
    const int vpitch = p.Ho + 1;
    const int nks = (tilepx + KSTEP - 1) / KSTEP;

    // ---- loop-invariant staging descriptors
    int g_yx[NVG];  // (ty << 16) | tx of the tile pixel of slot i, -1 = beyond the tile (zero row)
    const int g_part = tid % VPX;
#pragma unroll
    for (int i = 0; i < NVG; ++i) {
        const int m = (tid + i * 256) / VPX;
        g_yx[i] = -1;
        if (m < tilepx) {
            const int ty = m / p.TW;
            g_yx[i] = (ty << 16) | (m - ty * p.TW);
        }
    }
    int h_rc[NVH];
#pragma unroll
    for (int i = 0; i < NVH; ++i) {
        const int v = tid + i * 256;
        h_rc[i] = -1;
        if (v < k.HP * VPX) {
            const int hp = v / VPX, hr = hp / k.HC;
            h_rc[i] = (hr << 16) | (hp - hr * k.HC);
        }
    }
    const bool g_chok = (co0 + g_part * KV) < p.Co, h_chok = (ci0 + g_part * KV) < p.Ci;

    // ---- MFMA-side offsets for this wave's K steps (tile geometry is the same for every tile)
    // at most 2 K steps per wave (128 px / KSTEP / 4 waves: 1 for bf16, 2 for fp32)
    constexpr int NKS = (128 / KSTEP + 3) / 4;
    int rg[NKS][NR], rh[NKS][NR];
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        const int kb = (wave + 4 * s) * KSTEP;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            int m;
            if constexpr (sizeof(T) == 2)
                m = kb + 8 * g + 4 * i + ((lane & 15) >> 2);
            else
                m = kb + 4 * g + i;
            if (m > 127) m = 127;
            rg[s][i] = m * k.psg;  // rows >= tilepx are zero-filled in sG
            if (m >= tilepx) m = 0;
            const int ty = m / p.TW, tx = m - ty * p.TW;
            rh[s][i] = ((ty * p.stride) * k.HC + tx * p.stride) * k.psh;
        }
    }

    V16 rgv[NVG], rgq[GQ ? NVG : 1], rhv[NVH];
    int g_go[NVG], h_go[NVH];

    auto setup = [&](int t) {
        const int tr = t / k.tiles_c, tc = t - tr * k.tiles_c;
        const int vr0 = tr * p.TH, c0 = tc * p.TW;
        const int gb0 = vr0 / vpitch, gy0 = vr0 - gb0 * vpitch;
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
        const int hb0 = vrs / k.PI, hy0 = vrs - hb0 * k.PI - k.pad;
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
    // unconditional loads (a guarded load would be serialised by the compiler); invalid slots read
    // element 0 and are zeroed when written to LDS
    auto issue = [&](bool en) {
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
            if (GQ) val = xform_bnbwd<T>(val, rgq[i], cgc + cl, cgc + 32 + cl, cgc + 64 + cl);
            mask16(val, g_go[i] >= 0);
            const int v = tid + i * 256;
            *reinterpret_cast<V16*>(sG + (v / VPX) * k.psg + g_part * 16) = val;
        }
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            if (h_rc[i] < 0) continue;
            V16 val = rhv[i];
            if (p.h.mode == STL_SRC_BN) val = xform_bn<T>(val, chc + cl, chc + 32 + cl, relu_lo);
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
    if (have) setup(t);
    issue(have);
    __syncthreads();  // constants visible

    while (have) {
        write_lds();
        __syncthreads();
        const int tn = t + gridDim.x;
        const bool have_n = tn < k.npt;
        if (have_n) setup(tn);
        issue(have_n);  // next tile's loads fly during the MFMAs
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            if (wave + 4 * s < nks) {
                V16 a[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) a[mt] = frag_tr<T>(sG, rg[s], mt * 16 * (int)sizeof(T), lane);
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    const int toff = ((tap / KS) * k.HC + (tap % KS)) * k.psh;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const V16 b = frag_tr<T>(sH + toff, rh[s], nt * 16 * (int)sizeof(T), lane);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) mma16<T>(acc[mt][nt][tap], a[mt], b);
                    }
                }
            }
        }
        __syncthreads();
        t = tn, have = have_n;
    }
    // ---- reduce the 4 waves through LDS, write this block's slab
    float* red = reinterpret_cast<float*>(smem + k.off_g);  // [4*TAPS][256]
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* d = red + (((mt * 2 + nt) * TAPS + tap) * 256 + lane * 4 + r);
                            *d = (w == 0 ? 0.f : *d) + acc[mt][nt][tap][r];
                        }
        }
        __syncthreads();
    }
    float* slab = p.partial + (size_t)blockIdx.x * p.Co * TAPS * p.Ci;
    for (int e = tid; e < 32 * TAPS * 32; e += 256) {
        const int col = e / (TAPS * 32), rem = e - col * (TAPS * 32), tap = rem >> 5, cil = rem & 31;
        if (co0 + col >= p.Co || ci0 + cil >= p.Ci) continue;
        const int mt = col >> 4, nt = cil >> 4;
        const int ln = (cil & 15) + 16 * ((col & 15) >> 2), r = col & 3;
        slab[((size_t)(co0 + col) * TAPS + tap) * p.Ci + ci0 + cil] = red[((mt * 2 + nt) * TAPS + tap) * 256 + ln * 4 + r];
    }
}

template <typename T, int KS, int NVH, bool GQ>
int launch(const WgK& k, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, KS, NVH, GQ>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((wgrad_kernel<T, KS, NVH, GQ>), grid, dim3(256), lds, st, k);
    STL_LAUNCH_CHECK("conv_wgrad");
    return 0;
}

template <typename T, int KS>
int dispatch(const WgK& k, dim3 grid, size_t lds, hipStream_t st) {
    const int vpx = 32 / ET<T>::KV;
    const int nvh = ceil_div(k.HP * vpx, 256);
    const bool gq = k.p.g.mode == STL_SRC_BNBWD;
    if (nvh <= 3) return gq ? launch<T, KS, 3, true>(k, grid, lds, st) : launch<T, KS, 3, false>(k, grid, lds, st);
    if (nvh <= 6) return gq ? launch<T, KS, 6, true>(k, grid, lds, st) : launch<T, KS, 6, false>(k, grid, lds, st);
    if (nvh <= 9) return gq ? launch<T, KS, 9, true>(k, grid, lds, st) : launch<T, KS, 9, false>(k, grid, lds, st);
    if (nvh <= 18) return gq ? launch<T, KS, 18, true>(k, grid, lds, st) : launch<T, KS, 18, false>(k, grid, lds, st);
    return stl_set_error("wgrad: halo of %d pixels needs %d staging vectors per thread (max 18); shrink the tile", k.HP, nvh);
}

}  // namespace

extern "C" int stl_conv_wgrad(const stl_wgrad* pp, void* stream) {
    const stl_wgrad& p = *pp;
    STL_CHECK(p.dtype == STL_F32 || p.dtype == STL_BF16, "wgrad: bad dtype");
    STL_CHECK(p.ks == 1 || p.ks == 3, "wgrad: ks must be 1 or 3");
    STL_CHECK(p.stride == 1 || p.stride == 2, "wgrad: stride must be 1 or 2");
    const int kv = p.dtype == STL_BF16 ? 8 : 4;
    STL_CHECK(p.Ci % kv == 0 && p.Co % kv == 0, "wgrad: Ci=%d / Co=%d must be multiples of %d", p.Ci, p.Co, kv);
    STL_CHECK(p.TH >= 1 && p.TW >= 1 && p.TH * p.TW <= 128, "wgrad: tile exceeds 128 pixels");
    STL_CHECK((int64_t)p.B * p.Hi * p.Wi * p.Ci < (1ll << 31) && (int64_t)p.B * p.Ho * p.Wo * p.Co < (1ll << 31),
              "wgrad: tensors of 2^31 or more elements are not supported");
    const int pad = p.ks == 3 ? 1 : 0;
    STL_CHECK((p.Hi + 2 * pad - p.ks) / p.stride + 1 == p.Ho && (p.Wi + 2 * pad - p.ks) / p.stride + 1 == p.Wo,
              "wgrad: output %dx%d inconsistent with input %dx%d", p.Ho, p.Wo, p.Hi, p.Wi);
    STL_CHECK(p.h.x && p.g.x && p.partial && p.nsplit >= 1, "wgrad: null pointer / nsplit");
    STL_CHECK(p.h.mode == STL_SRC_PLAIN || p.h.mode == STL_SRC_BN, "wgrad: h must be PLAIN or BN");
    STL_CHECK(p.g.mode == STL_SRC_PLAIN || p.g.mode == STL_SRC_BNBWD, "wgrad: g must be PLAIN or BNBWD");
    STL_CHECK(p.g.mode != STL_SRC_BNBWD || (p.g.y && p.g.stats && p.g.rstats && p.g.gamma), "wgrad: BNBWD source incomplete");
    STL_CHECK(p.h.mode != STL_SRC_BN || (p.h.gamma && p.h.beta && (p.h.stats || (p.h.rmean && p.h.rvar))), "wgrad: BN source incomplete");
    WgK k;
    k.p = p;
    k.taps = p.ks * p.ks;
    k.pad = pad;
    k.PI = p.stride * (p.Ho + 1);
    k.HR = (p.TH - 1) * p.stride + p.ks;
    k.HC = (p.TW - 1) * p.stride + p.ks;
    k.HP = k.HR * k.HC;
    k.tiles_c = ceil_div(p.Wo, p.TW);
    k.npt = ceil_div(p.B * (p.Ho + 1), p.TH) * k.tiles_c;
    const int esz = p.dtype == STL_BF16 ? 2 : 4;
    k.psg = k.psh = 32 * esz + 16;
    k.off_cg = 0;
    k.off_ch = 3 * 32 * 4;
    k.off_g = 1024;  // consts: g [3][32] floats at 0, h [2][32] floats at 384 -> 640 B used
    int szG = 128 * k.psg;
    const int szRed = 4 * k.taps * 256 * 4;
    int szH = k.HP * k.psh;
    k.off_h = k.off_g + szG;
    size_t lds = (size_t)k.off_h + szH;
    if ((size_t)k.off_g + szRed > lds) lds = (size_t)k.off_g + szRed;
    STL_CHECK(lds <= 160 * 1024, "wgrad: tile needs %zu B of LDS (>160 KiB)", lds);
    STL_CHECK(p.nsplit <= k.npt || p.nsplit == 1, "wgrad: nsplit %d > tiles %d", p.nsplit, k.npt);
    dim3 grid(p.nsplit, ceil_div(p.Co, 32), ceil_div(p.Ci, 32));
    hipStream_t st = (hipStream_t)stream;
    if (p.dtype == STL_BF16) return p.ks == 3 ? dispatch<__bf16, 3>(k, grid, lds, st) : dispatch<__bf16, 1>(k, grid, lds, st);
    return p.ks == 3 ? dispatch<float, 3>(k, grid, lds, st) : dispatch<float, 1>(k, grid, lds, st);
}
