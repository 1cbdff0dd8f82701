// Implicit-GEMM convolution for gfx950: NHWC, im2col-free, MFMA 16x16 tiles, LDS-staged input
// halo tile and filter slice, "normalise on load" prologue and a fused epilogue.
// One kernel serves the forward conv and (with transposed/flipped filters) the data gradient.
//
//   GEMM view:  M = output channels (MFMA rows), N = output pixels (MFMA columns),
//               K = taps x input channels, walked as 64-byte channel chunks (32 bf16 / 16 f32).
//   The operands are swapped on purpose (filters = A, pixels = B): every lane then owns 4
//   CONSECUTIVE output channels of one pixel, so the epilogue stores 8/16-byte vectors straight
//   from the accumulators -- no LDS transpose, no extra barriers -- and reduces the BatchNorm
//   sums in registers.
//   Block = WM x WN waves; wave (wm, wn) owns MT pixel tiles x NTW channel tiles of 16x16.
//   Pixel tiles are taken from "virtual rows": the batch is stacked along y with one zero row
//   between images, so small feature maps (12x9, 8x6) still fill the tile.
//   Pipeline: a stage = (tile, channel chunk).  While the MFMAs of stage s run out of LDS, the
//   global loads of stage s+1 are already in flight into registers (issue-early / write-late);
//   all per-vector index arithmetic is hoisted out of the stage loop.
//   LDS strides (96 B per pixel, taps*64+32 B per filter row) are conflict-free for ds_read_b128.
#include <stdlib.h>
#include <algorithm>
#include <cstring>
#include <type_traits>
#include "common.cuh"

namespace {

#ifndef STL_EPRE_S8
#define STL_EPRE_S8 0   // 1: epilogue operands of the C <= 32 data gradient requested ahead of the next stage's loads at three waves per SIMD -- measured slower (one-stream step 21.10 -> 21.53 ms, step 15.09 -> 15.38): the second resident block is worth more
#endif
constexpr int PSA = 96;
#ifndef STL_CONV_S8_CO_DEFAULT
#define STL_CONV_S8_CO_DEFAULT ""
#endif

// debug-only phase stamps (block 0, thread 0; enabled by STL_CONV_STAMPS=1): never read by the kernel
__device__ long long g_stamps[32];
__device__ long long g_stamps2[64];  // wave-specialised kernel: [0..23] loader, [32..55] compute (6 stages x 4)
// Compiled in only with -DSTL_STAMPS (python -m stlpose_amd.build --stamps -> libstlpose_hip_stamps.so): even when
// disabled at run time, the stores make the compiler place s_waitcnt vmcnt() in front of whatever reuses their data
// registers -- in the stage loop that drains loads which were meant to stay in flight.
#ifdef STL_STAMPS
#define STAMP(i)                                                          \
    do {                                                                  \
        if (k.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps[i] = wall_clock64(); \
    } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

struct ConvK {
    stl_conv p;
    int tiles_c, npt;
    int HR, HC, HP;
    int PI, pad, seff;
    int nchunks, wres;
    int cipad;
    int off_cs, off_cm, off_a, off_b, off_red;
    int sz_a, sz_b;  // wave-specialised kernel: byte distance between the two LDS buffers (0 = single)
    int TH, TW;
    int dbg;
    float r_HC, r_TW, r_tc, r_PI, r_vp;  // reciprocals for fdiv
    unsigned long long m_tc, m_PI, m_vp;  // ceil(2^32 / d): uniform n / d = (n * m) >> 32 on the scalar unit (sdiv)
    int ny;  // output-channel blocks per pixel tile (they are the FAST block dimension, see kernel)
    // fused weight gradient (p.wg_partial != NULL): LDS offsets of the h-slab halo tile / its BN constants,
    // bytes per h pixel, K steps (of 4*KV pixels) that cover the pixel tile
    int off_h, off_ch, psh, nks;
};

// Integer multiplies: v_mul_lo_u32 and v_mad_u64_u32 are quarter rate (16 cycles a wave), the 24-bit forms full rate.
// Element offsets are therefore built with 24-bit mads (tensors of up to 2^24 pixels and 2^31 elements: checked on the
// host) and the tile-uniform terms with integer arithmetic on the scalar unit.  Inline asm keeps the mads out of reach
// of LLVM's 64-bit mad combine.
__device__ __forceinline__ int mad24_vsv(int a, int b_uniform, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}
// a * b + c, a and b unsigned 24-bit (bits 24-31 ignored), b uniform
__device__ __forceinline__ int madu24_vsv(int a, int b_uniform, int c) {
    int r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}
__device__ __forceinline__ int mulu24_vs(int a, int b_uniform) {
    int r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(b_uniform), "v"(a));
    return r;
}
// n / d for uniform n >= 0 with m = ceil(2^32 / d) (exact for n * d < 2^32): integer only, so it stays on the scalar unit
__device__ __forceinline__ int sdiv(int n, unsigned long long m) { return (int)(((unsigned long long)(uint32_t)n * m) >> 32); }

// transposed LDS fragment reads for the fused weight gradient (K = pixels; same forms as wgrad.hip)
template <typename T>
__device__ __forceinline__ V16 cfrag_tr(const char* base, const int* rowoff, int colbyte, int lane);
template <>
__device__ __forceinline__ V16 cfrag_tr<__bf16>(const char* base, const int* rowoff, int colbyte, int lane) {
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
__device__ __forceinline__ V16 cfrag_tr<float>(const char* base, const int* rowoff, int colbyte, int lane) {
    V16 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v.w[s] = *reinterpret_cast<const uint32_t*>(base + rowoff[s] + colbyte + (lane & 15) * 4);
    return v;
}

template <typename T>
__device__ __forceinline__ void load4(const void* base, size_t elem, float* f) {
    if constexpr (sizeof(T) == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>((const char*)base + elem * 2);
        f[0] = __uint_as_float(v.x << 16), f[1] = __uint_as_float(v.x & 0xFFFF0000u);
        f[2] = __uint_as_float(v.y << 16), f[3] = __uint_as_float(v.y & 0xFFFF0000u);
    } else {
        const V16 v = ldg16((const char*)base + elem * 4);
        unpack<float>(v, f);
    }
}
template <typename T>
__device__ __forceinline__ void store4(void* base, size_t elem, const float* f) {
    if constexpr (sizeof(T) == 2) {
        uint2 v;
        v.x = f32_to_bf16(f[0]) | (f32_to_bf16(f[1]) << 16);
        v.y = f32_to_bf16(f[2]) | (f32_to_bf16(f[3]) << 16);
        *reinterpret_cast<uint2*>((char*)base + elem * 2) = v;
    } else {
        stg16((char*)base + elem * 4, pack<float>(f));
    }
}

#ifndef STL_PIN_TAPS
#define STL_PIN_TAPS 1
#endif
#include "conv_common.inc"

// WM x WN waves, MT pixel tiles and NTW channel tiles per wave; NVA staging vectors per thread for
// the input halo; Q: source is BNBWD (second tensor on load).
// WR: 1 = the filters are known to be LDS-resident (whole K in one chunk): the per-stage filter
// staging registers and descriptors do not exist in that instantiation (no spills at 128 VGPRs);
// -1 = decided at run time (k.wres)
// NCO > 0: FUSED BACKWARD of a 3x3 stride-1 C -> C convolution (NCO = C / CK input chunks of the data
// gradient = output-channel chunks of the forward conv).  The block is the 128-pixel x 32-channel data-gradient
// block (WM 4, WN 1, MT 2, NTW 2); in every stage (tile, chunk) it ALSO accumulates the weight gradient
// dw[co in chunk][tap][ci in the block's 32-channel slab] += sum_pixels g[pixel][co] * h[pixel + tap][ci] from
// the g halo tile it has staged for the data gradient anyway (read transposed: K = pixels) and a halo tile of
// the conv's forward input h (same halo geometry, staged once per tile).  dt and y are fetched once for both
// gradients, one launch instead of two; every wave owns whole output tiles (no cross-wave reduction) and keeps
// them in registers across the block's tiles; the block writes one split-K slab at the end.
// ZM: the source is STL_SRC_BNADD (Q = true: second tensor on load): the staged value is the residual block end
// z = ReLU(BN(x) + y); it is also written to p.src_out by the block that owns the pixel (tile interior, channel block 0).
// PE: plain epilogue -- the launch has no bias / addend / mask operand (conv_common.inc, epilogue_apply).
template <typename T, int KS, int WM, int WN, int MT, int NTW, int NVA, bool Q, bool PE, int OCC, int WR, int NCO = 0, bool ZM = false>
__global__ __launch_bounds__(64 * WM * WN, OCC) void conv_core_kernel(const ConvK k) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTHR = 64 * WM * WN;
    constexpr int KV = ET<T>::KV, CK = ET<T>::CK, TAPS = KS * KS;
    constexpr bool FW = NCO > 0;
    static_assert(!FW || (KS == 3 && WM == 4 && WN == 1 && MT == 2 && NTW == 2 && Q), "fused weight gradient: 3x3, 128 px x 32 channels, BNBWD source");
    static_assert(!ZM || (Q && KS == 3 && !FW), "block-end source: 3x3, two tensors on load");
    constexpr int NH = FW ? (sizeof(T) == 2 ? 1 : 2) : 1;   // 16-byte h vectors per staging slot (32 channels per pixel)
    constexpr int WKS = 4 * KV;                              // pixels per MFMA K step of the weight gradient
    constexpr int WNR = sizeof(T) == 2 ? 2 : 4;              // row offsets per transposed fragment
    constexpr int CT = CK / 16;                              // 16-channel tiles per chunk (2 bf16, 1 fp32)
    constexpr int NJ = CT == 2 ? TAPS : (TAPS + 1) / 2;      // weight-gradient accumulator tiles per wave and chunk
    constexpr int BCO = WN * NTW * 16;
    constexpr int ROWB = TAPS * 64 + 32;
    constexpr int NVB = (BCO * TAPS * 4 + NTHR - 1) / NTHR;
    const stl_conv& p = k.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int wm = wave % WM, wn = wave / WM;
    // Block order: XCD = blockIdx.x & 7; within an XCD the ny channel blocks of one pixel tile are
    // neighbours, so they run at the same time on the same L2: the input tile is fetched from HBM
    // once for all of them, and the ny pieces of every output row are written together.
    const int bl = blockIdx.x >> 3;
    const int lx = __builtin_amdgcn_readfirstlane(bl / k.ny), by = bl - lx * k.ny;   // (the division runs on the vector unit: back to scalars)
    const int n0 = by * BCO;
    const bool wres = WR < 0 ? (k.wres != 0) : (WR == 1);

    STAMP(0);
#ifdef STL_STAMPS
    if (k.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps[12] = __builtin_amdgcn_s_memtime();
#endif
    float* cs = reinterpret_cast<float*>(smem + k.off_cs);  // [3][cipad] source transform
    float* cm = reinterpret_cast<float*>(smem + k.off_cm);  // [4][BCO]   mask BN: a, b, mean, rstd
    char* sA = smem + k.off_a;
    char* sB = smem + k.off_b;

    // ---- per-channel constants.  (Computing them AFTER the first tile's loads have been issued, so that the
    // statistics round trip overlaps the tile's, measured SLOWER: 19.2 -> 23.1 us for the 96x72 C=32 layer --
    // hipcc sinks the tile loads below the constants code, sched_barrier or not.)
    float* hcs = reinterpret_cast<float*>(smem + k.off_ch);  // fused: [2][32] BN affine of the h slab
    char* sH = smem + k.off_h;
    auto compute_consts = [&]() {
        for (int c = tid; c < k.cipad; c += NTHR) {
            float a = 0.f, b = 0.f, cc = 0.f;
            if (c < p.Ci) src_consts(p.src, c, p.Ci, a, b, cc);
            cs[c] = a, cs[k.cipad + c] = b, cs[2 * k.cipad + c] = cc;
        }
        if (!PE && p.mask_y) {
            // the LAST wave computes the mask constants while the first one(s) do the source's: the two sets are
            // dependent global round trips each, and run in parallel on different waves instead of back to back
            for (int c = tid - (NTHR - 64); c >= 0 && c < BCO; c += 64) {
                float a = 0.f, b = 0.f, mu = 0.f, rs = 0.f;
                if (n0 + c < p.Co) {
                    bn_mean_rstd(p.mask_bn, n0 + c, p.Co, mu, rs);
                    a = p.mask_bn.gamma[n0 + c] * rs;
                    b = p.mask_bn.beta[n0 + c] - mu * a;
                }
                cm[c] = a, cm[BCO + c] = b, cm[2 * BCO + c] = mu, cm[3 * BCO + c] = rs;
            }
        }

        if constexpr (FW) {
            if (tid >= 64 && tid < 96) {   // wave 1 (see above)
                const int c = tid - 64;
                float a = 1.f, b = 0.f, cc = 0.f;
                if (p.wg_h.mode != STL_SRC_PLAIN) src_consts(p.wg_h, n0 + c, p.Co, a, b, cc);
                hcs[c] = a, hcs[32 + c] = b;
            }
            if (tid < PSA / 16) *reinterpret_cast<V16*>(sA + k.HP * PSA + tid * 16) = zero16();   // zero pixel row behind the halo tile
        }
    };
    compute_consts();
    STAMP(1);
    // ---- loop-invariant per-thread descriptors
    int a_rc[NVA];  // (halo row << 16) | halo col, -1 when this slot is unused
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int v = tid + i * NTHR;
        if (v < k.HP * 4) {
            const int hp = v >> 2, hr = fdiv(hp, k.r_HC);
            a_rc[i] = (hr << 16) | (hp - hr * k.HC);
        } else {
            a_rc[i] = -1;
        }
    }
    const int a_part = tid & 3;
    uint32_t a_int = 0;   // ZM: bit i = staging slot i is a pixel of the tile interior (this block stores its z)
    if constexpr (ZM) {
#pragma unroll
        for (int i = 0; i < NVA; ++i)
            if (a_rc[i] >= 0) {
                const int hr = a_rc[i] >> 16, hc = a_rc[i] & 0xffff;
                if (hr >= 1 && hr <= k.TH && hc >= 1 && hc <= k.TW && by == 0 && p.src_out) a_int |= 1u << i;
            }
    }
    int b_g[NVB], b_l[NVB];
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
        const int v = tid + i * NTHR;
        b_g[i] = -1, b_l[i] = 0;
        if (v < BCO * TAPS * 4) {
            const int n = v / (TAPS * 4), r = v - n * (TAPS * 4), tap = r >> 2, part = r & 3;
            b_l[i] = n * ROWB + tap * 64 + part * 16;
            if (n0 + n < p.Co) b_g[i] = ((n0 + n) * TAPS + tap) * p.Ci + part * KV;
        }
    }
    const int tilepx = k.TH * k.TW;
    int xoff[MT];   // LDS offset of this lane's pixel (MFMA column r16) in each of its pixel tiles
    int e_yx[MT];   // (ty << 16) | tx of that pixel, -1 when outside the tile
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        int m = (wm * MT + mi) * 16 + r16;
        e_yx[mi] = -1;
        if (m < tilepx) {
            const int ty = fdiv(m, k.r_TW);
            e_yx[mi] = (ty << 16) | (m - ty * k.TW);
        } else {
            m = 0;
        }
        const int ty = fdiv(m, k.r_TW), tx = m - ty * k.TW;
        xoff[mi] = ((ty * k.seff) * k.HC + tx * k.seff) * PSA + g * 16;
    }
    const int woff = (wn * NTW * 16 + r16) * ROWB + g * 16;

    STAMP(2);
    // NSET register sets of staged loads: the loads of stage s + NSET are issued while stage s is multiplied, so a
    // stage's memory round trip (2-5 us when every block of a launch asks at once; 45-49 % of a wave's cycles were spent
    // parked behind it with one set) is spread over NSET iterations.  One set for the fused backward (registers) and
    // for the 4-wave shapes compiled for three or four blocks per CU (they hide the latency with residency).
#ifndef STL_NSET2
#define STL_NSET2 0   // two sets measured neutral-to-slower (serial 21.39 vs 21.26 ms, step 15.23 vs 15.18): kept behind the macro
#endif
    constexpr int NSET = (!STL_NSET2 || Q || FW || (WM * WN == 4 && OCC >= 3) || (WM == 8 && MT == 2)) ? 1 : 2;   // Q: two tensors per slot -- the 8-wave data-gradient shapes sit at 249 of 256 registers with one set
    V16 ra[NSET][NVA], rq[NSET][Q ? NVA : 1], rb[NSET][NVB];
    V16 rh[FW ? NVA : 1][NH];
    int a_go[NSET][NVA];
    int st_t[NSET], st_ch[NSET];      // the stage (tile, chunk) each set holds
    bool st_have[NSET];

    auto tile_setup = [&](int t, int* go) {
        const int tr = sdiv(t, k.m_tc), tc = t - tr * k.tiles_c;
        const int vrs = tr * k.TH * k.seff, cb = tc * k.TW * k.seff - k.pad;
        const int b0 = sdiv(vrs, k.m_PI), y0 = vrs - b0 * k.PI - k.pad;
        const int n_PI = -k.PI;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int iy0 = y0 + (a_rc[i] >> 16);
            int ix = cb + (a_rc[i] & 0xffff);
            const int wr_ = fdiv(iy0 < 0 ? 0 : iy0, k.r_PI);
            int iy = mad24_vsv(wr_, n_PI, iy0);
            bool ok = (a_rc[i] >= 0) & (iy0 >= 0) & (ix >= 0) & (b0 + wr_ < p.B);
            if (p.stuff) {   // zero-stuffed source (transposed convolution): only even rows / columns exist
                ok = ok & !((iy | ix) & 1);
                iy >>= 1, ix >>= 1;
            }
            ok = ok & (iy < p.Hi) & (ix < p.Wi);
            const int e = madu24_vsv(madu24_vsv(madu24_vsv(b0 + wr_, p.Hi, iy), p.Wi, ix), p.Ci, a_part * KV);
            go[i] = ok ? e : -1;
        }
    };
    // Loads are UNCONDITIONAL (invalid slots read element 0 and are zeroed at write time): a
    // guarded load makes hipcc branch around it and wait vmcnt(0) per element, which serialises
    // the whole staging burst into dependent round trips.
    auto issue = [&](auto SET, const int* go, int k0, bool en) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        const bool chok = en && (k0 + a_part * KV) < p.Ci;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int off = (go[i] >= 0 && chok) ? go[i] + k0 : 0;
            ra[S][i] = ldg16((const char*)p.src.x + (size_t)(uint32_t)off * sizeof(T));
            if (Q) rq[S][i] = ldg16((const char*)p.src.y + (size_t)(uint32_t)off * sizeof(T));
            if constexpr (FW) {   // h slab of the same halo pixel (Ci == Co): only with the first chunk of a tile
                const int hoff = (go[i] >= 0 && en && k0 == 0) ? go[i] + n0 : 0;
#pragma unroll
                for (int j = 0; j < NH; ++j) rh[i][j] = ldg16((const char*)p.wg_h.x + (size_t)(hoff + 4 * KV * j) * sizeof(T));
            }
        }
        if (!wres) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                const bool ok = en && b_g[i] >= 0 && (k0 + ((tid + i * NTHR) & 3) * KV) < p.Ci;
                rb[S][i] = ldg16((const char*)p.w + (size_t)(ok ? b_g[i] + k0 : 0) * sizeof(T));
            }
        }
    };
    const float relu_lo = p.src.relu ? 0.f : -INFINITY;
    const float relu_lo_h = (FW && p.wg_h.relu) ? 0.f : -INFINITY;
    auto write_lds = [&](auto SET, const int* go, int k0) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        const int ch = k0 + a_part * KV;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            if (a_rc[i] < 0) continue;
            const bool ok = go[i] >= 0 && ch < p.Ci;
            const int chc = ok ? ch : 0;
            V16 val = ra[S][i];
            if constexpr (ZM) {
                val = xform_bnadd<T>(val, rq[S][i], cs + chc, cs + k.cipad + chc, relu_lo);
                if (ok && ((a_int >> i) & 1u)) stg16((char*)p.src_out + (size_t)(go[i] + k0) * sizeof(T), val);   // z, once per pixel
            } else if (Q)
                val = xform_bnbwd<T>(val, rq[S][i], cs + chc, cs + k.cipad + chc, cs + 2 * k.cipad + chc);
            else if (p.src.mode != STL_SRC_PLAIN)
                val = xform_bn<T>(val, cs + chc, cs + k.cipad + chc, relu_lo);
            mask16(val, ok);  // zero padding applies AFTER the transform
            const int v = tid + i * NTHR;
            *reinterpret_cast<V16*>(sA + (v >> 2) * PSA + (v & 3) * 16) = val;
            if constexpr (FW) {
                if (k0 == 0) {
#pragma unroll
                    for (int j = 0; j < NH; ++j) {
                        V16 hv = rh[i][j];
                        const int cl = (a_part + 4 * j) * KV;
                        if (p.wg_h.mode != STL_SRC_PLAIN) hv = xform_bn<T>(hv, hcs + cl, hcs + 32 + cl, relu_lo_h);
                        mask16(hv, go[i] >= 0);   // zero padding applies AFTER the transform
                        *reinterpret_cast<V16*>(sH + (v >> 2) * k.psh + (a_part + 4 * j) * 16) = hv;
                    }
                }
            }
        }
        if (!wres) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                const bool ok = b_g[i] >= 0 && (k0 + ((tid + i * NTHR) & 3) * KV) < p.Ci;
                V16 val = rb[S][i];
                mask16(val, ok);
                if (tid + i * NTHR < BCO * TAPS * 4) *reinterpret_cast<V16*>(sB + b_l[i]) = val;
            }
        }
    };

    if (wres) {  // whole K fits one chunk: filters stay resident in LDS for all tiles (rb[] is free then)
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const bool ok = b_g[i] >= 0 && (((tid + i * NTHR) & 3) * KV) < p.Ci;
            rb[0][i] = ldg16((const char*)p.w + (size_t)(ok ? b_g[i] : 0) * sizeof(T));
        }
    }

    // statistics accumulators: this lane's 4 channels of each of its NTW channel tiles
    f2v s0[NTW][2], s1[NTW][2];
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
        for (int h = 0; h < 2; ++h) s0[ni][h] = f2v{0.f, 0.f}, s1[ni][h] = f2v{0.f, 0.f};

    const int xcd = blockIdx.x & 7, nx = __builtin_amdgcn_readfirstlane((gridDim.x >> 3) / k.ny);
    const int T8 = (k.npt + 7) >> 3;
    const int vpitch = p.Ho + 1;

    // cursor of the NEXT stage to request: tiles it_n = lx, lx + nx, ... of this XCD's share, chunks 0 .. nchunks-1 of each
    int it_n = lx, t_n = xcd * T8 + lx, ch_n = 0;
    bool have_n = (it_n < T8) && (t_n < k.npt);
    constexpr std::integral_constant<int, 0> I0{};
    constexpr std::integral_constant<int, NSET - 1> I1{};
    // request the cursor's stage into set S, remember which stage the set now holds, advance the cursor
    auto prefetch = [&](auto SET) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        if (have_n) {
            if (ch_n == 0) {
                tile_setup(t_n, a_go[S]);
            } else if (NSET == 2 && ch_n == 1) {   // chunk 0 of this tile went to the other set: same tile, same offsets
#pragma unroll
                for (int i = 0; i < NVA; ++i) a_go[S][i] = a_go[S ^ (NSET - 1)][i];
            }
        }
        st_t[S] = t_n, st_ch[S] = ch_n, st_have[S] = have_n;
        issue(SET, a_go[S], ch_n * CK, have_n);
        if (++ch_n == k.nchunks) {
            ch_n = 0, it_n += nx, t_n = xcd * T8 + it_n;
            have_n = have_n && (it_n < T8) && (t_n < k.npt);
        }
    };
    STAMP(3);
    prefetch(I0);
    if constexpr (NSET == 2) prefetch(I1);
    if (wres) {
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const bool ok = b_g[i] >= 0 && (((tid + i * NTHR) & 3) * KV) < p.Ci;
            V16 val = rb[0][i];
            mask16(val, ok);
            if (tid + i * NTHR < BCO * TAPS * 4) *reinterpret_cast<V16*>(sB + b_l[i]) = val;
        }
    }
    __syncthreads();  // constants + resident filters visible
    STAMP(4);

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fused weight gradient: accumulators and the lane's transposed-read row offsets
    // wave w owns ci tile nt = w & 1 of the slab; bf16: co tile mt = w >> 1 of the chunk, all 9 taps;
    // fp32 (one co tile per chunk): taps 2j + (w >> 1).
    f32x4 accw[FW ? NCO : 1][FW ? NJ : 1];
    const int w_nt = wave & 1, w_hw = wave >> 1;
    if constexpr (FW) {
#pragma unroll
        for (int c = 0; c < NCO; ++c)
#pragma unroll
            for (int j = 0; j < NJ; ++j) accw[c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // one chunk's weight-gradient MFMAs out of the staged tiles.  The K-step loop is a real loop (row offsets
    // recomputed per step, ~20 VALU against 9+ MFMAs): unrolled, hipcc hoists every fragment read of the tile
    // and spills ~70 registers.
    auto wg_mfma = [&](f32x4* aw) {
        const int acol = (CT == 2 ? w_hw : 0) * 16 * (int)sizeof(T), bcol = w_nt * 16 * (int)sizeof(T);
#pragma unroll 1
        for (int s_ = 0; s_ < k.nks; ++s_) {
            int og[WNR], oh[WNR];   // rows of this lane's pixels in the g / h halo tiles
#pragma unroll
            for (int i = 0; i < WNR; ++i) {
                int m;
                if constexpr (sizeof(T) == 2)
                    m = s_ * WKS + 8 * g + 4 * i + (r16 >> 2);
                else
                    m = s_ * WKS + 4 * g + i;
                const bool in = m < tilepx;
                const int mm = in ? m : 0;
                const int ty = fdiv(mm, k.r_TW), tx = mm - ty * k.TW;
                og[i] = in ? ((ty + 1) * k.HC + tx + 1) * PSA : k.HP * PSA;   // beyond the tile: the zero row
                oh[i] = (ty * k.HC + tx) * k.psh;                             // tap (0, 0) corner of the pixel's 3x3 window
            }
            // all fragment reads of the K step are issued back to back, the MFMAs follow behind counted waits
            // (one read pair per MFMA with a full wait in between costs an LDS round trip per MFMA: 2 us per tile)
            const V16 af = cfrag_tr<T>(sA, og, acol, lane);
            V16 bq[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int tap = CT == 2 ? j : 2 * j + w_hw;
                const int tc = tap < TAPS ? tap : 0;
                bq[j] = cfrag_tr<T>(sH + ((tc / KS) * k.HC + (tc % KS)) * k.psh, oh, bcol, lane);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int tap = CT == 2 ? j : 2 * j + w_hw;
                if (tap < TAPS) mma16<T>(aw[j], af, bq[j]);
            }
        }
    };

    // epilogue operands (addend / masks) of all MT tiles of the wave, fetched in one burst: before the MFMAs of the
    // tile's last chunk where the block owns its CU anyway (EPRE: 12 registers per tile), else at the start of the epilogue
        constexpr bool EPRE = ((OCC <= 1 && WM == 4 && WN == 2 && NTW == 2) || (STL_EPRE_S8 && OCC <= 3 && WM == 8 && WN == 1 && MT == 2 && NTW == 2)) &&
                          Q && !PE && !FW && !ZM && sizeof(T) == 2;   // fp32 would need 96 registers and spills
    // element index (times Co) of this lane's output pixel in pixel tile mi of the tile at (vr0, c0); pok = inside the image
    auto out_pixel = [&](int vr0, int c0, int mi, bool& pok) __attribute__((always_inline)) -> size_t {
        const int eb0 = sdiv(vr0, k.m_vp), ey0 = vr0 - eb0 * vpitch;
        const int ty = e_yx[mi] >> 16, tx = e_yx[mi] & 0xffff;
        const int oy0 = ey0 + ty, c = c0 + tx;
        const int wr_ = fdiv(oy0, k.r_vp);
        const int oy = mad24_vsv(wr_, -vpitch, oy0), b = eb0 + wr_;
        pok = (e_yx[mi] >= 0) & (b < p.B) & (oy < p.Ho) & (c < p.Wo);
        const int e = mulu24_vs(madu24_vsv(madu24_vsv(b, p.Ho, oy), p.Wo, c), p.Co);
        return pok ? (size_t)(uint32_t)e : 0;
    };
    auto epi_fetch = [&](int vr0, int c0, bool* pokv, size_t* pixv, EpiRaw<NTW>* er) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            pixv[mi] = out_pixel(vr0, c0, mi, pokv[mi]);
            epilogue_fetch<T, NTW, true>(p, pokv[mi], pixv[mi], n0, wn * NTW * 16, g, er[mi]);
        }
    };

    // flat loop over stages (tile, chunk), unrolled over the register sets; exactly ONE issue() site per set inside
    // the loop so that the staging registers need no PHI copies (which would force a vmcnt(0) before the MFMAs)
    auto stage = [&](auto SET) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        const int t = st_t[S], ch0 = st_ch[S];
        const int tr = sdiv(t, k.m_tc), tc = t - tr * k.tiles_c;
        const int vr0 = tr * k.TH, c0 = tc * k.TW;
        write_lds(SET, a_go[S], ch0 * CK);
        __syncthreads();
        if (ch0 == 0) STAMP(5);
        const bool last_chunk = (ch0 + 1 == k.nchunks);
        bool pokv[EPRE ? MT : 1];
        size_t pixv[EPRE ? MT : 1];
        EpiRaw<NTW> er[EPRE ? MT : 1];
        if constexpr (EPRE) {
            if (last_chunk) epi_fetch(vr0, c0, pokv, pixv, er);   // ahead of the next loads: consumed behind a counted wait
        }
        prefetch(SET);  // stage s + NSET into the set just drained: its loads land during the MFMAs of this and the next stage
        if (ch0 == 0) STAMP(6);
        {  // fragment reads of tap t+1 are issued before the MFMAs of tap t (static double buffer).  PIN: the order is
           // pinned with scheduling barriers -- left alone, the scheduler sinks every read to just in front of its
           // first use to save registers, and each MFMA pair then waits a full LDS round trip
            constexpr bool PIN = (STL_PIN_TAPS != 0) && !FW;
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
                if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NTW; ++ni) mma16<T>(acc[mi][ni], wf[tap & 1][ni], xf[tap & 1][mi]);
                if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (FW) {
#pragma unroll
            for (int c = 0; c < NCO; ++c)
                if (ch0 == c) wg_mfma(accw[c]);   // static accumulator index per chunk
        }
        __syncthreads();  // everyone is done with sA/sB of this stage
        if (ch0 == 0) STAMP(7);
        if (last_chunk) {
            STAMP(8);
            // ---- epilogue straight from the accumulators: lane = pixel r16, 4 channels per tile
            if constexpr (EPRE) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    epilogue_apply<T, NTW, BCO>(p, acc[mi], cm, pokv[mi], pixv[mi], n0, wn * NTW * 16, g, s0, s1, er[mi]);
            } else {   // register-tight instantiations: tile by tile
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    bool pok;
                    const size_t pix = out_pixel(vr0, c0, mi, pok);
                    epilogue_tile<T, NTW, BCO, PE>(p, acc[mi], cm, pok, pix, n0, wn * NTW * 16, g, s0, s1);
                }
            }
        }
        if (last_chunk) STAMP(9);
    };
    while (true) {
        if (!st_have[0]) break;
        stage(I0);
        if constexpr (NSET == 2) {
            if (!st_have[1]) break;
            stage(I1);
        }
    }
    STAMP(10);
    // ---- flush statistics: lanes of one 16-lane group hold the same channels -> xor-reduce them,
    // then the WM waves of a channel column through LDS, then one fp64 atomic per channel
    double* dst = p.out_stats ? p.out_stats : p.red;
    if (dst) {
        xor_reduce_stats<NTW>(s0, s1);
        float* red = reinterpret_cast<float*>(smem + k.off_red);  // [WM][2][BCO]
        __syncthreads();
        if (r16 == 0) {
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cl = (wn * NTW + ni) * 16 + 4 * g + r;
                    red[(wm * 2 + 0) * BCO + cl] = s0[ni][r >> 1][r & 1];
                    red[(wm * 2 + 1) * BCO + cl] = s1[ni][r >> 1][r & 1];
                }
        }
        __syncthreads();
        for (int e = tid; e < 2 * BCO; e += NTHR) {
            const int which = e / BCO, cl = e - which * BCO;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) s += red[(w * 2 + which) * BCO + cl];
            if (n0 + cl < p.Co)
                atomicAdd(dst + (size_t)(blockIdx.x & (STL_NSHARD - 1)) * 2 * p.Co + which * p.Co + n0 + cl, (double)s);
        }
    }
    if constexpr (FW) {
        // ---- weight-gradient slab of this block's pixel group: partial[sid][co][tap][ci], fp32
        // accw[c][j][r] = dw[co = c*CK + mt*16 + 4g + r][tap_j][ci = n0 + nt*16 + r16]
        const int sid = xcd * nx + lx;
        float* slab = p.wg_partial + (size_t)sid * p.Ci * TAPS * p.Co;
        const int ci = n0 + w_nt * 16 + r16;
#pragma unroll
        for (int c = 0; c < NCO; ++c) {
            const int co = c * CK + (CT == 2 ? w_hw : 0) * 16 + 4 * g;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int tap = CT == 2 ? j : 2 * j + w_hw;
                if (tap < TAPS) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[((size_t)(co + r) * TAPS + tap) * p.Co + ci] = accw[c][j][r];
                }
            }
        }
    }
    STAMP(11);
#ifdef STL_STAMPS
    if (k.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps[13] = __builtin_amdgcn_s_memtime();
#endif
}

#include "conv_ws.inc"
#include "conv1x1.inc"

template <typename T, int KS, int WM, int WN, int MT, int NTW, int NVA, bool Q, bool PE, int OCC = 1, int WR = -1, int NCO = 0, bool ZM = false>
int launch(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_core_kernel<T, KS, WM, WN, MT, NTW, NVA, Q, PE, OCC, WR, NCO, ZM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_core_kernel<T, KS, WM, WN, MT, NTW, NVA, Q, PE, OCC, WR, NCO, ZM>), grid, dim3(64 * WM * WN), lds, st, k);
    STL_LAUNCH_CHECK("conv_core");
    return 0;
}

// fused data + weight gradient (NCO chunks of CK channels): the 128 px x 32 channel block shape only
template <typename T>
int dispatch_fused(int nco, const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
    switch (nco) {
        case 1: return launch<T, 3, 4, 1, 2, 2, 3, true, false, 2, 1, 1>(k, grid, lds, st);    // bf16 C = 32: filters resident
        case 2: return launch<T, 3, 4, 1, 2, 2, 3, true, false, 2, -1, 2>(k, grid, lds, st);   // bf16 C = 64 / fp32 C = 32
        case 4: return launch<T, 3, 4, 1, 2, 2, 3, true, false, 1, -1, 4>(k, grid, lds, st);   // fp32 C = 64
    }
    return stl_set_error("conv(fused): %d chunks not supported", nco);
}

// block shapes: 0 = 128 px x 64 co (4 waves), 1 = 512 px x 32 co, 2 = 256 px x 64 co, 3 = 256 px x 128 co (8 waves),
// 4 = 128 px x 32 co (4 waves, <=128 VGPRs, <=40 KB LDS: four blocks per CU hide each other's latency)
struct Shape {
    int px, co, thr;   // pixels / output channels per block, threads
    int ws;            // 1: wave-specialised kernel (4 loader + 4 compute waves, double-buffered LDS)
    int lthr, nva_max; // threads that stage the halo, max staging vectors per such thread
};
// 0..4: uniform-wave kernel (conv_core_kernel); 5..7: wave-specialised kernel (conv_ws_kernel)
constexpr int NSHAPES = 10;  // 8 = 256 px x 32 co (8 waves: two 128-pixel halves share one filter copy and one halo tile); 9 = wave-specialised 128 px x 32 co
constexpr Shape SHAPES[NSHAPES] = {{128, 64, 256, 0, 256, 9}, {512, 32, 512, 0, 512, 6}, {256, 64, 512, 0, 512, 3},
                                   {256, 128, 512, 0, 512, 3}, {128, 32, 256, 0, 256, 6},
                                   {512, 32, 512, 1, 256, 10}, {256, 64, 512, 1, 256, 6}, {128, 64, 512, 1, 256, 9},
                                   {256, 32, 512, 0, 512, 3}, {128, 32, 512, 1, 256, 3}};

template <typename T, int KS, bool Q, bool PE>
int dispatch(int shape, int nva, const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
    switch (shape) {
        case 0:
            if (nva <= 3) return launch<T, KS, 4, 1, 2, 4, 3, Q, PE>(k, grid, lds, st);
            if (nva <= 9) return launch<T, KS, 4, 1, 2, 4, 9, Q, PE>(k, grid, lds, st);
            break;
        case 1:
            if (nva <= 6) return launch<T, KS, 8, 1, 4, 2, 6, Q, PE>(k, grid, lds, st);
            break;
        case 2:
#ifndef STL_S2Q_OCC
#define STL_S2Q_OCC 1   // waves per SIMD the register budget of the 256 px x 64 co DATA-GRADIENT kernel is sized for (1: up to 256 VGPRs, one block owns
                        // the CU; 3 = 168 VGPRs + 49 spilled, 4 = 128 + 160 spilled: 15.6 / 18.2 vs 14.64 ms per step -- leaving room for other
                        // kernels' waves needs a smaller per-wave tile, not a register cap)
#endif
            if (nva <= 3) return launch<T, KS, 4, 2, 4, 2, 3, Q, PE, (Q ? STL_S2Q_OCC : 1)>(k, grid, lds, st);
            break;
        case 3:
            if (nva <= 3) return launch<T, KS, 4, 2, 4, 4, 3, Q, PE>(k, grid, lds, st);
            break;
        case 8:   // forward (two register sets of staged loads): three waves per SIMD, no spills
            if (nva <= 3 && k.wres) return launch<T, KS, 8, 1, 2, 2, 3, Q, PE, ((Q && STL_EPRE_S8) ? 3 : 4), 1>(k, grid, lds, st);
            if (nva <= 3) return launch<T, KS, 8, 1, 2, 2, 3, Q, PE, (Q ? 3 : 4)>(k, grid, lds, st);
            break;
        case 4:
            if (nva <= 3 && k.wres) return launch<T, KS, 4, 1, 2, 2, 3, Q, PE, 4, 1>(k, grid, lds, st);  // 113-123 VGPRs, no spills
            if (nva <= 3) return launch<T, KS, 4, 1, 2, 2, 3, Q, PE, (Q ? 3 : 4)>(k, grid, lds, st);
            if (nva <= 6) return launch<T, KS, 4, 1, 2, 2, 6, Q, PE, 3>(k, grid, lds, st);
            break;
        case 5:
            if (nva <= 10) return launch_ws<T, KS, 8, 2, 10, Q>(k, grid, lds, st);
            break;
        case 6:
            if (nva <= 6) return launch_ws<T, KS, 4, 4, 6, Q>(k, grid, lds, st);
            break;
        case 7:
            if (nva <= 3) return launch_ws<T, KS, 2, 4, 3, Q>(k, grid, lds, st);
            if (nva <= 9) return launch_ws<T, KS, 2, 4, 9, Q>(k, grid, lds, st);
            break;
        case 9:   // 128 px x 32 co: twice the blocks of shape 7 for the small, deep maps (12x9 at C = 256: 128 -> 256 blocks)
            if (nva <= 3) return launch_ws<T, KS, 2, 2, 3, Q>(k, grid, lds, st);
            break;
    }
    return stl_set_error("conv: no kernel variant for block shape %d with %d staging vectors per thread", shape, nva);
}

// block-end source (STL_SRC_BNADD): the shapes the two-conv units of the network are planned with -- 256 px x 32 co
// (C <= 32), 256 px x 64 co (C = 64 / 128) and their small-map fallbacks; own instantiations, so that the
// data-gradient kernels (Q without ZM) carry none of this
static bool bnadd_shape_ok(int shape, int nva) { return nva <= 3 && (shape == 0 || shape == 2 || shape == 4 || shape == 8); }
template <typename T>
int dispatch_bnadd(int shape, int nva, const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
    if (nva <= 3) switch (shape) {
        case 0: return launch<T, 3, 4, 1, 2, 4, 3, true, true, 1, -1, 0, true>(k, grid, lds, st);
        case 2: return launch<T, 3, 4, 2, 4, 2, 3, true, true, 1, -1, 0, true>(k, grid, lds, st);
        case 4: return launch<T, 3, 4, 1, 2, 2, 3, true, true, 3, -1, 0, true>(k, grid, lds, st);
        case 8: return launch<T, 3, 8, 1, 2, 2, 3, true, true, 3, -1, 0, true>(k, grid, lds, st);
    }
    return stl_set_error("conv: no block-end (BNADD) variant for block shape %d with %d staging vectors per thread", shape, nva);
}

struct Plan {
    int shape, TH, TW;
    size_t lds;
    double cost;
};

// LDS bytes of a candidate (consts + input halo + filters + stats scratch)
size_t lds_bytes(const stl_conv& p, int shape, int TH, int TW, int ck, ConvK* out) {
    const int taps = p.ks * p.ks, seff = p.stride;
    const int HR = (TH - 1) * seff + p.ks, HC = (TW - 1) * seff + p.ks;
    const int nchunks = ceil_div(p.Ci, ck), cipad = nchunks * ck;
    const int bco = SHAPES[shape].co, ws = SHAPES[shape].ws;
    int off = 3 * cipad * 4;
    const int off_cm = off;
    off += 4 * bco * 4;
    off = (off + 15) & ~15;
    const int off_a = off;
    const bool fused = p.wg_partial != nullptr;
    const int sz_a = (((HR * HC + (fused ? 1 : 0)) * PSA) + 15) & ~15;   // fused: + one zero pixel row
    off += sz_a * (ws ? 2 : 1);
    const int off_b = off;
    const int sz_b = bco * (taps * 64 + 32);
    // wave-specialised kernel: keep the whole filter slab (all chunks) resident when it fits
    const bool resident = nchunks == 1 || (ws && (size_t)off + (size_t)sz_b * nchunks <= 150 * 1024);
    off += resident ? sz_b * nchunks : sz_b * (ws ? 2 : 1);
    const int off_red = off_a;  // reused after the last stage
    const int red = 8 * 2 * bco * 4;
    if (off - off_a < red) off = off_a + red;
    int off_ch = 0, off_h = 0;
    const int psh = 32 * (ck == 32 ? 2 : 4) + 32;   // 32 channels per h pixel (64 B bf16 / 128 B fp32) + 32 B pad
    if (fused) {
        off = (off + 15) & ~15;
        off_ch = off;
        off += 2 * 32 * 4;
        off_h = off;
        off += HR * HC * psh;
    }
    if (out) {
        out->off_ch = off_ch, out->off_h = off_h, out->psh = psh;
        out->HR = HR, out->HC = HC, out->HP = HR * HC, out->nchunks = nchunks, out->cipad = cipad;
        out->off_cs = 0, out->off_cm = off_cm, out->off_a = off_a, out->off_b = off_b, out->off_red = off_red;
        out->sz_a = ws ? sz_a : 0, out->sz_b = (ws && !resident) ? sz_b : 0;
        out->wres = resident ? 1 : 0;
    }
    return (size_t)off;
}

Plan choose_plan(const stl_conv& p, int ck) {
    Plan best{-1, 0, 0, 0, 1e300};
    const int vrows = p.B * (p.Ho + 1);
    for (int shape = 0; shape < NSHAPES; ++shape) {
        const Shape sh = SHAPES[shape];
        if (sh.px > 128 && p.stride == 2) continue;  // stride-2 halos only fit the small blocks
        // 256 px x 128 co blocks (every input pixel staged once for C = 128): spills ~30 registers; STL_CONV_S3_CO=128 tries it
        static const int s3_co = getenv("STL_CONV_S3_CO") ? atoi(getenv("STL_CONV_S3_CO")) : 0;
        const bool s3 = s3_co > 0 && p.Co == s3_co && p.Ci == s3_co && p.stride == 1 && p.ks == 3 && !p.stuff && !p.wg_partial;
        if (shape == 3 && !s3) continue;
        if (s3 && shape != 3) continue;
        // 128 px x 32 co blocks (four per CU) win in isolation for the C<=32 3x3 layers (21.7 vs 28.2 us)
        const int s4_maxco = getenv("STL_CONV_SHAPE4_MAXCO") ? atoi(getenv("STL_CONV_SHAPE4_MAXCO")) : 32;
        // ... except with many input channels on a large map (transition1: 256 -> 32 at 96x72, 8 chunks of K per tile): the
        // 512 px x 32 co block (8 waves) stages each filter chunk once per 512 pixels instead of once per 128 (105.8 -> 71.8 us)
        const bool wide_k = p.Co <= 32 && p.Ci >= 128 && p.stride == 1 && p.ks == 3 && !p.stuff && !p.wg_partial && (int64_t)p.B * p.Ho * p.Wo >= 65536;
        // STL_CONV_S8_CO="128[,64]": output-channel counts above 32 that also take the 256 px x 32 co blocks (ny = Co / 32
        // channel blocks per pixel tile; alone 24x18 C = 128 runs 14.2 us instead of 17.9 us: 232 blocks instead of 116)
        static const char* s8_co = getenv("STL_CONV_S8_CO") ? getenv("STL_CONV_S8_CO") : STL_CONV_S8_CO_DEFAULT;
        bool s8_extra = false;
        {
            char want[16];
            snprintf(want, sizeof(want), ",%d,", p.Co);
            char have[64];
            snprintf(have, sizeof(have), ",%s,", s8_co);
            s8_extra = strstr(have, want) != nullptr && p.stride == 1 && p.ks == 3 && !p.stuff && !p.wg_partial && p.Ci == p.Co &&
                       (!getenv("STL_CONV_S8_FWD_ONLY") || p.src.mode != STL_SRC_BNBWD);
        }
        if ((shape == 1 || shape == 5 || (shape == 8 && !s8_extra)) && p.Co > 32) continue;
        const bool c32 = p.wg_partial || s8_extra || (!getenv("STL_CONV_NO_C32_SHAPE4") && p.Co <= s4_maxco && p.stride == 1 && p.ks == 3 && !wide_k);
        // ... and 256 px x 32 co blocks of 8 waves (shape 8) win END TO END although they lose in isolation (C = 32 at
        // 96x72: 19.8 vs 17.9 us alone, 16.27 vs 16.47 ms per step): two 128-pixel halves share one copy of the filters
        // and one halo tile, i.e. fewer bytes per launch and half as many blocks competing for the CUs.  Shape 4 stays
        // the fallback where no 256-pixel tile fits (small maps) and for the fused backward.
        static const int c32shape = getenv("STL_CONV_C32_SHAPE") ? atoi(getenv("STL_CONV_C32_SHAPE")) : 8;
        if ((shape == 4 || shape == 8) && !c32) continue;
        if (c32 && shape != 4 && shape != 8 && !(shape == 1 && c32shape == 1)) continue;
        if (c32 && shape == 8 && (p.wg_partial || c32shape != 8)) continue;
        if (wide_k && shape != 1) continue;
        // kernel family: measured on MI355X (tools/conv_probe6.py; end to end the threshold Co >= 256 is the better one) the wave-specialised kernel wins for
        // stride-2 convs and for the small, deep maps (Co >= 256), the uniform kernel elsewhere
        const int ws_minco = getenv("STL_CONV_WS_MINCO") ? atoi(getenv("STL_CONV_WS_MINCO")) : 256;
        const int want_ws = p.wg_partial ? 0 : getenv("STL_CONV_WS") ? atoi(getenv("STL_CONV_WS"))
                                                  : ((p.stride == 2 || (p.ks == 3 && p.Co >= ws_minco && (int64_t)p.B * p.Ho * p.Wo <= 16384)) ? 1 : 0);
        if (sh.ws != want_ws) continue;
        static const int ws_co32 = getenv("STL_CONV_WS_CO32") ? atoi(getenv("STL_CONV_WS_CO32")) : 256;   // deep small maps on 128 px x 32 co blocks
        const bool deep_small = ws_co32 && p.stride == 1 && p.ks == 3 && p.Co >= ws_co32 && (int64_t)p.B * p.Ho * p.Wo <= 16384;
        // (in the step the chains of the low-resolution branches are the critical path of stage 4 -- tools/alone_time.py,
        // DESIGN.md 6.0 -- and their launches had 128 blocks: C = 256 at 12x9 on 128 x 32 blocks 14.95 -> 14.77 ms per step;
        // 64 x 64 and 64 x 32 blocks 15.10 / 14.94; the same for C = 128 via STL_CONV_S8_CO or the wave-specialised kernel: slower)
        if (shape == 9 && !deep_small) continue;
        if (deep_small && want_ws && shape != 9) continue;
        if (want_ws && !getenv("STL_CONV_WS") && shape != 7 && shape != 9) continue;
        static const char* s0_co = getenv("STL_CONV_S0_CO");   // experiment: output-channel counts that take the 128 px x 64 co 4-wave blocks
        if (s0_co && atoi(s0_co) == p.Co && p.Ci == p.Co && p.stride == 1 && p.ks == 3 && !p.stuff && !p.wg_partial && !want_ws && shape != 0) continue;
        const int nblk_co = ceil_div(p.Co, sh.co);
        for (int tw = (p.Wo < 4 ? p.Wo : 4); tw <= p.Wo && tw <= sh.px; ++tw)
          for (int frac = 4; frac >= 1; --frac) {
            int th = (sh.px / tw) * frac / 4;
            if (th > vrows) th = vrows;
            if (th < 1) continue;
            const int hr = (th - 1) * p.stride + p.ks, hc = (tw - 1) * p.stride + p.ks;
            const int nva = ceil_div(hr * hc * 4, sh.lthr);
            if (nva > sh.nva_max) continue;
            if (p.wg_partial && nva > 3) continue;   // fused backward: one instantiation (3 staging vectors per thread)
            const size_t lds = lds_bytes(p, shape, th, tw, ck, nullptr);
            if (shape == 4 && !p.wg_partial && lds > 40 * 1024 && nva <= 3) continue;  // keep four blocks per CU
            if (shape == 8 && lds > 80 * 1024) continue;                               // ... resp. two 8-wave blocks
            if (lds > 158 * 1024) continue;
            const double tiles = (double)ceil_div(vrows, th) * ceil_div(p.Wo, tw);
            // cost model (arbitrary units): MFMA work of all launched tiles (padding included), the
            // bytes each tile stages (halo + filters), and a penalty when few blocks exist
            const double mfma = tiles * sh.px * sh.co * nblk_co * (double)p.Ci * p.ks * p.ks / 2048.0;
            const double bytes = tiles * nblk_co * ((double)hr * hc * p.Ci + (double)sh.co * p.ks * p.ks * p.Ci) * 2.0 / 12.0;
            const double blocks = tiles * nblk_co;
            double cost = (mfma > bytes ? mfma : bytes) + 0.3 * (mfma < bytes ? mfma : bytes);
            const double waves = blocks * sh.thr / 64.0;
            if (waves < 2048.0) cost *= 1.0 + 0.15 * (2048.0 / waves - 1.0 > 4.0 ? 4.0 : 2048.0 / waves - 1.0);
            if (c32 && shape == 4 && c32shape != 4 && !p.wg_partial) cost *= 4.0;   // fallback only
            if (cost < best.cost) best = Plan{shape, th, tw, lds, cost};
        }
    }
    return best;
}

}  // namespace

extern "C" int stl_debug_conv_stamps(long long* host12) {
    return hipMemcpyFromSymbol(host12, HIP_SYMBOL(g_stamps), 14 * sizeof(long long)) == hipSuccess ? 0 : stl_set_error("stamps: copy failed");
}
extern "C" int stl_debug_conv_stamps2(long long* host64) {
    return hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_stamps2), 64 * sizeof(long long)) == hipSuccess ? 0 : stl_set_error("stamps: copy failed");
}

extern "C" int stl_conv_plan(stl_conv* pp) {
    stl_conv& p = *pp;
    STL_CHECK(p.dtype == STL_F32 || p.dtype == STL_BF16, "conv_plan: bad dtype");
    STL_CHECK((p.ks == 1 || p.ks == 3) && (p.stride == 1 || p.stride == 2) && p.Ci > 0 && p.Co > 0, "conv_plan: bad geometry");
    const int ck = p.dtype == STL_BF16 ? 32 : 16;
    Plan plan = choose_plan(p, ck);
    if (const char* e = getenv("STL_CONV_SHAPE")) {  // tuning knob: force a block shape where legal
        const int f = atoi(e);
        if (f >= 0 && f < NSHAPES && !(SHAPES[f].px > 128 && p.stride == 2)) {
            Plan best{-1, 0, 0, 0, 1e300};
            const int vrows = p.B * (p.Ho + 1);
            const Shape sh = SHAPES[f];
            for (int tw = (p.Wo < 4 ? p.Wo : 4); tw <= p.Wo && tw <= sh.px; ++tw) {
                int th = sh.px / tw;
                if (th > vrows) th = vrows;
                const int hr = (th - 1) * p.stride + p.ks, hc = (tw - 1) * p.stride + p.ks;
                const int nva = ceil_div(hr * hc * 4, sh.lthr);
                if (nva > sh.nva_max) continue;
                const size_t l = lds_bytes(p, f, th, tw, ck, nullptr);
                if (l > 158 * 1024) continue;
                const double waste = (double)ceil_div(vrows, th) * th * ceil_div(p.Wo, tw) * tw * (double)hr * hc / (th * tw);
                if (waste < best.cost) best = Plan{f, th, tw, l, waste};
            }
            if (best.shape >= 0) plan = best;
        }
    }
    STL_CHECK(plan.shape >= 0, "conv_plan: no tile fits LDS for %dx%d ks %d stride %d Ci %d", p.Ho, p.Wo, p.ks, p.stride, p.Ci);
    p.shape = plan.shape, p.TH = plan.TH, p.TW = plan.TW;
    return 0;
}

extern "C" int stl_conv_bnadd_ok(const stl_conv* pp) {
    const stl_conv& p = *pp;
    if (!(p.ks == 3 && p.stride == 1 && !p.stuff && p.shape >= 0 && p.shape < NSHAPES && p.TH > 0 && p.TW > 0) || use_1x1(p)) return 0;
    const Shape sh = SHAPES[p.shape];
    const int nva = ceil_div((p.TH + 2) * (p.TW + 2) * 4, sh.lthr);
    return bnadd_shape_ok(p.shape, nva) ? 1 : 0;
}

extern "C" int stl_conv_forward(const stl_conv* pp, void* stream) {
    const stl_conv& p = *pp;
    STL_CHECK(p.dtype == STL_F32 || p.dtype == STL_BF16, "conv: bad dtype %d", p.dtype);
    STL_CHECK(p.ks == 1 || p.ks == 3, "conv: ks must be 1 or 3 (got %d)", p.ks);
    STL_CHECK(p.stride == 1 || p.stride == 2, "conv: stride must be 1 or 2");
    STL_CHECK(!(p.stuff && p.stride != 1), "conv: stuff requires stride 1");
    STL_CHECK(!(p.ks == 1 && (p.stride != 1 || p.stuff)), "conv: 1x1 must be stride 1");
    const int kv = p.dtype == STL_BF16 ? 8 : 4, ck = 4 * kv;
    STL_CHECK(p.Ci % kv == 0 && p.Ci > 0, "conv: Ci=%d must be a multiple of %d", p.Ci, kv);
    STL_CHECK(p.Co % 8 == 0 && p.Co > 0, "conv: Co=%d must be a multiple of 8", p.Co);
    STL_CHECK(p.B > 0 && p.Hi > 0 && p.Wi > 0 && p.Ho > 0 && p.Wo > 0, "conv: empty tensor");
    STL_CHECK((int64_t)p.B * p.Hi * p.Wi * p.Ci < (1ll << 31) && (int64_t)p.B * p.Ho * p.Wo * p.Co < (1ll << 31),
              "conv: tensors of 2^31 or more elements are not supported");
    const int pad = p.ks == 3 ? 1 : 0;
    if (p.stuff) {
        STL_CHECK((p.Ho + 1) / 2 == p.Hi && (p.Wo + 1) / 2 == p.Wi, "conv(stuff): %dx%d is not the stride-2 image of %dx%d", p.Hi, p.Wi, p.Ho, p.Wo);
    } else {
        STL_CHECK((p.Hi + 2 * pad - p.ks) / p.stride + 1 == p.Ho && (p.Wi + 2 * pad - p.ks) / p.stride + 1 == p.Wo,
                  "conv: output %dx%d inconsistent with input %dx%d ks %d stride %d", p.Ho, p.Wo, p.Hi, p.Wi, p.ks, p.stride);
    }
    STL_CHECK(p.src.x && p.w && p.out, "conv: null tensor pointer");
    STL_CHECK(p.src.mode >= 0 && p.src.mode <= 3, "conv: bad src mode");
    const bool zm = p.src.mode == STL_SRC_BNADD;
    STL_CHECK(!zm || (p.ks == 3 && p.stride == 1 && !p.stuff && p.src.y && p.src.beta && (p.src.stats || (p.src.rmean && p.src.rvar)) && !p.wg_partial),
              "conv: a BNADD source needs a 3x3 stride-1 convolution, the skip tensor in src.y and BatchNorm parameters");
    STL_CHECK(zm || !p.src_out, "conv: src_out needs a BNADD source");
    STL_CHECK(!zm || (!p.bias && !p.addend && !p.mask_y && !p.mask_z && !p.red), "conv: a BNADD source takes no epilogue operands");
    STL_CHECK(p.src.mode == STL_SRC_PLAIN || p.src.gamma, "conv: BN source without gamma");
    STL_CHECK(p.src.mode != STL_SRC_BN || p.src.beta, "conv: BN source without beta");
    STL_CHECK(p.src.mode != STL_SRC_BN || p.src.stats || (p.src.rmean && p.src.rvar), "conv: BN source without statistics");
    STL_CHECK(p.src.mode != STL_SRC_BNBWD || (p.src.y && p.src.stats && p.src.rstats), "conv: BNBWD source incomplete");
    STL_CHECK(!(p.out_stats && p.red), "conv: out_stats and red are exclusive");
    STL_CHECK(!p.red || p.mask_y, "conv: red needs mask_y");
    STL_CHECK(!p.mask_y || (p.mask_bn.gamma && p.mask_bn.beta && (p.mask_bn.stats || (p.mask_bn.rmean && p.mask_bn.rvar))), "conv: mask BN incomplete");

    const bool fused = p.wg_partial != nullptr;
    if (fused) {
        STL_CHECK(p.ks == 3 && p.stride == 1 && !p.stuff, "conv(fused): 3x3 stride 1 only");
        STL_CHECK(p.Ci == p.Co && p.Ci % 32 == 0 && p.Ci <= 64, "conv(fused): Ci == Co == C with C %% 32 == 0 and C <= 64 (got %d, %d)", p.Ci, p.Co);
        STL_CHECK(p.src.mode == STL_SRC_BNBWD, "conv(fused): the gradient source must be BNBWD");
        STL_CHECK(p.wg_h.x && (p.wg_h.mode == STL_SRC_PLAIN || p.wg_h.mode == STL_SRC_BN), "conv(fused): wg_h must be a PLAIN or BN source");
        STL_CHECK(p.wg_h.mode != STL_SRC_BN || (p.wg_h.gamma && p.wg_h.beta && (p.wg_h.stats || (p.wg_h.rmean && p.wg_h.rvar))), "conv(fused): wg_h BN source incomplete");
        STL_CHECK(p.wg_nsplit >= 8 && p.wg_nsplit % 8 == 0, "conv(fused): wg_nsplit must be a positive multiple of 8 (got %d)", p.wg_nsplit);
    }
    if (use_1x1(p))  // wide 1x1 convolutions: streaming GEMM kernel (conv1x1.inc)
        return p.dtype == STL_BF16 ? run_1x1<__bf16>(p, (hipStream_t)stream) : run_1x1<float>(p, (hipStream_t)stream);

    // block shape and pixel tile: planned once by stl_conv_plan (shape >= 0), else searched here
    Plan plan;
    if (p.shape >= 0 && p.shape < NSHAPES && p.TH > 0 && p.TW > 0) {
        STL_CHECK(p.TH * p.TW <= SHAPES[p.shape].px, "conv: tile %dx%d exceeds block shape %d", p.TH, p.TW, p.shape);
        plan = Plan{p.shape, p.TH, p.TW, 0, 0.0};
    } else if (p.TH > 0 && p.TW > 0) {  // explicit 128-pixel tile (tests)
        STL_CHECK(p.TH * p.TW <= 128, "conv: explicit tile %dx%d exceeds 128 pixels", p.TH, p.TW);
        plan = Plan{0, p.TH, p.TW, 0, 0.0};
    } else {
        plan = choose_plan(p, ck);
        STL_CHECK(plan.shape >= 0, "conv: no tile fits LDS for %dx%d ks %d stride %d Ci %d", p.Ho, p.Wo, p.ks, p.stride, p.Ci);
    }
    {
        const Shape shp = SHAPES[plan.shape];
        const int nv = ceil_div(((plan.TH - 1) * p.stride + p.ks) * ((plan.TW - 1) * p.stride + p.ks) * 4, shp.lthr);
        STL_CHECK(nv <= shp.nva_max, "conv: tile %dx%d has too large a halo for block shape %d", plan.TH, plan.TW, plan.shape);
    }
    ConvK k;
    k.p = p;
    k.nks = 0;
    k.dbg = getenv("STL_CONV_STAMPS") ? 1 : 0;
    k.seff = p.stride;
    k.pad = pad;
    k.PI = p.stuff ? (p.Ho + 1) : p.stride * (p.Ho + 1);
    k.TH = plan.TH, k.TW = plan.TW;
    const size_t lds = lds_bytes(p, plan.shape, plan.TH, plan.TW, ck, &k);
    STL_CHECK(lds <= 160 * 1024, "conv: tile needs %zu B of LDS (>160 KiB)", lds);
    const int vrows = p.B * (p.Ho + 1);
    k.tiles_c = ceil_div(p.Wo, plan.TW);
    k.npt = ceil_div(vrows, plan.TH) * k.tiles_c;
    k.r_HC = 1.0f / k.HC, k.r_TW = 1.0f / k.TW, k.r_tc = 1.0f / k.tiles_c, k.r_PI = 1.0f / k.PI, k.r_vp = 1.0f / (p.Ho + 1);
    STL_CHECK(k.npt < (1 << 21) && (int64_t)vrows * 2 < (1 << 21) && k.HP < 4096, "conv: index range exceeds the fast-division limits");
    STL_CHECK(k.tiles_c < (1 << 11) && k.PI < (1 << 11) && p.Ho + 1 < (1 << 11), "conv: rows / columns exceed the scalar-division limits (2047)");
    {
        const unsigned long long two32 = 1ull << 32;
        k.m_tc = (two32 + k.tiles_c - 1) / k.tiles_c, k.m_PI = (two32 + k.PI - 1) / k.PI, k.m_vp = (two32 + p.Ho) / (p.Ho + 1);
    }
    STL_CHECK((int64_t)p.B * p.Hi * p.Wi < (1 << 24) && (int64_t)p.B * p.Ho * p.Wo < (1 << 24) && (int64_t)p.B * p.Hi * p.Wi * p.Ci < ((int64_t)1 << 31) &&
                  (int64_t)p.B * p.Ho * p.Wo * p.Co < ((int64_t)1 << 31),
              "conv: tensors beyond 2^24 pixels or 2^31 elements are not addressable (24-bit multiplies, 32-bit element offsets)");
    const Shape sh = SHAPES[plan.shape];
    int gx = ceil_div(k.npt, 8) * 8;
    // resident waves per SIMD ~2: persistent blocks that loop over their tiles with the next tile's
    // loads in flight beat a second round of fresh blocks (measured: tools/conv_probe7.py)
    k.ny = ceil_div(p.Co, sh.co);
    // The budget is for the whole grid (pixel blocks x channel blocks): one resident round of blocks
    // (256 for the 8-wave shapes, 1024 for the 4-wave ones); with ny channel blocks per pixel tile the
    // pixel dimension gets budget / ny (3x3 32->256 at 96x72: 102 -> 90 us; tools/conv_probe8.py)
    static const int cap4 = getenv("STL_CONV_GRID_CAP4") ? atoi(getenv("STL_CONV_GRID_CAP4")) : 768;   // 1024 -> 768: 16.62 -> 16.56 ms/step (three resident blocks per CU leave room for the co-running kernels)
    static const int cap8 = getenv("STL_CONV_GRID_CAP8") ? atoi(getenv("STL_CONV_GRID_CAP8")) : 256;
    int cap = sh.ws ? 256 : (sh.thr == 512 ? cap8 : cap4);
    if (plan.shape == 8) cap = getenv("STL_CONV_GRID_CAP_S8") ? atoi(getenv("STL_CONV_GRID_CAP_S8")) : cap4 / 2;   // two 8-wave blocks per CU
    if (!getenv("STL_CONV_CAP_PER_TILE")) cap = std::max(8, (cap / k.ny + 7) / 8 * 8);
    if (const char* e = getenv("STL_CONV_GRID_CAP")) cap = atoi(e) > 0 ? (atoi(e) + 7) / 8 * 8 : cap;
    if (p.grid_pct > 0 && p.grid_pct != 100) cap = std::max(8, (cap * p.grid_pct / 100 + 7) / 8 * 8);
    if (gx > cap) gx = cap;
    dim3 grid(gx * k.ny, 1);
    const int nva = ceil_div(k.HP * 4, sh.lthr);
    if (fused) {
        STL_CHECK(plan.shape == 4 && nva <= 3, "conv(fused): needs the 128 px x 32 channel block shape with a halo of at most 192 pixels (shape %d, %d px)", plan.shape, k.HP);
        k.nks = ceil_div(plan.TH * plan.TW, 4 * kv);
        grid = dim3(p.wg_nsplit * k.ny, 1);   // every pixel group writes one slab, tiles or not
    }
    if (getenv("STL_CONV_DEBUG"))
        fprintf(stderr, "[stl conv] %dx%d Ci%d Co%d ks%d s%d: shape=%d tile=%dx%d npt=%d grid=(%d x %d) lds=%zu nva=%d nchunks=%d\n", p.Ho,
                p.Wo, p.Ci, p.Co, p.ks, p.stride, plan.shape, plan.TH, plan.TW, k.npt, gx, k.ny, lds, nva, k.nchunks);
    hipStream_t st = (hipStream_t)stream;
    const bool q = p.src.mode == STL_SRC_BNBWD;
    if (fused)
        return p.dtype == STL_BF16 ? dispatch_fused<__bf16>(k.nchunks, k, grid, lds, st) : dispatch_fused<float>(k.nchunks, k, grid, lds, st);
    if (zm)
        return p.dtype == STL_BF16 ? dispatch_bnadd<__bf16>(plan.shape, nva, k, grid, lds, st) : dispatch_bnadd<float>(plan.shape, nva, k, grid, lds, st);
    const bool plain = !p.bias && !p.addend && !p.mask_y && !p.mask_z && !p.red;   // forward convs of the pose network
#define DISPATCH(T)                                                                                      \
    if (p.ks == 3) {                                                                                     \
        if (q) return dispatch<T, 3, true, false>(plan.shape, nva, k, grid, lds, st);                    \
        return plain ? dispatch<T, 3, false, true>(plan.shape, nva, k, grid, lds, st) : dispatch<T, 3, false, false>(plan.shape, nva, k, grid, lds, st); \
    }                                                                                                    \
    if (q) return dispatch<T, 1, true, false>(plan.shape, nva, k, grid, lds, st);                        \
    return plain ? dispatch<T, 1, false, true>(plan.shape, nva, k, grid, lds, st) : dispatch<T, 1, false, false>(plan.shape, nva, k, grid, lds, st);
    if (p.dtype == STL_BF16) {
        DISPATCH(__bf16)
    } else {
        DISPATCH(float)
    }
#undef DISPATCH
}
