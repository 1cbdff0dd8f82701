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

// The file is compiled three times by stlpose_amd/build.py -- -DSTL_DT=1: the bf16 kernels + the C ABI entry points, -DSTL_DT=0: the
// fp32 kernels, -DSTL_DT=3: the f16 forward kernels -- so that its kernel instantiations build in parallel; STL_DT=2 (default) =
// one unit with everything.  NOTE: STL_DT numbers TRANSLATION UNITS, not element types (the element type STL_F16 is 2, the f16
// unit is 3): only the STL_HAS_* macros below may look at it.
#ifndef STL_DT
#define STL_DT 2
#endif
#define STL_HAS_F32 (STL_DT == 0 || STL_DT == 2)
#define STL_HAS_BF16 (STL_DT == 1 || STL_DT == 2)   // + the C ABI entry points
#define STL_HAS_F16 (STL_DT == 3 || STL_DT == 2)    // forward kernels of the mixed 16-bit mode (STL_F16)

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
};

// dtype back ends (defined where their kernels are instantiated): path 0 = streaming 1x1 kernel, 1 = block-end (BNADD) source,
// 2 = everything else
int stl_conv_backend_bf16(int path, const stl_conv& p, const ConvK& k, int shape, int nva, dim3 grid, size_t lds, hipStream_t st);
int stl_conv_backend_f32(int path, const stl_conv& p, const ConvK& k, int shape, int nva, dim3 grid, size_t lds, hipStream_t st);
int stl_conv_backend_f16(int path, const stl_conv& p, const ConvK& k, int shape, int nva, dim3 grid, size_t lds, hipStream_t st);

namespace {

constexpr int PSA = 96;

// Compiled in only with -DSTL_STAMPS (python -m stlpose_amd.build --stamps -> libstlpose_hip_stamps.so): even when
// disabled at run time, the stores make the compiler place s_waitcnt vmcnt() in front of whatever reuses their data
// registers -- in the stage loop that drains loads which were meant to stay in flight.
#ifdef STL_STAMPS
#include "../../include/stlpose_hip_debug.h"
// debug-only phase stamps (block 0, thread 0; enabled by STL_CONV_STAMPS=1): never read by the kernel
__device__ long long g_stamps[32];
__device__ long long g_stamps2[64];  // wave-specialised kernel: [0..23] loader, [32..55] compute (6 stages x 4)
#define STAMP(i)                                                          \
    do {                                                                  \
        if (k.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps[i] = wall_clock64(); \
    } while (0)
#define STAMP2(i)                                                         \
    do {                                                                  \
        if (k.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (i) < 64) g_stamps2[i] = wall_clock64(); \
    } while (0)
#else
#define STAMP(i) do {} while (0)
#define STAMP2(i) do {} while (0)
#endif


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

#ifndef STL_PIN_TAPS
#define STL_PIN_TAPS 1
#endif
#include "conv_common.inc"

// WM x WN waves, MT pixel tiles and NTW channel tiles per wave; NVA staging vectors per thread for
// the input halo; Q: source is BNBWD (second tensor on load).
// WR: 1 = the filters are known to be LDS-resident (whole K in one chunk): the per-stage filter
// staging registers and descriptors do not exist in that instantiation (no spills at 128 VGPRs);
// -1 = decided at run time (k.wres)
// ZM: the source is STL_SRC_BNADD (Q = true: second tensor on load): the staged value is the residual block end
// z = ReLU(BN(x) + y); it is also written to p.src_out by the block that owns the pixel (tile interior, channel block 0).
// PE: plain epilogue -- the launch has no bias / addend / mask operand (conv_common.inc, epilogue_apply).
// TY: element type of the FORWARD tensors a data gradient reads (src.y, mask_y, mask_z); T everywhere else.
// DB: two LDS images [halo | filters] (k.sz_a / k.sz_b apart) and ONE barrier per stage: while stage s is multiplied out of one
// image, the staged registers of stage s + 1 are transformed and written into the other and the loads of stage s + 2 are
// requested, one staging vector per tap of the MFMA loop (round 4; block shape 3).
// EO: compile-time set of epilogue operands of a data gradient, or -1 (conv_common.inc, EpiOps).
template <typename T, int KS, int WM, int WN, int MT, int NTW, int NVA, bool Q, bool PE, int OCC, int WR, bool ZM = false, typename TY = T, bool DB = false, int EO = -1>
__global__ __launch_bounds__(64 * WM * WN, OCC) void conv_core_kernel(const ConvK k) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTHR = 64 * WM * WN;
    constexpr int KV = ET<T>::KV, CK = ET<T>::CK, TAPS = KS * KS;
    static_assert(!ZM || (Q && KS == 3), "block-end source: 3x3, two tensors on load");
    static_assert(!DB || (!ZM && WR == 0 && KS == 3), "double-buffered image: 3x3, plain / BN / BNBWD sources, filters staged per chunk");
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
    for (int c = tid; c < k.cipad; c += NTHR) {
        float a = 0.f, b = 0.f, cc = 0.f;
        if (c < p.Ci) src_consts(p.src, c, p.Ci, a, b, cc);
        cs[c] = a, cs[k.cipad + c] = b, cs[2 * k.cipad + c] = cc;
    }
    if (!PE && EpiOps<EO>::my(p)) {
        // the LAST wave computes the mask constants while the first one(s) do the source's: the two sets are
        // dependent global round trips each, and run in parallel on different waves instead of back to back
        for (int c = tid - (NTHR - 64); c >= 0 && c < BCO; c += 64) {
            float a = 0.f, b = 0.f, mu = 0.f, rs = 0.f;
            if (n0 + c < p.Co) {
                bn_mean_rstd(p.mask_bn, n0 + c, p.Co, mu, rs);
                bn_affine(p.mask_bn.gamma[n0 + c], p.mask_bn.beta[n0 + c], mu, rs, a, b);
            }
            cm[c] = a, cm[BCO + c] = b, cm[2 * BCO + c] = mu, cm[3 * BCO + c] = rs;
        }
    }
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
    // One register set of staged loads: the loads of stage s + 1 are issued while stage s is multiplied.  (A ring of two
    // sets measured neutral-to-slower in round 3 -- serial 21.39 vs 21.26 ms, step 15.23 vs 15.18 -- and was removed: a
    // stage is 1.3 us of transform + LDS write, 0.5 us of address set-up and 1.5 us of MFMA phase back to back; the loads
    // are not what it waits for.)
    V16 ra[NVA], rq[Q ? NVA : 1], rb[NVB];
    int a_go[NVA];
    int st_t, st_ch;      // the stage (tile, chunk) the set holds
    bool st_have;

    // tile-uniform terms of tile t (scalar unit), then the element offset of staging slot i (-1: padding / outside)
    auto tile_terms = [&](int t, int& y0, int& cb, int& b0) __attribute__((always_inline)) {
        const int tr = sdiv(t, k.m_tc), tc = t - tr * k.tiles_c;
        const int vrs = tr * k.TH * k.seff;
        cb = tc * k.TW * k.seff - k.pad;
        b0 = sdiv(vrs, k.m_PI), y0 = vrs - b0 * k.PI - k.pad;
    };
    auto slot_go = [&](int rc, int y0, int cb, int b0) __attribute__((always_inline)) -> int {   // rc = a_rc[] of the slot
        const int n_PI = -k.PI;
        const int iy0 = y0 + (rc >> 16);
        int ix = cb + (rc & 0xffff);
        const int wr_ = fdiv(iy0 < 0 ? 0 : iy0, k.r_PI);
        int iy = mad24_vsv(wr_, n_PI, iy0);
        bool ok = (rc >= 0) & (iy0 >= 0) & (ix >= 0) & (b0 + wr_ < p.B);
        if (p.stuff) {   // zero-stuffed source (transposed convolution): only even rows / columns exist
            ok = ok & !((iy | ix) & 1);
            iy >>= 1, ix >>= 1;
        }
        ok = ok & (iy < p.Hi) & (ix < p.Wi);
        const int e = madu24_vsv(madu24_vsv(madu24_vsv(b0 + wr_, p.Hi, iy), p.Wi, ix), p.Ci, a_part * KV);
        return ok ? e : -1;
    };
    auto tile_setup = [&](int t, int* go) {
        int y0, cb, b0;
        tile_terms(t, y0, cb, b0);
#pragma unroll
        for (int i = 0; i < NVA; ++i) go[i] = slot_go(a_rc[i], y0, cb, b0);
    };
    // Loads are UNCONDITIONAL (invalid slots read element 0 and are zeroed at write time): a
    // guarded load makes hipcc branch around it and wait vmcnt(0) per element, which serialises
    // the whole staging burst into dependent round trips.
    auto issue = [&](const int* go, int k0, bool en) __attribute__((always_inline)) {
        const bool chok = en && (k0 + a_part * KV) < p.Ci;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int off = (go[i] >= 0 && chok) ? go[i] + k0 : 0;
            ra[i] = ldg16((const char*)p.src.x + (size_t)(uint32_t)off * sizeof(T));
            if (Q) rq[i] = ldg16((const char*)p.src.y + (size_t)(uint32_t)off * sizeof(T));
        }
        if (!wres) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                const bool ok = en && b_g[i] >= 0 && (k0 + ((tid + i * NTHR) & 3) * KV) < p.Ci;
                rb[i] = ldg16((const char*)p.w + (size_t)(ok ? b_g[i] + k0 : 0) * sizeof(T));
            }
        }
    };
    const float relu_lo = p.src.relu ? 0.f : -INFINITY;
    auto write_lds = [&](const int* go, int k0) __attribute__((always_inline)) {
        const int ch = k0 + a_part * KV;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            if (a_rc[i] < 0) continue;
            const bool ok = go[i] >= 0 && ch < p.Ci;
            const int chc = ok ? ch : 0;
            V16 val = ra[i];
            if constexpr (ZM) {
                val = xform_bnadd<T>(val, rq[i], cs + chc, cs + k.cipad + chc, relu_lo);
                if (ok && ((a_int >> i) & 1u)) stg16((char*)p.src_out + (size_t)(go[i] + k0) * sizeof(T), val);   // z, once per pixel
            } else if (Q)
                val = xform_bnbwd<T, TY>(val, rq[i], cs + chc, cs + k.cipad + chc, cs + 2 * k.cipad + chc);
            else if (p.src.mode != STL_SRC_PLAIN)
                val = xform_bn<T>(val, cs + chc, cs + k.cipad + chc, relu_lo);
            mask16(val, ok);  // zero padding applies AFTER the transform
            const int v = tid + i * NTHR;
            *reinterpret_cast<V16*>(sA + (v >> 2) * PSA + (v & 3) * 16) = val;
        }
        if (!wres) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                const bool ok = b_g[i] >= 0 && (k0 + ((tid + i * NTHR) & 3) * KV) < p.Ci;
                V16 val = rb[i];
                mask16(val, ok);
                if (tid + i * NTHR < BCO * TAPS * 4) *reinterpret_cast<V16*>(sB + b_l[i]) = val;
            }
        }
    };

    if (wres) {  // whole K fits one chunk: filters stay resident in LDS for all tiles (rb[] is free then)
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const bool ok = b_g[i] >= 0 && (((tid + i * NTHR) & 3) * KV) < p.Ci;
            rb[i] = ldg16((const char*)p.w + (size_t)(ok ? b_g[i] : 0) * sizeof(T));
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
    // request the cursor's stage, remember which stage the registers now hold, advance the cursor
    auto prefetch = [&]() __attribute__((always_inline)) {
        if (have_n && ch_n == 0) tile_setup(t_n, a_go);
        st_t = t_n, st_ch = ch_n, st_have = have_n;
        issue(a_go, ch_n * CK, have_n);
        if (++ch_n == k.nchunks) {
            ch_n = 0, it_n += nx, t_n = xcd * T8 + it_n;
            have_n = have_n && (it_n < T8) && (t_n < k.npt);
        }
    };
    STAMP(3);
    if constexpr (!DB) {
        prefetch();
        if (wres) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                const bool ok = b_g[i] >= 0 && (((tid + i * NTHR) & 3) * KV) < p.Ci;
                V16 val = rb[i];
                mask16(val, ok);
                if (tid + i * NTHR < BCO * TAPS * 4) *reinterpret_cast<V16*>(sB + b_l[i]) = val;
            }
        }
        __syncthreads();  // constants + resident filters visible
    }
    STAMP(4);

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // epilogue operands (addend / masks) of all MT tiles of the wave, fetched in one burst: before the MFMAs of the
    // tile's last chunk where the block owns its CU anyway (EPRE: 12 registers per tile), else at the start of the epilogue
    constexpr bool EPRE = (OCC <= 1 && WM == 4 && WN == 2 && NTW == 2) && Q && !PE && !ZM && sizeof(T) == 2;   // fp32 would need 96 registers and spills
    // register-tight blocks with a compile-time operand set: the operands of ALL the wave's tiles in one burst at the start of
    // the epilogue (the fragment registers are dead by then) -- one memory round trip per block tile instead of one per 16 pixels
    // (13.92 vs 13.96 ms per step)
    constexpr bool EBURST = !EPRE && (EO == 1 || EO == 2) && !PE && sizeof(T) == 2;   // (all three operands: 25 - 30 spilled registers at the 128 budget)
    // element index (times Co) of this lane's output pixel in pixel tile mi of the tile at (vr0, c0); pok = inside the image
    auto out_pixel = [&](int vr0, int c0, int mi, bool& pok) __attribute__((always_inline)) -> size_t {
        const int eb0 = sdiv(vr0, k.m_vp), ey0 = vr0 - eb0 * vpitch;
        int eyx = e_yx[mi];
        if constexpr (DB) {   // not kept in registers across the stage loop
            const int m = (wm * MT + mi) * 16 + r16, ty_ = fdiv(m, k.r_TW);
            eyx = m < tilepx ? ((ty_ << 16) | (m - ty_ * k.TW)) : -1;
        }
        const int ty = eyx >> 16, tx = eyx & 0xffff;
        const int oy0 = ey0 + ty, c = c0 + tx;
        const int wr_ = fdiv(oy0, k.r_vp);
        const int oy = mad24_vsv(wr_, -vpitch, oy0), b = eb0 + wr_;
        pok = (eyx >= 0) & (b < p.B) & (oy < p.Ho) & (c < p.Wo);
        const int e = mulu24_vs(madu24_vsv(madu24_vsv(b, p.Ho, oy), p.Wo, c), p.Co);
        return pok ? (size_t)(uint32_t)e : 0;
    };
    auto epi_fetch = [&](int vr0, int c0, bool* pokv, size_t* pixv, EpiRaw<NTW>* er) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            pixv[mi] = out_pixel(vr0, c0, mi, pokv[mi]);
            epilogue_fetch<T, NTW, true, EO>(p, pokv[mi], pixv[mi], n0, wn * NTW * 16, g, er[mi]);
        }
    };

    if constexpr (DB) {
        // ---- double-buffered image.  Registers hold stage s + 1 while stage s is multiplied; slot j (halo vector j, then
        // filter vector j - NVA) is transformed + written into the OTHER image and immediately re-requested for stage s + 2
        // behind the MFMAs of tap j, so every load has a whole stage to land and the barrier at the end of the stage is
        // the only one: it publishes image (s + 1) and retires image (s).
        constexpr int NSLOT = NVA + NVB;
        prefetch();                       // stage 0 -> registers
        int cu_t = st_t, cu_ch = st_ch;
        bool cu_have = st_have;
        __syncthreads();                  // constants visible
        write_lds(a_go, st_ch * CK);      // -> image 0
        prefetch();                       // stage 1 -> registers
        __syncthreads();
        int img = 0;
        [[maybe_unused]] int dbg_it = 0;   // stamps: [4 i .. 4 i + 3] = top of stage i, taps done, barrier passed, epilogue done
        while (cu_have) {
            STAMP2(4 * dbg_it);
            const int t = cu_t, ch0 = cu_ch;
            const int tr = sdiv(t, k.m_tc), tc = t - tr * k.tiles_c;
            const int vr0 = tr * k.TH, c0 = tc * k.TW;
            const bool last_chunk = (ch0 + 1 == k.nchunks);
            bool pokv[EPRE ? MT : 1];
            size_t pixv[EPRE ? MT : 1];
            EpiRaw<NTW> er[EPRE ? MT : 1];
            // the stage in the registers (to be written) and the stage to request
            const int k0w = st_ch * CK, k0n = ch_n * CK;
            const bool en_n = have_n, nt = have_n && ch_n == 0;
            int y0n = 0, cbn = 0, b0n = 0;
            if (nt) tile_terms(t_n, y0n, cbn, b0n);
            // channel constants of the stage in the registers (this thread's KV channels of the chunk: the same for all its
            // halo slots); live over the halo slots only
            // (not beside the 48 operand registers of the three-operand data gradient: 11 - 19 spilled registers)
            constexpr bool KREG = !Q || EO == 0 || EO == 1 || EO == 2;
            constexpr int NPK = sizeof(T) == 2 ? 4 : 2;
            f2v kca[NPK], kcb[NPK], kcc[Q ? NPK : 1];
            if constexpr (KREG) {
                const int ch = k0w + a_part * KV;   // < cipad: the tables cover it
#pragma unroll
                for (int i = 0; i < NPK; ++i) {
                    kca[i] = reinterpret_cast<const f2v*>(cs + ch)[i];
                    kcb[i] = reinterpret_cast<const f2v*>(cs + k.cipad + ch)[i];
                    if (Q) kcc[i] = reinterpret_cast<const f2v*>(cs + 2 * k.cipad + ch)[i];
                }
            }
            const char* sAr = sA + img * k.sz_a;
            const char* sBr = sB + img * k.sz_b;
            char* sAw = sA + (img ^ 1) * k.sz_a;
            char* sBw = sB + (img ^ 1) * k.sz_b;
            auto slot = [&](int j) __attribute__((always_inline)) {
                if (j < NVA) {
                    const int i = j;
                    const int ch = k0w + a_part * KV;
                    const bool ok = a_go[i] >= 0 && ch < p.Ci;
                    V16 val = ra[i];
                    if constexpr (KREG) {
                        if (Q)
                            val = xform_bnbwd_r<T, TY>(val, rq[i], kca, kcb, kcc);
                        else if (p.src.mode != STL_SRC_PLAIN)
                            val = xform_bn_r<T>(val, kca, kcb, relu_lo);
                    } else {
                        val = xform_bnbwd<T, TY>(val, rq[i], cs + ch, cs + k.cipad + ch, cs + 2 * k.cipad + ch);
                    }
                    mask16(val, ok);
                    int v = tid + i * NTHR;
                    asm volatile("" : "+v"(v));   // keeps the slot's address terms from being hoisted out of the stage loop (and spilled)
                    if (v < k.HP * 4) *reinterpret_cast<V16*>(sAw + (v >> 2) * PSA + (v & 3) * 16) = val;
                    if (nt) {   // a new tile (rare: these blocks mostly own one): the slot's halo position again, not kept in registers
                        const int hp = v >> 2, hr = fdiv(hp, k.r_HC);
                        a_go[i] = slot_go(v < k.HP * 4 ? ((hr << 16) | (hp - hr * k.HC)) : -1, y0n, cbn, b0n);
                    }
                    const bool chok = en_n && (k0n + a_part * KV) < p.Ci;
                    const int off = (a_go[i] >= 0 && chok) ? a_go[i] + k0n : 0;
                    ra[i] = ldg16((const char*)p.src.x + (size_t)(uint32_t)off * sizeof(T));
                    if (Q) rq[i] = ldg16((const char*)p.src.y + (size_t)(uint32_t)off * sizeof(T));
                } else {
                    // filter vector v = (row n, tap, 16-byte part): LDS and global offsets from v itself (the b_g / b_l
                    // registers of the single-image kernel do not fit beside the second tensor's staging registers)
                    const int i = j - NVA;
                    int v = tid + i * NTHR;
                    asm volatile("" : "+v"(v));   // (as above)
                    const int part = (v & 3) * KV;
                    const int n = mulu24_vs(v, 1821) >> 16;   // v / 36 for v < 3311
                    const bool vok = (i + 1) * NTHR <= BCO * TAPS * 4 || v < BCO * TAPS * 4;
                    const bool rok = vok && n0 + n < p.Co;
                    const int bg = madu24_vsv((v >> 2) + n0 * TAPS, p.Ci, part);
                    V16 val = rb[i];
                    mask16(val, rok && (k0w + part) < p.Ci);
                    if (vok) *reinterpret_cast<V16*>(sBw + v * 16 + n * 32) = val;
                    const bool okn = en_n && rok && (k0n + part) < p.Ci;
                    rb[i] = ldg16((const char*)p.w + (size_t)(uint32_t)(okn ? bg + k0n : 0) * sizeof(T));
                }
            };
            {
                V16 wf[2][NTW], xf[2][MT];
#pragma unroll
                for (int ni = 0; ni < NTW; ++ni) wf[0][ni] = *reinterpret_cast<const V16*>(sBr + woff + ni * 16 * ROWB);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) xf[0][mi] = *reinterpret_cast<const V16*>(sAr + xoff[mi]);
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    if (tap + 1 < TAPS) {
                        const int toff = (((tap + 1) / KS) * k.HC + ((tap + 1) % KS)) * PSA;
#pragma unroll
                        for (int ni = 0; ni < NTW; ++ni)
                            wf[(tap + 1) & 1][ni] = *reinterpret_cast<const V16*>(sBr + woff + ni * 16 * ROWB + (tap + 1) * 64);
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi) xf[(tap + 1) & 1][mi] = *reinterpret_cast<const V16*>(sAr + xoff[mi] + toff);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NTW; ++ni) mma16<T>(acc[mi][ni], wf[tap & 1][ni], xf[tap & 1][mi]);
#pragma unroll
                    for (int j = tap; j < NSLOT; j += TAPS) slot(j);
                    if constexpr (EPRE) {
                        // epilogue operands of the tile: requested behind the last halo slot (whose transform temporaries
                        // and these 48 registers do not fit side by side), six taps ahead of their use
                        if (tap == (NVA < TAPS ? NVA : TAPS - 1) && last_chunk) epi_fetch(vr0, c0, pokv, pixv, er);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // stage s + 1 becomes the current one, the requested stage is the one in the registers; advance the cursor
            cu_t = st_t, cu_ch = st_ch, cu_have = st_have;
            st_t = t_n, st_ch = ch_n, st_have = have_n;
            if (++ch_n == k.nchunks) {
                ch_n = 0, it_n += nx, t_n = xcd * T8 + it_n;
                have_n = have_n && (it_n < T8) && (t_n < k.npt);
            }
            STAMP2(4 * dbg_it + 1);
            __syncthreads();  // image (s + 1) complete, image (s) free
            STAMP2(4 * dbg_it + 2);
            img ^= 1;
            if (last_chunk) {
                if constexpr (EPRE) {
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
                        epilogue_apply<T, NTW, BCO, false, TY, EO>(p, acc[mi], cm, pokv[mi], pixv[mi], n0, wn * NTW * 16, g, s0, s1, er[mi]);
                } else {
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) {
                        bool pok;
                        const size_t pix = out_pixel(vr0, c0, mi, pok);
                        epilogue_tile<T, NTW, BCO, PE, TY, EO>(p, acc[mi], cm, pok, pix, n0, wn * NTW * 16, g, s0, s1);
                    }
                }
            }
            STAMP2(4 * dbg_it + 3);
#ifdef STL_STAMPS
            ++dbg_it;
#endif
        }
    } else
    // flat loop over stages (tile, chunk); exactly ONE issue() site inside the loop so that the staging registers need
    // no PHI copies (which would force a vmcnt(0) before the MFMAs)
    while (st_have) {
        const int t = st_t, ch0 = st_ch;
        const int tr = sdiv(t, k.m_tc), tc = t - tr * k.tiles_c;
        const int vr0 = tr * k.TH, c0 = tc * k.TW;
        write_lds(a_go, ch0 * CK);
        __syncthreads();
        if (ch0 == 0) STAMP(5);
        const bool last_chunk = (ch0 + 1 == k.nchunks);
        bool pokv[EPRE ? MT : 1];
        size_t pixv[EPRE ? MT : 1];
        EpiRaw<NTW> er[EPRE ? MT : 1];
        if constexpr (EPRE) {
            if (last_chunk) epi_fetch(vr0, c0, pokv, pixv, er);   // ahead of the next loads: consumed behind a counted wait
        }
        prefetch();  // stage s + 1 into the registers just drained: its loads land during the MFMAs of this stage
        if (ch0 == 0) STAMP(6);
        {  // fragment reads of tap t+1 are issued before the MFMAs of tap t (static double buffer).  PIN: the order is
           // pinned with scheduling barriers -- left alone, the scheduler sinks every read to just in front of its
           // first use to save registers, and each MFMA pair then waits a full LDS round trip
            constexpr bool PIN = (STL_PIN_TAPS != 0);
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
        __syncthreads();  // everyone is done with sA/sB of this stage
        if (ch0 == 0) STAMP(7);
        if (last_chunk) {
            STAMP(8);
            // ---- epilogue straight from the accumulators: lane = pixel r16, 4 channels per tile
            if constexpr (EPRE) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    epilogue_apply<T, NTW, BCO, false, TY, EO>(p, acc[mi], cm, pokv[mi], pixv[mi], n0, wn * NTW * 16, g, s0, s1, er[mi]);
            } else if constexpr (EBURST) {
                bool pokb[MT];
                size_t pixb[MT];
                EpiRaw<NTW> erb[MT];
                epi_fetch(vr0, c0, pokb, pixb, erb);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    epilogue_apply<T, NTW, BCO, false, TY, EO>(p, acc[mi], cm, pokb[mi], pixb[mi], n0, wn * NTW * 16, g, s0, s1, erb[mi]);
            } else {   // register-tight instantiations: tile by tile
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    bool pok;
                    const size_t pix = out_pixel(vr0, c0, mi, pok);
                    epilogue_tile<T, NTW, BCO, PE, TY, EO>(p, acc[mi], cm, pok, pix, n0, wn * NTW * 16, g, s0, s1);
                }
            }
        }
        if (last_chunk) STAMP(9);
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
    STAMP(11);
#ifdef STL_STAMPS
    if (k.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps[13] = __builtin_amdgcn_s_memtime();
#endif
}

#include "conv_ws.inc"
#include "conv1x1.inc"
#include "conv_r2.inc"

template <typename T, typename TY, int KS, int WM, int WN, int MT, int NTW, int NVA, bool Q, bool PE, int OCC = 1, int WR = -1, bool ZM = false, bool DB = false, int EO = -1>
int launch(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_core_kernel<T, KS, WM, WN, MT, NTW, NVA, Q, PE, OCC, WR, ZM, TY, DB, EO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    STL_LAUNCH((conv_core_kernel<T, KS, WM, WN, MT, NTW, NVA, Q, PE, OCC, WR, ZM, TY, DB, EO>), grid, dim3(64 * WM * WN), lds, st, k);
    static char nbuf[160];
    static const char* nm = stl_kname<T>(nbuf, "conv_core_kernel", {KS, WM, WN, MT, NTW, NVA, Q, PE, OCC, WR, ZM, stl_code<TY>(), DB, EO});
    stl_note_kernel(nm, true);
    STL_LAUNCH_CHECK("conv_core");
    return 0;
}

// Block shapes (px x co, waves) and why each exists (DESIGN.md 5 lists the instantiations):
//   0 = 128 x 64, 4 waves  -- small maps where no 256-pixel tile fits, and explicit 128-pixel tiles (tests)
//   1 = 512 x 32, 8 waves  -- Co <= 32 with >= 128 input channels on a large map (transition1 256 -> 32: each filter chunk staged
//                             once per 512 pixels instead of once per 128: 105.8 -> 71.8 us)
//   2 = 256 x 64, 8 waves  -- C >= 64 layers whose K is one chunk, 1x1 convolutions on that block, the block-end (BNADD) source
//   3 = 256 x 64, 8 waves, two LDS images, one barrier per stage -- the C >= 64 3x3 layers (forward and data gradient; round 4)
//   5 = 256 x 64, 8 waves, <= 128 VGPRs and <= 76 KB of LDS: TWO PER CU (conv_r2.inc, round 5: filters by LDS-DMA, planar halo image) --
//       the C >= 64 3x3 stride-1 forward convolutions of the 16-bit modes; launches it has no instantiation for fall back to shape 3
//  10 = 256 x 32, 8 waves along the pixels, the same kernel for the C <= 32 layers (STL_CONV_R2 bit 4; fallback: shape 8)
//   6 = 128 x 64, 8 waves, the same kernel with two pixel tiles per wave -- their data gradients (a second source tensor and up
//       to three epilogue operands beside 32 accumulators do not fit 128 registers; with 16 they do); fallback: shape 0
//   4 = 128 x 32, 4 waves  -- C <= 32 fallback where no 256-pixel tile fits (<= 128 VGPRs, <= 40 KB LDS: four blocks per CU)
//   7 = 128 x 64, wave-specialised (4 loader + 4 compute waves, double-buffered LDS image) -- stride-2 convolutions
//   8 = 256 x 32, 8 waves  -- the C <= 32 3x3 layers (two 128-pixel halves share one filter copy and one halo tile)
//   9 = 128 x 32, wave-specialised -- the deep small maps (Co >= 256 on <= 16384 pixels: 12x9 at C = 256 gets 256 blocks, not 128)
// (256 x 128 and the wave-specialised 512 x 32 / 256 x 64 blocks -- ids 5 / 6 -- lost end to end in rounds 2 and 3 and were removed.)
struct Shape {
    int px, co, thr;   // pixels / output channels per block, threads
    int ws;            // 1: wave-specialised kernel (4 loader + 4 compute waves, double-buffered LDS); 2: uniform kernel, two LDS images; 3: conv_r2
    int lthr, nva_max; // threads that stage the halo, max staging vectors per such thread
};
constexpr int NSHAPES = 11;
constexpr Shape SHAPES[NSHAPES] = {{128, 64, 256, 0, 256, 9}, {512, 32, 512, 0, 512, 6}, {256, 64, 512, 0, 512, 3},
                                   {256, 64, 512, 2, 512, 3}, {128, 32, 256, 0, 256, 6},
                                   {256, 64, 512, 3, 512, 3}, {128, 64, 512, 3, 512, 2}, {128, 64, 512, 1, 256, 9},
                                   {256, 32, 512, 0, 512, 3}, {128, 32, 512, 1, 256, 3}, {256, 32, 512, 3, 512, 3}};

// EO >= 0 (data gradients whose operand set the planner emits in bulk): own instantiations of the C >= 64 3x3 block (the
// C <= 32 block at its 128-register budget spills 6 - 12 registers once the compiler may interleave the tiles)
template <typename T, typename TY, int KS, bool Q, bool PE, int EO = -1>
int dispatch(int shape, int nva, const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
    constexpr int EOC = (KS == 3 && (Q || EO == 48)) ? EO : -1;   // conv_core: data gradients and the bias + ReLU forward form (its other forward convs have the plain epilogue)
    constexpr int EOW = (KS == 3 && EO != 48) ? EO : -1;   // wave-specialised kernel: forward with statistics (EO 8) and data gradients
    switch (shape) {
        case 0:
            if (nva <= 3) return launch<T, TY, KS, 4, 1, 2, 4, 3, Q, PE>(k, grid, lds, st);
            if (nva <= 9) return launch<T, TY, KS, 4, 1, 2, 4, 9, Q, PE>(k, grid, lds, st);
            break;
        case 1:
            if (nva <= 6) return launch<T, TY, KS, 8, 1, 4, 2, 6, Q, PE>(k, grid, lds, st);
            break;
        case 2:   // the data-gradient form is sized for one block per CU (up to 256 VGPRs): a register cap alone spills
                  // (168 / 128 VGPRs: 49 / 160 spilled registers, 15.6 / 18.2 vs 14.64 ms per step in round 3)
            if (nva <= 3) return launch<T, TY, KS, 4, 2, 4, 2, 3, Q, PE, 1>(k, grid, lds, st);
            break;
        case 3:
            if constexpr (KS == 3) {
                if (nva <= 3 && !k.wres) return launch<T, TY, KS, 4, 2, 4, 2, 3, Q, PE, 1, 0, false, true, EOC>(k, grid, lds, st);
            }
            break;
        case 8:   // three (data gradient) resp. four waves per SIMD, no spills
            if (nva <= 3 && k.wres) return launch<T, TY, KS, 8, 1, 2, 2, 3, Q, PE, 4, 1, false, false, EOC>(k, grid, lds, st);
            if (nva <= 3) return launch<T, TY, KS, 8, 1, 2, 2, 3, Q, PE, (Q ? 3 : 4)>(k, grid, lds, st);
            break;
        case 4:
            if (nva <= 3 && k.wres) return launch<T, TY, KS, 4, 1, 2, 2, 3, Q, PE, 4, 1>(k, grid, lds, st);  // 113-123 VGPRs, no spills
            if (nva <= 3) return launch<T, TY, KS, 4, 1, 2, 2, 3, Q, PE, (Q ? 3 : 4)>(k, grid, lds, st);
            if (nva <= 6) return launch<T, TY, KS, 4, 1, 2, 2, 6, Q, PE, 3>(k, grid, lds, st);
            break;
        case 7:
            if (nva <= 3) return launch_ws<T, TY, KS, 2, 4, 3, Q, (Q ? -1 : EOW)>(k, grid, lds, st);   // (no stride-2 data gradient uses it)
            if (nva <= 9) return launch_ws<T, TY, KS, 2, 4, 9, Q, (Q ? -1 : EOW)>(k, grid, lds, st);
            break;
        case 9:
            if (nva <= 3) return launch_ws<T, TY, KS, 2, 2, 3, Q, EOW>(k, grid, lds, st);
            break;
    }
    return stl_set_error("conv: no kernel variant for block shape %d with %d staging vectors per thread", shape, nva);
}

// block-end source (STL_SRC_BNADD): the shapes the two-conv units of the network are planned with -- 256 px x 32 co
// (C <= 32), 256 px x 64 co (C = 64 / 128) and their small-map fallbacks; own instantiations, so that the
// data-gradient kernels (Q without ZM) carry none of this
static bool bnadd_shape_ok(int shape, int nva) { return nva <= 3 && (shape == 0 || shape == 2 || shape == 3 || shape == 4 || shape == 5 || shape == 6 || shape == 8 || shape == 10); }
template <typename T>
int dispatch_bnadd(int shape, int nva, const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
    if (nva <= 3) switch (shape) {
        case 0: return launch<T, T, 3, 4, 1, 2, 4, 3, true, true, 1, -1, true>(k, grid, lds, st);
        case 2:
        case 3: return launch<T, T, 3, 4, 2, 4, 2, 3, true, true, 1, -1, true>(k, grid, lds, st);   // (shape 3: image 0 of its layout)
        case 4: return launch<T, T, 3, 4, 1, 2, 2, 3, true, true, 3, -1, true>(k, grid, lds, st);
        case 8: return launch<T, T, 3, 8, 1, 2, 2, 3, true, true, 3, -1, true>(k, grid, lds, st);
    }
    return stl_set_error("conv: no block-end (BNADD) variant for block shape %d with %d staging vectors per thread", shape, nva);
}

template <typename T, typename TY>
int conv_backend(int path, const stl_conv& p, const ConvK& k, int shape, int nva, dim3 grid, size_t lds, hipStream_t st) {
    const bool q = p.src.mode == STL_SRC_BNBWD;
    if constexpr (!std::is_same<T, TY>::value) {   // mixed mode: only data gradients read forward tensors of another type
        if (!q) return stl_set_error("conv: ydtype differs from dtype, but the launch is not a data gradient (BNBWD source)");
    }
    if constexpr (std::is_same<T, f16>::value) {   // f16 tensors exist in the forward pass only
        if (q) return stl_set_error("conv: STL_F16 is a forward-tensor type; gradients are bf16 (dtype STL_BF16 + ydtype STL_F16)");
    }
    constexpr bool FWD = std::is_same<T, TY>::value, BWD = !std::is_same<T, f16>::value;
    if (path == 0) return run_1x1<T, TY>(p, st);
    if (path == 3) {
        if constexpr (sizeof(T) == 2) return dispatch_r2<T, TY>(p, k, shape == 5 ? 0 : (shape == 6 ? 1 : 2), grid, lds, st);
        return stl_set_error("conv_r2: 16-bit element types only");
    }
    if (path == 1) {
        if constexpr (FWD) return dispatch_bnadd<T>(shape, nva, k, grid, lds, st);
    }
    const bool plain = !p.bias && !p.addend && !p.mask_y && !p.mask_z && !p.red;   // forward convs of the pose network
    // compile-time epilogue operand set where the launch matches one of the planner's forms (STL_CONV_NO_EO=1: never -- A/B)
    const char* noeo = getenv("STL_CONV_NO_EO");   // bit 0: conv_core / conv_ws, bit 1: conv1x1
    const int eo = (noeo && (atoi(noeo) & 1)) ? -1 : epi_code(p);
    if (p.ks == 3) {
        if (q) {
            if constexpr (BWD) {
                if (eo == 1) return dispatch<T, TY, 3, true, false, 1>(shape, nva, k, grid, lds, st);
                if (eo == 7) return dispatch<T, TY, 3, true, false, 7>(shape, nva, k, grid, lds, st);
                if (eo == 2) return dispatch<T, TY, 3, true, false, 2>(shape, nva, k, grid, lds, st);
                if (eo == 0) return dispatch<T, TY, 3, true, false, 0>(shape, nva, k, grid, lds, st);
                return dispatch<T, TY, 3, true, false>(shape, nva, k, grid, lds, st);
            }
        }
        if constexpr (FWD) {
            if (plain && eo == 8) return dispatch<T, T, 3, false, true, 8>(shape, nva, k, grid, lds, st);
            if (!plain && eo == 48) return dispatch<T, T, 3, false, false, 48>(shape, nva, k, grid, lds, st);
            return plain ? dispatch<T, T, 3, false, true>(shape, nva, k, grid, lds, st) : dispatch<T, T, 3, false, false>(shape, nva, k, grid, lds, st);
        }
    }
    if (q) {
        if constexpr (BWD) return dispatch<T, TY, 1, true, false>(shape, nva, k, grid, lds, st);
    }
    if constexpr (FWD) return plain ? dispatch<T, T, 1, false, true>(shape, nva, k, grid, lds, st) : dispatch<T, T, 1, false, false>(shape, nva, k, grid, lds, st);
    return stl_set_error("conv: no kernel for this dtype combination");
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
    if (ws == 3) {   // conv_r2: [cs | cm | two halo planes of 32 B per pixel | unpadded filter image | per-tile sums]
        int off = 3 * cipad * 4;
        const int off_cm = off;
        off = (off + 4 * bco * 4 + 15) & ~15;
        const int off_a = off, pl = ((HR * HC * R2_PS) + 15) & ~15;
        off += 2 * pl;
        const int off_b = (off + 1023) & ~1023;   // 1-KB DMA pieces
        off = off_b + bco * R2_ROWF;
        const int off_red = off;
        off += 512 * 4;   // [WM][2][BCO] floats: 4 x 2 x 64 = 8 x 2 x 32
        if (out) {
            out->HR = HR, out->HC = HC, out->HP = HR * HC, out->nchunks = nchunks, out->cipad = cipad;
            out->off_cs = 0, out->off_cm = off_cm, out->off_a = off_a, out->off_b = off_b, out->off_red = off_red;
            out->sz_a = pl, out->sz_b = 0, out->wres = 0;
        }
        return (size_t)off;
    }
    int off = 3 * cipad * 4;
    const int off_cm = off;
    off += 4 * bco * 4;
    off = (off + 15) & ~15;
    const int off_a = off;
    const int sz_a = ((HR * HC * PSA) + 15) & ~15;
    off += sz_a * (ws ? 2 : 1);
    const int off_b = off;
    const int sz_b = bco * (taps * 64 + 32);
    // wave-specialised kernel: keep the whole filter slab (all chunks) resident when it fits
    const bool resident = nchunks == 1 || (ws == 1 && (size_t)off + (size_t)sz_b * nchunks <= 150 * 1024);
    off += resident ? sz_b * nchunks : sz_b * (ws ? 2 : 1);
    const int off_red = off_a;  // reused after the last stage
    const int red = 8 * 2 * bco * 4;
    if (off - off_a < red) off = off_a + red;
    if (out) {
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
    // kernel family (measured on MI355X): the wave-specialised kernel for stride-2 convolutions and for the
    // small, deep maps (Co >= 256 on <= 16384 pixels: their chains are the critical path of stage 4), the uniform kernel elsewhere
    // (STL_CONV_R2 bit 3, default since round 5: these layers on the two-per-CU kernel's 128-pixel form instead -- 13.94 -> 13.83 ms per step)
    const int r2_env = getenv("STL_CONV_R2") ? atoi(getenv("STL_CONV_R2")) : 11;
    const bool deep_small = p.stride == 1 && p.ks == 3 && p.Co >= 256 && (int64_t)p.B * p.Ho * p.Wo <= 16384 &&
                            !((r2_env & 8) && p.dtype != STL_F32 && p.Ci % 32 == 0 && p.Ci >= 64 && (r2_env & (p.src.mode == STL_SRC_BNBWD ? 2 : 1)));
    const bool want_ws = p.stride == 2 || deep_small;
    // 512 px x 32 co blocks: Co <= 32 with many input channels on a large map (8 chunks of K per tile)
    const bool wide_k = p.Co <= 32 && p.Ci >= 128 && p.stride == 1 && p.ks == 3 && !p.stuff && (int64_t)p.B * p.Ho * p.Wo >= 65536;
    // C <= 32 3x3 stride-1 layers: 256 px x 32 co blocks of 8 waves (shape 8) win END TO END although they lose in isolation
    // to the 128 px x 32 co blocks of 4 waves (C = 32 at 96x72: 19.8 vs 17.9 us alone, 16.27 vs 16.47 ms per step): two
    // 128-pixel halves share one copy of the filters and one halo tile, i.e. fewer bytes per launch and half as many blocks
    // competing for the CUs.  Shape 4 stays the fallback where no 256-pixel tile fits (small maps).
    const bool c32 = p.Co <= 32 && p.stride == 1 && p.ks == 3 && !wide_k;
    for (int shape = 0; shape < NSHAPES; ++shape) {
        const Shape sh = SHAPES[shape];
        if (sh.px == 0) continue;                    // retired shape ids
        if (sh.px > 128 && p.stride == 2) continue;  // stride-2 halos only fit the small blocks
        if ((sh.ws == 1) != want_ws) continue;
        if (want_ws && shape != (deep_small ? 9 : 7)) continue;
        // two LDS images (shape 3) wherever the 256 x 64 block walks K in more than one chunk: the staging of chunk c + 1
        // hides behind the MFMAs of chunk c.  STL_CONV_DB=0: the single-image kernel everywhere (A/B).
        if (shape == 2 || shape == 3 || shape == 5 || shape == 6) {
            static const bool db_on = !(getenv("STL_CONV_DB") && atoi(getenv("STL_CONV_DB")) == 0);
            // STL_CONV_R2 (A/B): bit 0 forward convolutions, bit 1 data gradients on the two-per-CU kernel (else the one-per-CU
            // two-image block, shape 3); bit 2: forward convolutions on its 128-pixel form as well; bit 3: the deep small maps
            // (Co >= 256 on <= 16384 pixels) on its 128-pixel form instead of the wave-specialised kernel.  Default 11.
            const int r2_mode = r2_env;   // (read per plan: plans are made once per layer)
            const bool db = db_on && p.ks == 3 && p.stride == 1 && p.Ci > ck;
            const bool dgrad = p.src.mode == STL_SRC_BNBWD;   // (the planner sets the source before it asks for a plan)
            const bool r2 = db && (r2_mode & (dgrad ? 2 : 1)) && p.dtype != STL_F32 && p.Ci % 32 == 0 && p.Ci >= 64;
            const bool small_map = (r2_mode & 8) && p.Co >= 256 && (int64_t)p.B * p.Ho * p.Wo <= 16384;
            const int want = r2 ? ((dgrad || (r2_mode & 4) || small_map) ? 6 : 5) : (db ? 3 : 2);
            if (shape != want) continue;
        }
        if ((shape == 1 || shape == 8 || shape == 10) && p.Co > 32) continue;
        if ((shape == 4 || shape == 8 || shape == 10) && !c32) continue;
        if (c32 && shape != 4 && shape != 8 && shape != 10) continue;
        {   // C <= 32 on the two-per-CU kernel's 256 x 32 form (STL_CONV_R2 bit 4) instead of conv_core_kernel's shape 8
            const bool r2c32 = (r2_env & 16) && p.dtype != STL_F32 && p.Ci == 32 && (r2_env & (p.src.mode == STL_SRC_BNBWD ? 2 : 1));
            if (c32 && (shape == 8 || shape == 10) && (shape == 10) != r2c32) continue;
        }
        if (wide_k && shape != 1) continue;
        const int nblk_co = ceil_div(p.Co, sh.co);
        for (int tw = (p.Wo < 4 ? p.Wo : 4); tw <= p.Wo && tw <= sh.px; ++tw)
          for (int frac = 4; frac >= 1; --frac) {
            int th = (sh.px / tw) * frac / 4;
            if (th > vrows) th = vrows;
            if (th < 1) continue;
            const int hr = (th - 1) * p.stride + p.ks, hc = (tw - 1) * p.stride + p.ks;
            const int nva = ceil_div(hr * hc * 4, sh.lthr);
            if (nva > sh.nva_max) continue;
            const size_t lds = lds_bytes(p, shape, th, tw, ck, nullptr);
            if (shape == 4 && lds > 40 * 1024 && nva <= 3) continue;  // keep four blocks per CU
            if (shape == 8 && lds > 80 * 1024) continue;              // ... resp. two 8-wave blocks
            if ((shape == 5 || shape == 6 || shape == 10) && lds > 76 * 1024) continue;   // two per CU in mixed company
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
            if (c32 && shape == 4) cost *= 4.0;   // fallback only
            if (shape == 5 || shape == 6 || shape == 10) cost *= 0.98;   // wins ties against the four-wave 128 x 64 block (small problems: both at the wave-count penalty's cap)
            if (cost < best.cost) best = Plan{shape, th, tw, lds, cost};
        }
    }
    return best;
}

}  // namespace

#if STL_HAS_BF16
static int ydtype_of(const stl_conv& p) { return (p.dtype == STL_BF16 && p.ydtype == STL_F16) ? STL_F16 : p.dtype; }   // 0 = same as dtype
int stl_conv_backend_bf16(int path, const stl_conv& p, const ConvK& k, int shape, int nva, dim3 grid, size_t lds, hipStream_t st) {
    if (ydtype_of(p) == STL_F16) return conv_backend<__bf16, f16>(path, p, k, shape, nva, grid, lds, st);   // mixed-mode data gradient
    return conv_backend<__bf16, __bf16>(path, p, k, shape, nva, grid, lds, st);
}
#endif
#if STL_HAS_F32
int stl_conv_backend_f32(int path, const stl_conv& p, const ConvK& k, int shape, int nva, dim3 grid, size_t lds, hipStream_t st) {
    return conv_backend<float, float>(path, p, k, shape, nva, grid, lds, st);
}
#endif
#if STL_HAS_F16
int stl_conv_backend_f16(int path, const stl_conv& p, const ConvK& k, int shape, int nva, dim3 grid, size_t lds, hipStream_t st) {
    return conv_backend<f16, f16>(path, p, k, shape, nva, grid, lds, st);
}
#endif

#if STL_HAS_BF16   // the C ABI entry points live in the bf16 (or the only) unit
#ifdef STL_STAMPS
extern "C" int stl_debug_conv_stamps(long long* host12) {
    return hipMemcpyFromSymbol(host12, HIP_SYMBOL(g_stamps), 14 * sizeof(long long)) == hipSuccess ? 0 : stl_set_error("stamps: copy failed");
}
extern "C" int stl_debug_conv_stamps2(long long* host64) {
    return hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_stamps2), 64 * sizeof(long long)) == hipSuccess ? 0 : stl_set_error("stamps: copy failed");
}
#endif

extern "C" int stl_conv_plan(stl_conv* pp) {
    stl_conv& p = *pp;
    STL_CHECK(p.dtype == STL_F32 || p.dtype == STL_BF16 || p.dtype == STL_F16, "conv_plan: bad dtype");
    STL_CHECK((p.ks == 1 || p.ks == 3) && (p.stride == 1 || p.stride == 2) && p.Ci > 0 && p.Co > 0, "conv_plan: bad geometry");
    const int ck = p.dtype == STL_F32 ? 16 : 32;
    Plan plan = choose_plan(p, ck);
    STL_CHECK(plan.shape >= 0, "conv_plan: no tile fits LDS for %dx%d ks %d stride %d Ci %d", p.Ho, p.Wo, p.ks, p.stride, p.Ci);
    p.shape = plan.shape, p.TH = plan.TH, p.TW = plan.TW;
    return 0;
}

extern "C" int stl_conv_bnadd_ok(const stl_conv* pp) {
    const stl_conv& p = *pp;
    if (use_1x1(p)) return 1;   // the streaming 1x1 kernel's block-end form (conv1x1.inc, ZM)
    if (!(p.ks == 3 && p.stride == 1 && !p.stuff && p.shape >= 0 && p.shape < NSHAPES && p.TH > 0 && p.TW > 0)) return 0;
    const Shape sh = SHAPES[p.shape];
    const int nva = ceil_div((p.TH + 2) * (p.TW + 2) * 4, sh.lthr);
    return bnadd_shape_ok(p.shape, nva) ? 1 : 0;
}

extern "C" int stl_conv_forward(const stl_conv* pp, void* stream) {
    const stl_conv& p = *pp;
    STL_CHECK(p.dtype == STL_F32 || p.dtype == STL_BF16 || p.dtype == STL_F16, "conv: bad dtype %d", p.dtype);
    STL_CHECK(p.ydtype == 0 || p.ydtype == p.dtype || (p.dtype == STL_BF16 && p.ydtype == STL_F16), "conv: ydtype %d does not go with dtype %d", p.ydtype, p.dtype);
    STL_CHECK(p.ks == 1 || p.ks == 3, "conv: ks must be 1 or 3 (got %d)", p.ks);
    STL_CHECK(p.stride == 1 || p.stride == 2, "conv: stride must be 1 or 2");
    STL_CHECK(!(p.stuff && p.stride != 1), "conv: stuff requires stride 1");
    STL_CHECK(!(p.ks == 1 && (p.stride != 1 || p.stuff)), "conv: 1x1 must be stride 1");
    const int kv = p.dtype == STL_F32 ? 4 : 8, ck = 4 * kv;
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
    STL_CHECK(!zm || (((p.ks == 3 && p.stride == 1 && !p.stuff) || use_1x1(p)) && p.src.y && p.src.beta && (p.src.stats || (p.src.rmean && p.src.rvar))),
              "conv: a BNADD source needs a 3x3 stride-1 (or wide 1x1) convolution, the skip tensor in src.y and BatchNorm parameters");
    STL_CHECK(zm || !p.src_out, "conv: src_out needs a BNADD source");
    STL_CHECK(!zm || (!p.bias && !p.addend && !p.mask_y && !p.mask_z && !p.red), "conv: a BNADD source takes no epilogue operands");
    STL_CHECK(p.src.mode == STL_SRC_PLAIN || p.src.gamma, "conv: BN source without gamma");
    STL_CHECK(p.src.mode != STL_SRC_BN || p.src.beta, "conv: BN source without beta");
    STL_CHECK(p.src.mode != STL_SRC_BN || p.src.stats || (p.src.rmean && p.src.rvar), "conv: BN source without statistics");
    STL_CHECK(p.src.mode != STL_SRC_BNBWD || (p.src.y && p.src.stats && p.src.rstats), "conv: BNBWD source incomplete");
    STL_CHECK(!(p.out_stats && p.red), "conv: out_stats and red are exclusive");
    STL_CHECK(!p.red || p.mask_y, "conv: red needs mask_y");
    STL_CHECK(!p.mask_y || (p.mask_bn.gamma && p.mask_bn.beta && (p.mask_bn.stats || (p.mask_bn.rmean && p.mask_bn.rvar))), "conv: mask BN incomplete");

    const auto backend = p.dtype == STL_BF16 ? stl_conv_backend_bf16 : (p.dtype == STL_F16 ? stl_conv_backend_f16 : stl_conv_backend_f32);
    if (use_1x1(p))  // wide 1x1 convolutions: streaming GEMM kernel (conv1x1.inc)
        return backend(0, p, ConvK{}, 0, 0, dim3(1), 0, (hipStream_t)stream);

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
    // an operand set / source conv_r2 has no instantiation for: conv_core_kernel's block of the same pixel count, same tile
    if (plan.shape == 5 && !r2_takes(p, 0)) plan.shape = 3;
    if (plan.shape == 6 && !r2_takes(p, 1)) plan.shape = 0;
    if (plan.shape == 10 && !r2_takes(p, 2)) plan.shape = 8;
    {
        const Shape shp = SHAPES[plan.shape];
        const int nv = ceil_div(((plan.TH - 1) * p.stride + p.ks) * ((plan.TW - 1) * p.stride + p.ks) * 4, shp.lthr);
        STL_CHECK(nv <= shp.nva_max, "conv: tile %dx%d has too large a halo for block shape %d", plan.TH, plan.TW, plan.shape);
    }
    ConvK k;
    k.p = p;
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
    // 768 (not 1024) blocks for the 4-wave shapes: three resident blocks per CU leave room for the co-running kernels
    // (16.62 -> 16.56 ms per step in round 2); 432 for the 8-wave 256 px x 32 co shape (two blocks per CU on most CUs; the 864
    // tiles of a 96 x 72 map at batch 32 are exactly two per block -- 384 blocks gave a quarter of them a third tile: 13.33 -> 13.28 ms)
    int cap = sh.thr == 512 ? 256 : 768;
    if (plan.shape == 8) cap = getenv("STL_CONV_CAP8") ? atoi(getenv("STL_CONV_CAP8")) : 432;
    if (plan.shape == 5) cap = 512;   // two per CU
    if (plan.shape == 6) cap = 768;   // three per CU (<= 55 KB of LDS, <= 128 VGPRs)
    if (plan.shape == 10) cap = getenv("STL_R2_CAP10") ? atoi(getenv("STL_R2_CAP10")) : 768;   // <= 45 KB of LDS: three per CU
    cap = std::max(8, (cap / k.ny + 7) / 8 * 8);
    if (const char* e = getenv("STL_CONV_GRID_CAP")) cap = atoi(e) > 0 ? (atoi(e) + 7) / 8 * 8 : cap;   // tools/conv_probe.py
    if (gx > cap) gx = cap;
    dim3 grid(gx * k.ny, 1);
    const int nva = ceil_div(k.HP * 4, sh.lthr);
    if (getenv("STL_CONV_DEBUG"))
        fprintf(stderr, "[stl conv] %dx%d Ci%d Co%d ks%d s%d: shape=%d tile=%dx%d npt=%d grid=(%d x %d) lds=%zu nva=%d nchunks=%d\n", p.Ho,
                p.Wo, p.Ci, p.Co, p.ks, p.stride, plan.shape, plan.TH, plan.TW, k.npt, gx, k.ny, lds, nva, k.nchunks);
    hipStream_t st = (hipStream_t)stream;
    return backend((plan.shape == 5 || plan.shape == 6 || plan.shape == 10) ? 3 : (zm ? 1 : 2), p, k, plan.shape, nva, grid, lds, st);
}
#endif   // STL_HAS_BF16
