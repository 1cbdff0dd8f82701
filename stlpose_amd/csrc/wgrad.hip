// Weight gradient of the implicit-GEMM convolution on gfx950.
//   dw[co][tap][ci] = sum over output pixels of  g[pixel][co] * h[pixel*stride + tap - pad][ci]
// GEMM view: M = 32 output channels, N = 32 input channels per tap, K = output pixels.
// Both operands are K-major in memory (NHWC), so fragments are read TRANSPOSED from LDS:
// ds_read_b64_tr_b16 for bf16, plain 4-byte reads for fp32 (one element per lane per MFMA).
// Each of the 4 waves owns one 16x16 quadrant of the block's 32x32 channels for all taps (4 x taps accumulator
// registers) and multiplies every pixel of every tile of the block; the block writes one fp32 slab
// [Co][taps][Ci] (deterministic split-K over blocks; slabs are summed by stl_reduce_slabs).
// Pipeline: two register sets of staged loads -- the global loads of tile t+2 are issued (unconditionally,
// clamped addresses) before the MFMAs of tile t; index arithmetic is hoisted out of the loop.
#include <type_traits>
#include "common.cuh"

// Compiled twice by stlpose_amd/build.py (-DSTL_DT=1: bf16 kernels + the C ABI entry points, -DSTL_DT=0: fp32 kernels) so that
// the two halves build in parallel; STL_DT=2 (default) = one unit.
#ifndef STL_DT
#define STL_DT 2
#endif

constexpr int WG_MAXG = STL_WGRAD_GROUP_MAX;
struct WgK {
    stl_wgrad p;           // problem 0 (geometry, tile, nsplit: shared by every member of a group)
    stl_wgrad_io io[WG_MAXG];   // per member: sources and slab pointer (io[0] mirrors p)
    int ng;                // members of this launch (1 = plain launch); grid.x = ng * p.nsplit
    int dbg;
    int tiles_c, npt, HR, HC, HP, PI, pad, taps;
    int psg, psh;  // LDS bytes per pixel (32 channels + 16 B pad)
    int off_cg, off_ch, off_g, off_h;
    float r_TW, r_HC, r_tc, r_vp, r_PI;  // reciprocals for fdiv
    unsigned long long m_tc, m_vp, m_PI;  // ceil(2^32 / d): exact n / d = (n * m) >> 32 on the scalar unit for n * d < 2^32
    int xmap, units, gy, gz;   // XCD-aware block order (see wg_block): units = ng * nsplit pixel ranges, gy x gz channel chunks
};

// dtype back ends (defined where their kernels are instantiated); wide = the 64 x 64-channel variant (bf16 only)
int stl_wgrad_backend_bf16(bool wide, const WgK& k, dim3 grid, size_t lds, hipStream_t st);
int stl_wgrad_backend_f32(bool wide, const WgK& k, dim3 grid, size_t lds, hipStream_t st);

namespace {

#include "conv_common.inc"

#ifdef STL_STAMPS
#include "../../include/stlpose_hip_debug.h"
// debug-only phase stamps (block 0, thread 0; STL_CONV_STAMPS=1): never read by the kernel
__device__ long long g_wstamps[16];
__device__ long long g_wstamps2[64];   // per tile of block 0: [4 i .. 4 i + 3] = tile start, staged image written + barrier, loads of tile i + 2 requested, MFMAs + barrier
#define WSTAMP(i)                                                                                   \
    do {                                                                                            \
        if (k.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_wstamps[i] = wall_clock64(); \
    } while (0)
#define WSTAMP2(i)                                                                                  \
    do {                                                                                            \
        if (k.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && (i) < 64) g_wstamps2[i] = wall_clock64(); \
    } while (0)
#else
#define WSTAMP(i) do {} while (0)
#define WSTAMP2(i) do {} while (0)
#endif

// a * b + c with 24-bit a, b (b uniform): full rate, and kept out of reach of the 64-bit mad combine
__device__ __forceinline__ int mad24_vsv(int a, int b_uniform, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}

// n / d for uniform n >= 0 with m = ceil(2^32 / d): integer only, so it stays on the scalar unit (fdiv's conversions are
// vector instructions, and everything computed from their result would be vector arithmetic too)
__device__ __forceinline__ int sdiv(int n, unsigned long long m) { return (int)(((unsigned long long)(uint32_t)n * m) >> 32); }


// Which (member, split, co chunk, ci chunk) a block works on.  Plain order: grid (ng * nsplit, gy, gz).  XCD-aware order
// (k.xmap, 1-D grid): workgroups go to the eight XCDs round-robin by linear id, so the gy * gz chunk blocks of ONE pixel
// range get ids 8 apart -- same XCD, dispatched back to back -- and share its L2: every chunk block of a range reads the
// same pixels (its 32- or 64-channel slice of g, and of h), and two adjacent 32-channel slices share each 128-byte line.
struct WgBlock { int member, bsplit, cy, cz; bool live; };
__device__ __forceinline__ WgBlock wg_block(const WgK& k) {
    WgBlock b;
    if (!k.xmap) {
        b.member = __builtin_amdgcn_readfirstlane(blockIdx.x / k.p.nsplit);
        b.bsplit = blockIdx.x - b.member * k.p.nsplit, b.cy = blockIdx.y, b.cz = blockIdx.z, b.live = true;
        return b;
    }
    const int nc = k.gy * k.gz, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int ju = __builtin_amdgcn_readfirstlane(j / nc), chunk = j - ju * nc, unit = ju * 8 + xcd;
    b.live = unit < k.units;
    b.member = __builtin_amdgcn_readfirstlane(unit / k.p.nsplit), b.bsplit = unit - b.member * k.p.nsplit;
    b.cy = __builtin_amdgcn_readfirstlane(chunk / k.gz), b.cz = chunk - b.cy * k.gz;
    return b;
}


// Which pixel of a 32-pixel K step a lane's transposed read fetches (16-bit types): register i of lane (g = lane >> 4,
// j = (lane & 15) >> 2) is MFMA k index 8 g + 4 i + j, and WHICH pixel sits at a k index is ours to choose as long as both
// operands agree.  ds_read_b64_tr_b16 is served in two groups of 32 lanes (g = 0, 1 / g = 2, 3), eight rows of 32 bytes each:
// with pixel = 8 g + 4 i + j (rounds 2-4) the rows are {4 i .. 4 i + 3} and {8 + 4 i ..}, eight apart, and rows eight apart
// share a bank window at every pixel stride that is a multiple of 32 bytes (and strides that are not misalign the windows): every
// fragment read took 4 LDS cycles instead of 2.  With pixel = 16 (g >> 1) + 8 i + 4 (g & 1) + j the eight rows are consecutive.
__device__ __forceinline__ int krow16(int g, int i, int lane) { return 16 * (g >> 1) + 8 * i + 4 * (g & 1) + ((lane & 15) >> 2); }

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

// Same fragments from per-lane LDS byte addresses (lane term and row offset folded in) plus a compile-time offset:
// with the address register opaque to the optimiser inside the tile loop the offset lands in the instruction's
// immediate field (otherwise LLVM hoists one address register PER READ out of the loop: 36 VGPRs for 3x3).
template <typename T>
__device__ __forceinline__ V16 frag_at(const uint32_t* addr, int off);
template <>
__device__ __forceinline__ V16 frag_at<__bf16>(const uint32_t* addr, int off) {
    V16 v;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)(addr[h] + off));
        const uint64_t bits = __builtin_bit_cast(uint64_t, r);
        v.w[2 * h] = (uint32_t)bits;
        v.w[2 * h + 1] = (uint32_t)(bits >> 32);
    }
    return v;
}
template <>
__device__ __forceinline__ V16 frag_at<float>(const uint32_t* addr, int off) {
    V16 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v.w[s] = *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)(addr[s] + off);
    return v;
}

// NVH: h (input halo) staging vectors per thread; GQ: g is BNBWD (second tensor on load)
// OCC: blocks per CU the register budget is sized for
//
// Work split: wave w owns ONE 16 x 16 quadrant (mt = w >> 1, nt = w & 1) of the block's 32 x 32 channels for ALL taps
// and ALL pixels of every tile (M/N split).  The earlier K split (each wave a quarter of the pixels, all four
// quadrants) needed 144 accumulator registers per lane and a cross-wave reduction through LDS at the end; this one
// needs 4 * TAPS = 36, no reduction, and leaves room for TWO tiles of staged loads per thread: the loads of tile
// t + 2 are issued while tile t is multiplied, so a tile's memory round trip (~2 us when 256 blocks load at once)
// is covered by two iterations instead of being exposed once per tile (it was ~80 % of the K loop).
// TY: element type of the FORWARD tensors (h.x, g.y); T: the gradient dt = g.x and the MFMA operands
// CH: channels per block in each direction (round 5).  32: the block's 32 x 32 channels are four 16 x 16 quadrants, NW / 4 tap
// groups.  64 (NW = 16, 1024 threads, bf16, 3x3): 64 x 64 channels = sixteen quadrants, one per wave, all nine taps (36
// accumulator registers): ONE staged pixel tile feeds four times the MFMAs -- with 32-channel blocks a C = 64 layer stages every
// pixel four times (once per chunk pair), a C = 128 layer sixteen times, and the staging (loads, BatchNorm-backward transform,
// LDS writes, barriers), not the matrix pipe, is what a tile costs.  The per-thread shape is the 32-channel kernel's (one g and
// two h staging vectors, twice the threads for twice the channels in each tensor); 128 registers for 16 waves means ONE register
// set of staged loads (NSET = 1: the next tile is requested behind the barrier and lands during the 36 MFMAs per wave).
// NSET: register sets of staged loads (2: tile t + 2 in flight while tile t is multiplied).
template <typename T, int KS, int NVH, bool GQ, int TPX, int OCC, int NW, typename TY = T, int CH = 32, int NSET = 2>
__global__ __launch_bounds__(64 * NW, OCC) void wgrad_kernel(const WgK k) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KV = ET<T>::KV, TAPS = KS * KS;
    constexpr int KSTEP = 4 * KV;               // pixels per MFMA K step (32 bf16 / 16 f32)
    constexpr int NR = sizeof(T) == 2 ? 2 : 4;  // row offsets a lane needs per fragment
    constexpr int VPX = CH / KV;                // 16-byte vectors per pixel (CH channels)
    constexpr int NT = 64 * NW;                 // threads per block
    constexpr int NVG = TPX * VPX / NT;         // g staging vectors per thread
    static_assert(NVG >= 1 && NVG * NT == TPX * VPX, "g staging slots must tile the block");
    constexpr int QW = CH / 16, WPG = QW * QW;  // 16 x 16 quadrants per side / waves per tap group (one quadrant each)
    static_assert((CH == 32 || CH == 64) && NW % WPG == 0 && (NSET == 1 || NSET == 2), "channel tile / wave count");
    constexpr int NTG = NW / WPG;               // tap groups (CH = 32, NW = 8: waves 0-3 take taps 0..4, waves 4-7 taps 5..8)
    constexpr int TPG = (TAPS + NTG - 1) / NTG; // taps per group (accumulators per wave)
    // LDS bytes per pixel (== k.psg == k.psh): compile-time, so that tap and K-step offsets of the fragment reads are
    // instruction immediates.  16-bit types: 32 bytes of padding, an ODD multiple of 32 in total -- with the K order of
    // krow() below the eight pixel rows one transposed read touches per 32 lanes are CONSECUTIVE pixels, and
    // consecutive rows of 96 (160) bytes fall into the eight disjoint 32-byte bank windows: conflict-free.
    constexpr int PS = CH * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);
    constexpr int NKT = TPX / KSTEP;            // K steps per tile
    const stl_wgrad& p = k.p;
    // grouped launch: blockIdx.x = member * nsplit + split.  The member's tensors come from k.io[member] (a uniform,
    // run-time index into the kernel arguments: scalar loads), everything geometric from k.p.
    const WgBlock wb = wg_block(k);
    if (!wb.live) return;
    const int member = wb.member, bsplit = wb.bsplit;
    const stl_wgrad_io& io = k.io[member];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int mt = (wave % WPG) / QW, nt = wave % QW, tg = wave / WPG;
    const int co0 = wb.cy * CH, ci0 = wb.cz * CH;
    WSTAMP(0);
    float* cgc = reinterpret_cast<float*>(smem + k.off_cg);  // [3][CH]
    float* chc = reinterpret_cast<float*>(smem + k.off_ch);  // [2][CH]
    char* sG = smem + k.off_g;
    char* sH = smem + k.off_h;

    const int tilepx = p.TH * p.TW;
    const int vpitch = p.Ho + 1;
    const int hrow = k.HC * PS;   // LDS bytes per halo row

    // ---- loop-invariant staging descriptors
    int g_yx[NVG];  // (ty << 16) | tx of the tile pixel of slot i, -1 = beyond the tile (zero row)
    const int g_part = tid % VPX;
#pragma unroll
    for (int i = 0; i < NVG; ++i) {
        const int m = (tid + i * NT) / VPX;
        g_yx[i] = -1;
        if (m < tilepx) {
            const int ty = fdiv(m, k.r_TW);
            g_yx[i] = (ty << 16) | (m - ty * p.TW);
        }
    }
    int h_rc[NVH];
#pragma unroll
    for (int i = 0; i < NVH; ++i) {
        const int v = tid + i * NT;
        h_rc[i] = -1;
        if (v < k.HP * VPX) {
            const int hp = v / VPX, hr = fdiv(hp, k.r_HC);
            h_rc[i] = (hr << 16) | (hp - hr * k.HC);
        }
    }
    const bool g_chok = (co0 + g_part * KV) < p.Co, h_chok = (ci0 + g_part * KV) < p.Ci;

    // ---- fragment-read addresses (LDS byte addresses, lane term and this wave's quadrant folded in).  g rows are
    // linear in the K step (row = s * KSTEP + c), so one base per row register; h rows follow the tile geometry.
    const uint32_t lterm = sizeof(T) == 2 ? (lane & 3) * 8 : (lane & 15) * 4;
    uint32_t gaw[NR], hbw[NKT][NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = sizeof(T) == 2 ? krow16(g, i, lane) : 4 * g + i;
        gaw[i] = (uint32_t)(uintptr_t)sG + c * PS + lterm + mt * 16 * (int)sizeof(T);
#pragma unroll
        for (int s = 0; s < NKT; ++s) {
            int m = s * KSTEP + c;
            if (m >= tilepx) m = 0;   // rows >= tilepx are zero in sG: any valid h row will do
            const int ty = fdiv(m, k.r_TW), tx = m - ty * p.TW;
            hbw[s][i] = (uint32_t)(uintptr_t)sH + ((ty * p.stride) * k.HC + tx * p.stride) * PS + lterm + nt * 16 * (int)sizeof(T);
        }
    }

    // ---- register sets of staged loads
    V16 rgv[NSET][NVG], rgq[NSET][GQ ? NVG : 1], rhv[NSET][NVH];
    uint32_t okm[NSET] = {};   // validity bits of a set: bit i = g slot i, bit NVG + i = h slot i

    // unconditional loads (a guarded load would be serialised by the compiler); invalid slots read element 0 and
    // are zeroed when written to LDS
    // Byte offset of a slot = (tile term: scalar unit) + (slot term: computed once) - wraps * (row-pitch term, one 24-bit mad);
    // see wgrad_ws_kernel.  Straight-line, no quarter-rate multiplies; the host checks the ranges (dispatch).
    const int g_cbytes = p.Co * (int)sizeof(T), h_cbytes = p.Ci * (int)sizeof(T);
    const int g_wrap = -p.Wo * g_cbytes, h_wrap = (p.Hi - k.PI) * p.Wi * h_cbytes;
    const int n_vp = -vpitch, n_PI = -k.PI;
    uint32_t g_lo[NVG], h_lo[NVH], slot_ok = 0;
#pragma unroll
    for (int i = 0; i < NVG; ++i) {
        g_lo[i] = (uint32_t)(((g_yx[i] >> 16) * p.Wo + (g_yx[i] & 0xffff)) * g_cbytes + (co0 + g_part * KV) * (int)sizeof(T));
        if (g_yx[i] >= 0 && g_chok) slot_ok |= 1u << i;
    }
#pragma unroll
    for (int i = 0; i < NVH; ++i) {
        h_lo[i] = (uint32_t)(((h_rc[i] >> 16) * p.Wi + (h_rc[i] & 0xffff)) * h_cbytes + (ci0 + g_part * KV) * (int)sizeof(T));
        if (h_rc[i] >= 0 && h_chok) slot_ok |= 1u << (NVG + i);
    }
    // byte offsets of tile t's staging slots (0 for slots outside the tensors) and their validity bits
    auto tile_offsets = [&](int t, uint32_t* go, uint32_t* ho) __attribute__((always_inline)) -> uint32_t {
        const int tr = sdiv(t, k.m_tc), tc = t - tr * k.tiles_c;
        const int vr0 = tr * p.TH, c0 = tc * p.TW;
        const int gb0 = sdiv(vr0, k.m_vp), gy0 = vr0 - gb0 * vpitch;
        const uint32_t g_t = (uint32_t)(((vr0 - gb0) * p.Wo + c0) * g_cbytes);
        uint32_t ok = 0;
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            const int oy0 = gy0 + (g_yx[i] >> 16), c = c0 + (g_yx[i] & 0xffff);
            const int wr_ = fdiv(oy0 < 0 ? 0 : oy0, k.r_vp);
            const int oy = mad24_vsv(wr_, n_vp, oy0);
            const bool in = ((slot_ok >> i) & 1u) & (gb0 + wr_ < p.B) & (oy < p.Ho) & (c < p.Wo);
            go[i] = in ? (uint32_t)mad24_vsv(wr_, g_wrap, (int)(g_t + g_lo[i])) : 0u;
            ok |= in ? 1u << i : 0u;
        }
        const int vrs = vr0 * p.stride, cb = c0 * p.stride - k.pad;
        const int hb0 = sdiv(vrs, k.m_PI), hy0 = vrs - hb0 * k.PI - k.pad;
        const uint32_t h_t = (uint32_t)(((hb0 * p.Hi + hy0) * p.Wi + cb) * h_cbytes);
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            const int iy0 = hy0 + (h_rc[i] >> 16), ix = cb + (h_rc[i] & 0xffff);
            const int wr_ = fdiv(iy0 < 0 ? 0 : iy0, k.r_PI);
            const int iy = mad24_vsv(wr_, n_PI, iy0);
            const bool in = ((slot_ok >> (NVG + i)) & 1u) & (iy0 >= 0) & (ix >= 0) & (ix < p.Wi) & (hb0 + wr_ < p.B) & (iy < p.Hi);
            ho[i] = in ? (uint32_t)mad24_vsv(wr_, h_wrap, (int)(h_t + h_lo[i])) : 0u;
            ok |= in ? 1u << (NVG + i) : 0u;
        }
        return ok;
    };
    auto load_set = [&](auto SET, const uint32_t* go, const uint32_t* ho, uint32_t ok) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            rgv[S][i] = ldg16((const char*)io.g.x + go[i]);
            if (GQ) rgq[S][i] = ldg16((const char*)io.g.y + go[i]);
        }
#pragma unroll
        for (int i = 0; i < NVH; ++i) rhv[S][i] = ldg16((const char*)io.h.x + ho[i]);
        okm[S] = ok;
    };
    auto fetch = [&](auto SET, int t) __attribute__((always_inline)) {
        uint32_t go[NVG], ho[NVH];
        const uint32_t ok = tile_offsets(t, go, ho);
        load_set(SET, go, ho, ok);
    };
    const float relu_lo = io.h.relu ? 0.f : -INFINITY;
    auto write_lds = [&](auto SET) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        const int cl = g_part * KV;
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            V16 val = rgv[S][i];
            if (GQ) val = xform_bnbwd<T, TY>(val, rgq[S][i], cgc + cl, cgc + CH + cl, cgc + 2 * CH + cl);
            mask16(val, (okm[S] >> i) & 1u);
            const int v = tid + i * NT;
            *reinterpret_cast<V16*>(sG + (v / VPX) * PS + g_part * 16) = val;
        }
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            // the loaded registers are READ unconditionally (only the LDS store is predicated): a slot that is skipped
            // as a whole leaves its load pending on that path, and the wait-count pass then drains EVERY outstanding
            // load (the other set's too) in front of the next instruction that overwrites the register
            V16 val = rhv[S][i];
            if (io.h.mode == STL_SRC_BN) val = xform_bn<T, TY>(val, chc + cl, chc + CH + cl, relu_lo);
            else val = xform_cvt<T, TY>(val);
            mask16(val, (okm[S] >> (NVG + i)) & 1u);
            const int v = tid + i * NT;
            if (h_rc[i] >= 0) *reinterpret_cast<V16*>(sH + (v / VPX) * PS + g_part * 16) = val;
        }
    };

    f32x4 acc[TPG];
#pragma unroll
    for (int t = 0; t < TPG; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // all pixels of the staged tile x this wave's quadrant x all taps.  Fragment reads run PF (tap, K step) pairs
    // ahead of their MFMA through a register ring; the address registers are made opaque inside the loop so that
    // LLVM folds the tap / K-step offsets into the instructions instead of hoisting one address register per read.
    auto mfma_taps = [&](auto TAP0, auto NTAP) __attribute__((always_inline)) {
        constexpr int T0 = decltype(TAP0)::value, NTP = decltype(NTAP)::value;   // this wave's taps [T0, T0 + NTP)
        constexpr int NJ = NKT * NTP, PF = NTP >= 4 ? (CH == 64 ? 4 : 6) : 2;
        uint32_t ga[NR], hb[NKT][NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            ga[i] = gaw[i];
            asm volatile("" : "+v"(ga[i]));
#pragma unroll
            for (int s = 0; s < NKT; ++s) {
                hb[s][i] = hbw[s][i];
                asm volatile("" : "+v"(hb[s][i]));
            }
        }
        V16 aq[2], bq[PF];
        auto bfrag = [&](int j) __attribute__((always_inline)) {
            const int s = j / NTP, tap = T0 + j % NTP;
            uint32_t ad[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) ad[i] = hb[s][i] + (tap / KS) * hrow;
            return frag_at<T>(ad, (tap % KS) * PS);
        };
        aq[0] = frag_at<T>(ga, 0);
#pragma unroll
        for (int j = 0; j < PF - 1 && j < NJ; ++j) bq[j] = bfrag(j);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int s = j / NTP, jt = j % NTP;
            if (jt == 0 && s + 1 < NKT) aq[(s + 1) & 1] = frag_at<T>(ga, (s + 1) * KSTEP * PS);
            if (j + PF - 1 < NJ) bq[(j + PF - 1) % PF] = bfrag(j + PF - 1);
            mma16<T>(acc[jt], aq[s & 1], bq[j % PF]);
        }
    };
    auto mfma_tile = [&]() __attribute__((always_inline)) {
        if constexpr (NTG == 1) {
            mfma_taps(std::integral_constant<int, 0>{}, std::integral_constant<int, TAPS>{});
        } else {
            if (tg == 0) mfma_taps(std::integral_constant<int, 0>{}, std::integral_constant<int, TPG>{});          // wave-uniform
            else mfma_taps(std::integral_constant<int, TPG>{}, std::integral_constant<int, TAPS - TPG>{});
        }
    };

    const int step = p.nsplit;
    int t = bsplit;
    WSTAMP(1);
    // BatchNorm constants (the last 2 CH threads of waves 0-3: CH channels of g, then CH of h; CH = 32: wave 3): the statistics
    // loads are issued ahead of the first tiles' loads, the arithmetic runs while those are in flight
    SrcRaw raw;
    const int cidx = tid - (256 - 2 * CH);
    const bool cw = cidx >= 0 && cidx < 2 * CH, cg = cidx < CH;
    const int cch = cidx & (CH - 1);
    const bool cok = cw && (cg ? co0 + cch < p.Co : ci0 + cch < p.Ci);
    if (cok) {
        if (cg) src_raw_load(io.g, co0 + cch, p.Co, raw);
        else src_raw_load(io.h, ci0 + cch, p.Ci, raw);
    }
    constexpr std::integral_constant<int, 0> I0{};
    constexpr std::integral_constant<int, 1> I1{};
    if (t < k.npt) fetch(I0, t);
    if constexpr (NSET == 2) {
        if (t + step < k.npt) fetch(I1, t + step);
    }
    WSTAMP(2);
    if (cw) {
        float a = 0.f, b = 0.f, cc = 0.f;
        if (cok) {
            if (cg) src_raw_finish(io.g, raw, a, b, cc);
            else src_raw_finish(io.h, raw, a, b, cc);
        }
        if (cg) cgc[cch] = a, cgc[CH + cch] = b, cgc[2 * CH + cch] = cc;
        else chc[cch] = a, chc[CH + cch] = b;
    }
    __syncthreads();  // constants visible
    WSTAMP(3);
    bool first = true;
    [[maybe_unused]] int dbg_i = 0;
    auto body = [&](auto SET) __attribute__((always_inline)) {
        WSTAMP2(4 * dbg_i);
        write_lds(SET);
        __syncthreads();
        WSTAMP2(4 * dbg_i + 1);
        if (first) WSTAMP(4);
        // (A register-free L2 prefetch of tile t + 2 -- 4-byte LDS-DMA touches of every staging slot -- made the 16-wave block
        // SLOWER, 67.7 -> 73.6 us per eight-layer launch: its tile is bound by vector-instruction issue, not by the round trip.)
        if (t + NSET * step < k.npt) fetch(SET, t + NSET * step);   // block-uniform: NSET tiles ahead, into the set just drained
        WSTAMP2(4 * dbg_i + 2);
        mfma_tile();
        __syncthreads();
        WSTAMP2(4 * dbg_i + 3);
#ifdef STL_STAMPS
        ++dbg_i;
#endif
        if (first) WSTAMP(5);
        first = false;
        t += step;
    };
    // NOTE (round 3, from the ISA): hipcc's wait-count insertion does not realise the "two tiles in flight" intent fully.
    // The first use of set 0's registers in the loop gets vmcnt(2) / (1) / (0) -- a full drain, the other set's younger
    // loads included (the join of the prologue's and the back edge's pending-load orders is resolved conservatively; a
    // four-fold unrolled loop gets the same full drain at every second tile) -- and set 1's use then needs no wait at
    // all.  So one tile out of two is covered by a full iteration, the other by one MFMA phase.  Counted waits would
    // need the staging loads and their waits written as inline assembly.
    while (t < k.npt) {
        body(I0);
        if constexpr (NSET == 2) {
            if (t >= k.npt) break;
            body(I1);
        }
    }
    WSTAMP(6);
    {   // every wave writes its quadrant (and its taps) of the block's slab [Co][taps][Ci]
        float* slab = io.partial + (size_t)bsplit * p.Co * TAPS * p.Ci;
        const int ci = ci0 + nt * 16 + (lane & 15), co = co0 + mt * 16 + 4 * g;
        const int tap0 = tg * TPG;
        float* dst = slab + ((size_t)co * TAPS + tap0) * p.Ci + ci;
        if (ci < p.Ci) {
#pragma unroll
            for (int jt = 0; jt < TPG; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (tap0 + jt < TAPS && co + r < p.Co) dst[((size_t)r * TAPS + jt) * p.Ci] = acc[jt][r];
        }
    }
    WSTAMP(8);
}


template <typename T, typename TY, int KS, int NVH, bool GQ, int TPX = 128, int NW = 4, int CH = 32>
int launch(const WgK& k, dim3 grid, size_t lds, hipStream_t st) {
    constexpr int OCC = NW == 16 ? 4 : ((NVH * NW <= 24 && TPX == 128 && sizeof(T) == 2) ? 2 : 1);
    constexpr int NSET = CH == 64 ? 1 : 2;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, KS, NVH, GQ, TPX, OCC, NW, TY, CH, NSET>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    STL_LAUNCH((wgrad_kernel<T, KS, NVH, GQ, TPX, OCC, NW, TY, CH, NSET>), grid, dim3(64 * NW), lds, st, k);
    static char nbuf[160];
    static const char* nm = stl_kname<T>(nbuf, "wgrad_kernel", {KS, NVH, GQ, TPX, OCC, NW, stl_code<TY>(), CH});
    stl_note_kernel(nm, true);
    STL_LAUNCH_CHECK("conv_wgrad");
    return 0;
}


// ------------------------------------------------------------------------------------------------
// 64 x 64 variant (bf16, stride 1, Co >= 64 and Ci >= 64): the block owns 64 output x 64 input
// channels, each wave a 32 x 32 quadrant of them for ALL pixels of the tile (M/N split instead of
// the K split above).  Per staged byte it does twice the MFMA work of the 32 x 32 kernel (the
// activations are re-read Co/64 + Ci/64 times instead of Co/32 + Ci/32), a tile carries 4x the
// MFMAs per barrier pair, and there is no cross-wave reduction: every wave writes its own quadrant
// of the slab straight from the accumulators.  LDS pixel stride 160 B (128 B data + 32 B pad: the
// four rows of a transposed 16-lane read fall into disjoint bank groups).
template <typename T, int KS, int NVH, bool GQ, int TPX, typename TY = T>
__global__ __launch_bounds__(256) void wgrad64_kernel(const WgK k) {
    static_assert(sizeof(T) == 2, "bf16 only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KV = 8, TAPS = KS * KS, KSTEP = 32, NR = 2;
    constexpr int VPX = 8;                  // 16-byte vectors per pixel (64 channels)
    constexpr int NVG = TPX * VPX / 256;    // g staging vectors per thread
    constexpr int NKS = TPX / KSTEP;        // K steps per tile, all done by every wave
    const stl_wgrad& p = k.p;
    const WgBlock wb = wg_block(k);   // grouped launch, see wgrad_kernel
    if (!wb.live) return;
    const int member = wb.member, bsplit = wb.bsplit;
    const stl_wgrad_io& io = k.io[member];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int co0 = wb.cy * 64, ci0 = wb.cz * 64;
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
    for (int i = 0; i < NR; ++i) rg0[i] = krow16(g, i, lane) * k.psg;
#pragma unroll
    for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            int m = s * KSTEP + krow16(g, i, lane);
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
                { const int wr_ = fdiv(oy, k.r_vp); oy -= wr_ * vpitch, b += wr_; }
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
                    { const int wr_ = fdiv(iy, k.r_PI); iy -= wr_ * k.PI, b += wr_; }
                    if (b < p.B && iy < p.Hi) h_go[i] = ((b * p.Hi + iy) * p.Wi + ix) * p.Ci + ci0 + g_part * KV;
                }
            }
        }
    };
    auto issue = [&](bool en) {  // unconditional loads, clamped addresses
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            const size_t off = (en && g_go[i] >= 0) ? (size_t)g_go[i] : 0;
            rgv[i] = ldg16((const char*)io.g.x + off * sizeof(T));
            if (GQ) rgq[i] = ldg16((const char*)io.g.y + off * sizeof(T));
        }
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            const size_t off = (en && h_go[i] >= 0) ? (size_t)h_go[i] : 0;
            rhv[i] = ldg16((const char*)io.h.x + off * sizeof(T));
        }
    };
    const float relu_lo = io.h.relu ? 0.f : -INFINITY;
    auto write_lds = [&]() {
        const int cl = g_part * KV;
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            V16 val = rgv[i];
            if (GQ) val = xform_bnbwd<T, TY>(val, rgq[i], cgc + cl, cgc + 64 + cl, cgc + 128 + cl);
            mask16(val, g_go[i] >= 0);
            const int v = tid + i * 256;
            *reinterpret_cast<V16*>(sG + (v / VPX) * k.psg + g_part * 16) = val;
        }
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            if (h_rc[i] < 0) continue;
            V16 val = rhv[i];
            if (io.h.mode == STL_SRC_BN) val = xform_bn<T, TY>(val, chc + cl, chc + 64 + cl, relu_lo);
            else val = xform_cvt<T, TY>(val);
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

    int t = bsplit;
    bool have = t < k.npt;
    WSTAMP(1);
    // BatchNorm constants (wave 3: the 64 channels of g, wave 2: those of h): statistics loads go out
    // ahead of the first tile's loads, the arithmetic runs while those are in flight
    SrcRaw raw;
    const bool cw = wave >= 2, cg = wave == 3;
    const bool cok = cw && (cg ? co0 + lane < p.Co : ci0 + lane < p.Ci);
    if (cok) {
        if (cg) src_raw_load(io.g, co0 + lane, p.Co, raw);
        else src_raw_load(io.h, ci0 + lane, p.Ci, raw);
    }
    if (have) setup(t);
    issue(have);
    WSTAMP(2);
    if (cw) {
        float a = 0.f, b = 0.f, cc = 0.f;
        if (cok) {
            if (cg) src_raw_finish(io.g, raw, a, b, cc);
            else src_raw_finish(io.h, raw, a, b, cc);
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
        const int tn = t + p.nsplit;
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
        float* slab = io.partial + (size_t)bsplit * p.Co * TAPS * p.Ci;
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

template <typename T, typename TY, int KS, int NVH, bool GQ, int TPX>
int launch64(const WgK& k, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad64_kernel<T, KS, NVH, GQ, TPX, TY>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    STL_LAUNCH((wgrad64_kernel<T, KS, NVH, GQ, TPX, TY>), grid, dim3(256), lds, st, k);
    static char nbuf[160];
    static const char* nm = stl_kname<T>(nbuf, "wgrad64_kernel", {KS, NVH, GQ, TPX, stl_code<TY>()});
    stl_note_kernel(nm, true);
    STL_LAUNCH_CHECK("conv_wgrad64");
    return 0;
}

#if STL_DT != 0
template <typename TY>
int dispatch64(const WgK& k, dim3 grid, size_t lds, hipStream_t st) {   // 1x1 layers, 128-pixel tiles
    const int nvh = ceil_div(k.HP * 8, 256);
    const bool gq = k.p.g.mode == STL_SRC_BNBWD;
    if (k.p.TH * k.p.TW <= 128 && nvh <= 6) return gq ? launch64<__bf16, TY, 1, 6, true, 128>(k, grid, lds, st) : launch64<__bf16, TY, 1, 6, false, 128>(k, grid, lds, st);
    return stl_set_error("wgrad64: halo of %d pixels is too large for a %d-pixel tile", k.HP, k.p.TH * k.p.TW);
}
#endif

// launch grid + block-order fields of k (wg_block): XCD-aware order when there is more than one channel chunk
// (STL_WGRAD_XCD=0: plain 3-D grid)
dim3 wg_grid(WgK& k, int units, int gy, int gz) {
    const int xcd_env = getenv("STL_WGRAD_XCD") ? atoi(getenv("STL_WGRAD_XCD")) : 1;   // read per call (tests cover both orders)
    k.units = units, k.gy = gy, k.gz = gz;
    k.xmap = xcd_env && gy * gz > 1;
    if (!k.xmap) return dim3(units, gy, gz);
    return dim3(8 * ceil_div(units, 8) * gy * gz, 1, 1);
}

// channel tile (64 or 32) of the kernel variant stl_conv_wgrad picks for this problem
// 3x3 stride-1 layers with at least 64 channels on both sides (bf16): 64 x 64 channels per 16-wave block of wgrad_kernel
// (round 5; STL_WGRAD_C64=0: the 32-channel blocks -- A/B)
static bool wgrad_c64_3x3(const stl_wgrad& p) {
    const char* e = getenv("STL_WGRAD_C64");
    return !(e && atoi(e) == 0) && p.dtype == STL_BF16 && p.ks == 3 && p.stride == 1 && p.Co >= 64 && p.Ci >= 64;
}

int wgrad_chunk(const stl_wgrad& p) {
    if (wgrad_c64_3x3(p)) return 64;
    // The wide variant serves the 1x1 convolutions only: there it halves the re-reads of the 113 MB layer1 tensors and its
    // slabs are small (21.8 vs 22.0 ms per step in round 1; without it 15.48 vs 14.5 in round 3).  For 3x3 layers it loses as
    // much again to the 4x larger split-K slabs and to its 256 VGPRs (15.33 -> 16.42-17.24 ms per step): removed in round 4.
    return (p.dtype == STL_BF16 && p.ks == 1 && p.stride == 1 && p.Co >= 64 && p.Ci >= 64) ? 64 : 32;
}

template <typename T, typename TY, int KS>
int dispatch(const WgK& k, dim3 grid, size_t lds, hipStream_t st) {
    {   // both 32-channel kernels address with 24-bit multiplies, scalar-unit division and 32-bit byte offsets
        const int64_t gpx = (int64_t)k.p.B * k.p.Ho * k.p.Wo, hpx = (int64_t)k.p.B * k.p.Hi * k.p.Wi;
        const bool small = gpx * k.p.Co * (int64_t)sizeof(T) < ((int64_t)1 << 32) && hpx * k.p.Ci * (int64_t)sizeof(T) < ((int64_t)1 << 32) &&
                           (int64_t)k.p.Wo * k.p.Co * (int64_t)sizeof(T) < (1 << 23) &&
                           (int64_t)(k.PI > k.p.Hi ? k.PI - k.p.Hi : k.p.Hi - k.PI) * k.p.Wi * k.p.Ci * (int64_t)sizeof(T) < (1 << 23) &&
                           k.PI < (1 << 12) && k.tiles_c < (1 << 12) && (int64_t)k.npt * k.p.TH * k.p.stride < (1 << 20);
        STL_CHECK(small, "wgrad: tensors of 4 GB or more, rows beyond 8 MB or more than 4095 rows / tile columns are not addressable");
    }
    const int vpx = 32 / ET<T>::KV;
    const int nvh = ceil_div(k.HP * vpx, 256);
    const bool gq = k.p.g.mode == STL_SRC_BNBWD;
    STL_CHECK(k.p.TH * k.p.TW <= 128, "wgrad: tiles of more than 128 pixels are not supported");
    if constexpr (KS == 3 && sizeof(T) == 2) {
        if (k.p.ks == 3 && wgrad_c64_3x3(k.p)) {   // 64 x 64 channels per 16-wave block
            const int nvh16 = ceil_div(k.HP * 8, 1024);
            if (nvh16 <= 2) return gq ? launch<T, TY, KS, 2, true, 128, 16, 64>(k, grid, lds, st) : launch<T, TY, KS, 2, false, 128, 16, 64>(k, grid, lds, st);
            if (nvh16 <= 3) return gq ? launch<T, TY, KS, 3, true, 128, 16, 64>(k, grid, lds, st) : launch<T, TY, KS, 3, false, 128, 16, 64>(k, grid, lds, st);
            return stl_set_error("wgrad: halo of %d pixels is too large for the 64-channel 3x3 kernel; shrink the tile", k.HP);
        }
    }
    if constexpr (KS == 3) {   // 8 waves: quadrants x two tap groups, half the staging work per thread (bf16 and fp32)
        const int nvh8 = ceil_div(k.HP * vpx, 512);
        if (nvh8 <= 2) return gq ? launch<T, TY, KS, 2, true, 128, 8>(k, grid, lds, st) : launch<T, TY, KS, 2, false, 128, 8>(k, grid, lds, st);
        if (nvh8 <= 3) return gq ? launch<T, TY, KS, 3, true, 128, 8>(k, grid, lds, st) : launch<T, TY, KS, 3, false, 128, 8>(k, grid, lds, st);
        if (nvh8 <= 5) return gq ? launch<T, TY, KS, 5, true, 128, 8>(k, grid, lds, st) : launch<T, TY, KS, 5, false, 128, 8>(k, grid, lds, st);
        if (nvh8 <= 9) return gq ? launch<T, TY, KS, 9, true, 128, 8>(k, grid, lds, st) : launch<T, TY, KS, 9, false, 128, 8>(k, grid, lds, st);
        return stl_set_error("wgrad: halo of %d pixels is too large for the 3x3 kernel; shrink the tile", k.HP);
    }
    if (nvh <= 3) return gq ? launch<T, TY, KS, 3, true>(k, grid, lds, st) : launch<T, TY, KS, 3, false>(k, grid, lds, st);
    if (nvh <= 6) return gq ? launch<T, TY, KS, 6, true>(k, grid, lds, st) : launch<T, TY, KS, 6, false>(k, grid, lds, st);
    if (nvh <= 9) return gq ? launch<T, TY, KS, 9, true>(k, grid, lds, st) : launch<T, TY, KS, 9, false>(k, grid, lds, st);
    if (nvh <= 18) return gq ? launch<T, TY, KS, 18, true>(k, grid, lds, st) : launch<T, TY, KS, 18, false>(k, grid, lds, st);
    return stl_set_error("wgrad: halo of %d pixels needs %d staging vectors per thread (max 18); shrink the tile", k.HP, nvh);
}

}  // namespace

#if STL_DT != 0
int stl_wgrad_backend_bf16(bool wide, const WgK& k, dim3 grid, size_t lds, hipStream_t st) {
    if (k.p.ydtype == STL_F16) {   // mixed mode: h and g.y are f16 forward tensors, dt and the MFMA operands bf16
        if (wide) return dispatch64<f16>(k, grid, lds, st);
        return k.p.ks == 3 ? dispatch<__bf16, f16, 3>(k, grid, lds, st) : dispatch<__bf16, f16, 1>(k, grid, lds, st);
    }
    if (wide) return dispatch64<__bf16>(k, grid, lds, st);
    return k.p.ks == 3 ? dispatch<__bf16, __bf16, 3>(k, grid, lds, st) : dispatch<__bf16, __bf16, 1>(k, grid, lds, st);
}
#endif
#if STL_DT != 1
int stl_wgrad_backend_f32(bool wide, const WgK& k, dim3 grid, size_t lds, hipStream_t st) {
    if (wide) return stl_set_error("wgrad: the 64 x 64-channel variant is bf16 only");
    return k.p.ks == 3 ? dispatch<float, float, 3>(k, grid, lds, st) : dispatch<float, float, 1>(k, grid, lds, st);
}
#endif

#if STL_DT != 0   // the C ABI entry points live in the bf16 (or the only) unit
#ifdef STL_STAMPS
extern "C" int stl_debug_wgrad_stamps2(long long* host64) {
    return hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_wstamps2), 64 * sizeof(long long)) == hipSuccess ? 0 : stl_set_error("stamps: copy failed");
}
extern "C" int stl_debug_wgrad_stamps(long long* host16) {
    return hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_wstamps), 16 * sizeof(long long)) == hipSuccess ? 0 : stl_set_error("stamps: copy failed");
}
#endif

extern "C" int stl_wgrad_chunk(const stl_wgrad* pp) { return wgrad_chunk(*pp); }

static int wgrad_run(const stl_wgrad* const* ps, int ng, void* stream);

extern "C" int stl_conv_wgrad(const stl_wgrad* pp, void* stream) { return wgrad_run(&pp, 1, stream); }

// Several weight gradients of identical geometry (same layer shape, tile, split count and source modes -- e.g. the
// eight 3x3 convolutions of one branch of an exchange module) in ONE launch: grid.x = n * nsplit, block
// (member, split) works on member's tensors.  One launch instead of n keeps the hardware queues free for the
// data-gradient chain, and with the block budget shared by the members every block walks n times as many pixel
// tiles (longer steady state, n times fewer split-K slabs to write and to reduce).
extern "C" int stl_conv_wgrad_group(const stl_wgrad_group* grp, void* stream) {
    STL_CHECK(grp && grp->n >= 1 && grp->n <= WG_MAXG, "wgrad_group: 1..%d members", WG_MAXG);
    const stl_wgrad& a = *grp->p[0];
    for (int i = 1; i < grp->n; ++i) {
        const stl_wgrad& b = *grp->p[i];
        STL_CHECK(b.dtype == a.dtype && b.B == a.B && b.Hi == a.Hi && b.Wi == a.Wi && b.Ci == a.Ci && b.Ho == a.Ho && b.Wo == a.Wo &&
                      b.Co == a.Co && b.ks == a.ks && b.stride == a.stride && b.TH == a.TH && b.TW == a.TW && b.nsplit == a.nsplit &&
                      b.g.mode == a.g.mode,
                  "wgrad_group: member %d differs from member 0 in geometry, tile, nsplit or gradient source mode", i);
    }
    return wgrad_run(grp->p, grp->n, stream);
}

static int wgrad_run(const stl_wgrad* const* ps, int ng, void* stream) {
    const stl_wgrad& p = *ps[0];
  for (int i_ = 0; i_ < ng; ++i_) {
    const stl_wgrad& p = *ps[i_];
    STL_CHECK(p.dtype == STL_F32 || p.dtype == STL_BF16, "wgrad: bad dtype (gradients are fp32 or bf16)");
    STL_CHECK(p.ydtype == 0 || p.ydtype == p.dtype || (p.dtype == STL_BF16 && p.ydtype == STL_F16), "wgrad: ydtype %d does not go with dtype %d", p.ydtype, p.dtype);
    STL_CHECK(p.ydtype == ps[0]->ydtype, "wgrad_group: members differ in ydtype");
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
  }
    const int pad = p.ks == 3 ? 1 : 0;
    WgK k;
    k.p = p;
    k.ng = ng;
    for (int i = 0; i < ng; ++i) k.io[i].h = ps[i]->h, k.io[i].g = ps[i]->g, k.io[i].partial = ps[i]->partial;
    k.dbg = getenv("STL_CONV_STAMPS") ? 1 : 0;
    k.taps = p.ks * p.ks;
    k.pad = pad;
    k.PI = p.stride * (p.Ho + 1);
    k.HR = (p.TH - 1) * p.stride + p.ks;
    k.HC = (p.TW - 1) * p.stride + p.ks;
    k.HP = k.HR * k.HC;
    k.tiles_c = ceil_div(p.Wo, p.TW);
    k.npt = ceil_div(p.B * (p.Ho + 1), p.TH) * k.tiles_c;
    k.r_TW = 1.0f / p.TW, k.r_HC = 1.0f / k.HC, k.r_tc = 1.0f / k.tiles_c, k.r_vp = 1.0f / (p.Ho + 1), k.r_PI = 1.0f / k.PI;
    const unsigned long long two32 = 1ull << 32;
    k.m_tc = (two32 + k.tiles_c - 1) / k.tiles_c, k.m_vp = (two32 + p.Ho) / (p.Ho + 1), k.m_PI = (two32 + k.PI - 1) / k.PI;
    STL_CHECK((int64_t)k.npt < (1 << 21) && (int64_t)p.B * k.PI < (1 << 21), "wgrad: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    if (wgrad_c64_3x3(p)) {   // wgrad_kernel with 64-channel blocks: pixel stride 144 B, constants [3][64] + [2][64]
        k.psg = k.psh = 64 * 2 + 32;
        k.off_cg = 0, k.off_ch = 3 * 64 * 4, k.off_g = 2048;
        k.off_h = k.off_g + 128 * k.psg;
        const size_t ldsw = (size_t)k.off_h + (size_t)k.HP * k.psh;
        STL_CHECK(ldsw <= 160 * 1024, "wgrad: tile needs %zu B of LDS (>160 KiB)", ldsw);
        STL_CHECK(p.nsplit <= k.npt || p.nsplit == 1, "wgrad: nsplit %d > tiles %d", p.nsplit, k.npt);
        dim3 gridw = wg_grid(k, p.nsplit * ng, ceil_div(p.Co, 64), ceil_div(p.Ci, 64));
        return stl_wgrad_backend_bf16(false, k, gridw, ldsw, st);
    }
    if (wgrad_chunk(p) == 64) {
        k.psg = k.psh = 160;
        k.off_cg = 0, k.off_ch = 3 * 64 * 4, k.off_g = 2048;  // consts: g [3][64] at 0, h [2][64] at 768
        k.off_h = k.off_g + 128 * k.psg;
        const size_t lds64 = (size_t)k.off_h + (size_t)k.HP * k.psh;
        STL_CHECK(lds64 <= 160 * 1024, "wgrad64: tile needs %zu B of LDS (>160 KiB)", lds64);
        STL_CHECK(p.nsplit <= k.npt || p.nsplit == 1, "wgrad: nsplit %d > tiles %d", p.nsplit, k.npt);
        dim3 grid64 = wg_grid(k, p.nsplit * ng, ceil_div(p.Co, 64), ceil_div(p.Ci, 64));
        return stl_wgrad_backend_bf16(true, k, grid64, lds64, st);
    }
    const int esz = p.dtype == STL_BF16 ? 2 : 4;
    k.psg = k.psh = 32 * esz + (esz == 2 ? 32 : 16);
    k.off_cg = 0;
    k.off_ch = 3 * 32 * 4;
    k.off_g = 1024;  // consts: g [3][32] floats at 0, h [2][32] floats at 384 -> 640 B used
    int szG = 128 * k.psg;
    int szH = k.HP * k.psh;
    k.off_h = k.off_g + szG;
    size_t lds = (size_t)k.off_h + szH;
    STL_CHECK(lds <= 160 * 1024, "wgrad: tile needs %zu B of LDS (>160 KiB)", lds);
    STL_CHECK(p.nsplit <= k.npt || p.nsplit == 1, "wgrad: nsplit %d > tiles %d", p.nsplit, k.npt);
    dim3 grid = wg_grid(k, p.nsplit * ng, ceil_div(p.Co, 32), ceil_div(p.Ci, 32));
    return p.dtype == STL_BF16 ? stl_wgrad_backend_bf16(false, k, grid, lds, st) : stl_wgrad_backend_f32(false, k, grid, lds, st);
}
#endif   // STL_DT != 0
