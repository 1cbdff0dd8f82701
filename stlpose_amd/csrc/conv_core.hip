// Implicit-GEMM convolution for gfx950: NHWC, im2col-free, MFMA 16x16 tiles, LDS-staged input
// halo tile and filter slice, "normalise on load" prologue and a fused epilogue.
// One kernel serves the forward conv and (with transposed/flipped filters) the data gradient.
//
//   GEMM view:  M = output pixels (tile of <=128, 8 MFMA row tiles), N = BN output channels,
//               K = taps x input channels, walked as 64-byte channel chunks (32 bf16 / 16 f32).
//   Block = 256 threads = 4 waves; wave w owns row tiles {2w, 2w+1} x all BN/16 column tiles.
//   Pixel tiles are taken from "virtual rows": the batch is stacked along y with one zero row
//   between images, so small feature maps (12x9, 8x6) still fill 128-pixel tiles.
//   Pipeline: a stage = (tile, channel chunk).  While the MFMAs of stage s run out of LDS, the
//   global loads of stage s+1 are already in flight into registers (issue-early / write-late);
//   all per-vector index arithmetic is hoisted out of the stage loop.
#include <stdlib.h>
#include "common.cuh"

namespace {

constexpr int PSA = 80;  // LDS bytes per halo pixel: 64 B of channels + 16 B pad (bank spread)

struct ConvK {
    stl_conv p;
    int tiles_c, npt;  // pixel tiles: columns, total
    int HR, HC, HP;    // halo rows, cols, pixels
    int PI, pad, seff; // source virtual pitch, padding, effective stride of the core
    int nchunks, wres;
    int cipad, copad;
    int off_cs, off_cm, off_main, off_b;  // LDS byte offsets
    int taps;
};

template <typename T>
__device__ __forceinline__ void load8(const void* base, size_t elem, float* f) {
    if constexpr (sizeof(T) == 2) {
        V16 v = ldg16((const char*)base + elem * 2);
        unpack<__bf16>(v, f);
    } else {
        V16 a = ldg16((const char*)base + elem * 4), b = ldg16((const char*)base + elem * 4 + 16);
        unpack<float>(a, f);
        unpack<float>(b, f + 4);
    }
}
template <typename T>
__device__ __forceinline__ void store8(void* base, size_t elem, const float* f) {
    if constexpr (sizeof(T) == 2) {
        stg16((char*)base + elem * 2, pack<__bf16>(f));
    } else {
        stg16((char*)base + elem * 4, pack<float>(f));
        stg16((char*)base + elem * 4 + 16, pack<float>(f + 4));
    }
}

// NVA: A (input halo) vectors per thread per stage; Q: source is BNBWD (second tensor on load)
template <typename T, int KS, int BN, int NVA, bool Q>
__global__ __launch_bounds__(256) void conv_core_kernel(const ConvK k) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KV = ET<T>::KV, CK = ET<T>::CK, TAPS = KS * KS, NT = BN / 16;
    constexpr int ROWB = TAPS * 64 + 16;
    constexpr int OSTR = BN + 4;
    constexpr int VPP = BN / 8;
    constexpr int NVB = (BN * TAPS * 4 + 255) / 256;
    constexpr int NSLOT = 128 * VPP / 256;
    const stl_conv& p = k.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * BN;

    float* cs = reinterpret_cast<float*>(smem + k.off_cs);  // [3][cipad] source transform
    float* cm = reinterpret_cast<float*>(smem + k.off_cm);  // [4][BN]    mask BN: a, b, mean, rstd
    char* sA = smem + k.off_main;
    char* sB = smem + k.off_b;
    float* sOut = reinterpret_cast<float*>(smem + k.off_main);

    // ---- per-channel constants
    for (int c = tid; c < k.cipad; c += 256) {
        float a = 0.f, b = 0.f, cc = 0.f;
        if (c < p.Ci) src_consts(p.src, c, p.Ci, a, b, cc);
        cs[c] = a, cs[k.cipad + c] = b, cs[2 * k.cipad + c] = cc;
    }
    if (p.mask_y) {
        for (int c = tid; c < BN; c += 256) {
            float a = 0.f, b = 0.f, mu = 0.f, rs = 0.f;
            if (n0 + c < p.Co) {
                bn_mean_rstd(p.mask_bn, n0 + c, p.Co, mu, rs);
                a = p.mask_bn.gamma[n0 + c] * rs;
                b = p.mask_bn.beta[n0 + c] - mu * a;
            }
            cm[c] = a, cm[BN + c] = b, cm[2 * BN + c] = mu, cm[3 * BN + c] = rs;
        }
    }

    // ---- loop-invariant per-thread descriptors
    int a_rc[NVA];  // (halo row << 16) | halo col, -1 when this slot is unused
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int v = tid + i * 256;
        if (v < k.HP * 4) {
            const int hp = v >> 2, hr = hp / k.HC;
            a_rc[i] = (hr << 16) | (hp - hr * k.HC);
        } else {
            a_rc[i] = -1;
        }
    }
    const int a_part = tid & 3;  // (tid + i*256) & 3
    int b_g[NVB], b_l[NVB];      // weight element offset (without chunk) / LDS byte offset
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
        const int v = tid + i * 256;
        b_g[i] = -1, b_l[i] = 0;
        if (v < BN * TAPS * 4) {
            const int n = v / (TAPS * 4), r = v - n * (TAPS * 4), tap = r >> 2, part = r & 3;
            b_l[i] = n * ROWB + tap * 64 + part * 16;
            if (n0 + n < p.Co) b_g[i] = ((n0 + n) * TAPS + tap) * p.Ci + part * KV;
        }
    }
    const int tilepx = p.TH * p.TW;
    int aoff[2];  // lane's A row offsets for its two row tiles
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        int m = (wave * 2 + mi) * 16 + r16;
        if (m >= tilepx) m = 0;
        const int ty = m / p.TW, tx = m - ty * p.TW;
        aoff[mi] = ((ty * k.seff) * k.HC + tx * k.seff) * PSA + g * 16;
    }
    const int boff = r16 * ROWB + g * 16;
    const int cg = tid % VPP;
    const int co = n0 + cg * 8;
    int e_yx[NSLOT];  // epilogue pixel of slot s: (ty << 16) | tx, -1 unused
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int m = (tid + s * 256) / VPP;
        if (m < tilepx) {
            const int ty = m / p.TW;
            e_yx[s] = (ty << 16) | (m - ty * p.TW);
        } else {
            e_yx[s] = -1;
        }
    }

    // ---- staging registers + helpers
    V16 ra[NVA], rq[Q ? NVA : 1], rb[NVB];
    int a_go[NVA];  // element offset of this stage's tile pixel (+part), -1 = zero fill

    auto tile_setup = [&](int t, int* go) {
        const int tr = t / k.tiles_c, tc = t - tr * k.tiles_c;
        const int vrs = tr * p.TH * k.seff, cb = tc * p.TW * k.seff - k.pad;
        const int b0 = vrs / k.PI, y0 = vrs - b0 * k.PI - k.pad;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            go[i] = -1;
            if (a_rc[i] >= 0) {
                int iy = y0 + (a_rc[i] >> 16), ix = cb + (a_rc[i] & 0xffff), b = b0;
                if (iy >= 0 && ix >= 0) {
                    while (iy >= k.PI) iy -= k.PI, ++b;
                    bool ok;
                    if (p.stuff) {
                        ok = (b < p.B) && !((iy | ix) & 1) && ((iy >> 1) < p.Hi) && ((ix >> 1) < p.Wi);
                        iy >>= 1, ix >>= 1;
                    } else {
                        ok = (b < p.B) && (iy < p.Hi) && (ix < p.Wi);
                    }
                    if (ok) go[i] = ((b * p.Hi + iy) * p.Wi + ix) * p.Ci + a_part * KV;
                }
            }
        }
    };
    auto issue = [&](const int* go, int k0) {
        const bool chok = (k0 + a_part * KV) < p.Ci;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            ra[i] = zero16();
            if (Q) rq[i] = zero16();
            if (go[i] >= 0 && chok) {
                ra[i] = ldg16((const char*)p.src.x + (size_t)(go[i] + k0) * sizeof(T));
                if (Q) rq[i] = ldg16((const char*)p.src.y + (size_t)(go[i] + k0) * sizeof(T));
            }
        }
        if (!k.wres) {
#pragma unroll
            for (int i = 0; i < NVB; ++i) {
                rb[i] = zero16();
                if (b_g[i] >= 0 && (k0 + ((tid + i * 256) & 3) * KV) < p.Ci)
                    rb[i] = ldg16((const char*)p.w + (size_t)(b_g[i] + k0) * sizeof(T));
            }
        }
    };
    auto write_lds = [&](const int* go, int k0) {
        const int ch = k0 + a_part * KV;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            if (a_rc[i] < 0) continue;
            V16 val = ra[i];
            if (p.src.mode != STL_SRC_PLAIN && go[i] >= 0 && ch < p.Ci) {
                float f[KV];
                unpack<T>(val, f);
                if (!Q) {
#pragma unroll
                    for (int j = 0; j < KV; ++j) {
                        float u = cs[ch + j] * f[j] + cs[k.cipad + ch + j];
                        f[j] = p.src.relu ? fmaxf(u, 0.f) : u;
                    }
                } else {
                    float q[KV];
                    unpack<T>(rq[i], q);
#pragma unroll
                    for (int j = 0; j < KV; ++j)
                        f[j] = cs[ch + j] * f[j] + cs[k.cipad + ch + j] * q[j] + cs[2 * k.cipad + ch + j];
                }
                val = pack<T>(f);
            }
            const int v = tid + i * 256;
            *reinterpret_cast<V16*>(sA + (v >> 2) * PSA + (v & 3) * 16) = val;
        }
        if (!k.wres) {
#pragma unroll
            for (int i = 0; i < NVB; ++i)
                if (tid + i * 256 < BN * TAPS * 4) *reinterpret_cast<V16*>(sB + b_l[i]) = rb[i];
        }
    };

    if (k.wres) {  // whole K fits one chunk: filters stay resident in LDS for all tiles
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            if (tid + i * 256 < BN * TAPS * 4) {
                V16 val = zero16();
                if (b_g[i] >= 0 && (((tid + i * 256) & 3) * KV) < p.Ci) val = ldg16((const char*)p.w + (size_t)b_g[i] * sizeof(T));
                *reinterpret_cast<V16*>(sB + b_l[i]) = val;
            }
        }
    }

    // statistics accumulators (this thread always handles channel group cg)
    float acc_s0[8], acc_s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc_s0[j] = acc_s1[j] = 0.f;

    const int xcd = blockIdx.x & 7, lx = blockIdx.x >> 3, nx = gridDim.x >> 3;
    const int T8 = (k.npt + 7) >> 3;
    const int vpitch = p.Ho + 1;

    int it = lx;
    int t = xcd * T8 + it;
    bool have = (it < T8) && (t < k.npt);
    if (have) {
        tile_setup(t, a_go);
        issue(a_go, 0);
    }
    __syncthreads();  // constants + resident filters visible

    while (have) {
        const int tr = t / k.tiles_c, tc = t - tr * k.tiles_c;
        const int vr0 = tr * p.TH, c0 = tc * p.TW;
        const int itn = it + nx, tn = xcd * T8 + itn;
        const bool have_next = (itn < T8) && (tn < k.npt);

        f32x4 acc[2][NT];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mi][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ch0 = 0; ch0 < k.nchunks; ++ch0) {
            write_lds(a_go, ch0 * CK);
            __syncthreads();
            // ---- next stage's global loads go out now and land during the MFMAs below
            if (ch0 + 1 < k.nchunks) {
                issue(a_go, (ch0 + 1) * CK);
            } else if (have_next) {
                tile_setup(tn, a_go);
                issue(a_go, 0);
            }
            // ---- MFMA over the taps of this chunk
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int toff = ((tap / KS) * k.HC + (tap % KS)) * PSA;
                V16 a[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const V16*>(sA + aoff[mi] + toff);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const V16 b = *reinterpret_cast<const V16*>(sB + boff + nt * 16 * ROWB + tap * 64);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) mma16<T>(acc[mi][nt], a[mi], b);
                }
            }
            __syncthreads();  // everyone is done with sA/sB of this stage
        }
        // ---- epilogue: accumulators -> LDS [pixel][channel] -> fused elementwise -> global
        // (when filters are resident the output staging must not overwrite them)
        float* so = k.wres ? reinterpret_cast<float*>(smem + k.off_b + BN * ROWB) : sOut;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    so[((wave * 2 + mi) * 16 + 4 * g + r) * OSTR + nt * 16 + r16] = acc[mi][nt][r];
        __syncthreads();
        if (co < p.Co) {
            const int eb0 = vr0 / vpitch, ey0 = vr0 - eb0 * vpitch;
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                if (e_yx[s] < 0) continue;
                const int m = (tid + s * 256) / VPP;
                int oy = ey0 + (e_yx[s] >> 16), b = eb0;
                const int c = c0 + (e_yx[s] & 0xffff);
                while (oy >= vpitch) oy -= vpitch, ++b;
                if (b >= p.B || oy >= p.Ho || c >= p.Wo) continue;
                float f[8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(so + m * OSTR + cg * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(so + m * OSTR + cg * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = lo[j], f[4 + j] = hi[j];
                const size_t off = (((size_t)b * p.Ho + oy) * p.Wo + c) * p.Co + co;
                if (p.bias) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] += p.bias[co + j];
                }
                if (p.addend) {
                    float ad[8];
                    load8<T>(p.addend, off, ad);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] += ad[j];
                }
                float yh[8];
                if (p.mask_y) {
                    float my[8];
                    load8<T>(p.mask_y, off, my);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int cl = cg * 8 + j;
                        if (p.mask_bn.relu && !(cm[cl] * my[j] + cm[BN + cl] > 0.f)) f[j] = 0.f;
                        yh[j] = (my[j] - cm[2 * BN + cl]) * cm[3 * BN + cl];
                    }
                }
                if (p.out_relu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], 0.f);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = round_to<T>(f[j]);
                if (p.out_stats) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc_s0[j] += f[j], acc_s1[j] += f[j] * f[j];
                } else if (p.red) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc_s0[j] += f[j], acc_s1[j] += f[j] * yh[j];
                }
                store8<T>(p.out, off, f);
            }
        }
        __syncthreads();  // output staging consumed before the next tile's write_lds
        it = itn, t = tn, have = have_next;
    }
    // ---- flush statistics: deterministic in-block tree, then one fp64 atomic per channel
    double* dst = p.out_stats ? p.out_stats : p.red;
    if (dst) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem + k.off_main);  // [256][16]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid * 16 + j] = acc_s0[j], red[tid * 16 + 8 + j] = acc_s1[j];
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, cl = tid - which * BN, cgr = cl >> 3, j = cl & 7;
            float s = 0.f;
            for (int q = cgr; q < 256; q += VPP) s += red[q * 16 + which * 8 + j];
            if (n0 + cl < p.Co)
                atomicAdd(dst + (size_t)(blockIdx.x & (STL_NSHARD - 1)) * 2 * p.Co + which * p.Co + n0 + cl, (double)s);
        }
    }
}

template <typename T, int KS, int BN, int NVA, bool Q>
int launch(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_core_kernel<T, KS, BN, NVA, Q>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_core_kernel<T, KS, BN, NVA, Q>), grid, dim3(256), lds, st, k);
    STL_LAUNCH_CHECK("conv_core");
    return 0;
}

template <typename T, int KS, int BN>
int launch_nq(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
    const int nva = ceil_div(k.HP * 4, 256);
    const bool q = k.p.src.mode == STL_SRC_BNBWD;
    if (nva <= 3) return q ? launch<T, KS, BN, 3, true>(k, grid, lds, st) : launch<T, KS, BN, 3, false>(k, grid, lds, st);
    if (nva <= 6) return q ? launch<T, KS, BN, 6, true>(k, grid, lds, st) : launch<T, KS, BN, 6, false>(k, grid, lds, st);
    if (nva <= 9) return q ? launch<T, KS, BN, 9, true>(k, grid, lds, st) : launch<T, KS, BN, 9, false>(k, grid, lds, st);
    return stl_set_error("conv: halo of %d pixels needs %d staging vectors per thread (max 9); shrink the tile", k.HP, nva);
}

}  // namespace

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
    STL_CHECK(p.TH >= 1 && p.TW >= 1 && p.TH * p.TW <= 128, "conv: tile %dx%d exceeds 128 pixels", p.TH, p.TW);
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
    STL_CHECK(p.src.mode >= 0 && p.src.mode <= 2, "conv: bad src mode");
    STL_CHECK(p.src.mode == STL_SRC_PLAIN || p.src.gamma, "conv: BN source without gamma");
    STL_CHECK(p.src.mode != STL_SRC_BN || p.src.beta, "conv: BN source without beta");
    STL_CHECK(p.src.mode != STL_SRC_BN || p.src.stats || (p.src.rmean && p.src.rvar), "conv: BN source without statistics");
    STL_CHECK(p.src.mode != STL_SRC_BNBWD || (p.src.y && p.src.stats && p.src.rstats), "conv: BNBWD source incomplete");
    STL_CHECK(!(p.out_stats && p.red), "conv: out_stats and red are exclusive");
    STL_CHECK(!p.red || p.mask_y, "conv: red needs mask_y");
    STL_CHECK(!p.mask_y || (p.mask_bn.gamma && p.mask_bn.beta && (p.mask_bn.stats || (p.mask_bn.rmean && p.mask_bn.rvar))), "conv: mask BN incomplete");

    ConvK k;
    k.p = p;
    k.taps = p.ks * p.ks;
    k.seff = p.stride;
    k.pad = pad;
    k.PI = p.stuff ? (p.Ho + 1) : p.stride * (p.Ho + 1);
    k.HR = (p.TH - 1) * k.seff + p.ks;
    k.HC = (p.TW - 1) * k.seff + p.ks;
    k.HP = k.HR * k.HC;
    const int vrows = p.B * (p.Ho + 1);
    const int tiles_r = ceil_div(vrows, p.TH);
    k.tiles_c = ceil_div(p.Wo, p.TW);
    k.npt = tiles_r * k.tiles_c;
    k.nchunks = ceil_div(p.Ci, ck);
    k.wres = k.nchunks == 1;
    k.cipad = k.nchunks * ck;
    const int BN = (p.Co <= 32) ? 32 : 64;
    k.copad = BN;
    const int rowb = k.taps * 64 + 16;
    const int szA = (k.HP * PSA + 15) & ~15, szB = BN * rowb;
    const int szOut = 128 * (BN + 4) * 4, szRed = 256 * 16 * 4;
    int szMain = szA + szB + (k.wres ? szOut : 0);  // resident filters: output staging gets its own region
    if (szOut > szMain) szMain = szOut;
    if (szRed > szMain) szMain = szRed;
    k.off_cs = 0;
    k.off_cm = 3 * k.cipad * 4;
    k.off_main = (k.off_cm + 4 * BN * 4 + 15) & ~15;
    k.off_b = k.off_main + szA;
    const size_t lds = (size_t)k.off_main + szMain;
    STL_CHECK(lds <= 160 * 1024, "conv: tile needs %zu B of LDS (>160 KiB); shrink TH/TW", lds);
    int gx = ceil_div(k.npt, 8) * 8;
    int cap = 1280;
    if (const char* e = getenv("STL_CONV_GRID_CAP")) cap = atoi(e) > 0 ? (atoi(e) + 7) / 8 * 8 : cap;  // tuning knob
    if (getenv("STL_CONV_DEBUG")) {
        int nb = -1;
        fprintf(stderr, "[stl conv] npt=%d grid=(%d,%d) lds=%zu HP=%d nchunks=%d wres=%d\n", k.npt, gx < cap ? gx : cap,
                ceil_div(p.Co, BN), lds, k.HP, k.nchunks, k.wres);
        (void)nb;
    }
    if (gx > cap) gx = cap;
    dim3 grid(gx, ceil_div(p.Co, BN));
    hipStream_t st = (hipStream_t)stream;
#define DISPATCH(T)                                                                                   \
    if (p.ks == 3) {                                                                                  \
        return BN == 32 ? launch_nq<T, 3, 32>(k, grid, lds, st) : launch_nq<T, 3, 64>(k, grid, lds, st); \
    } else {                                                                                          \
        return BN == 32 ? launch_nq<T, 1, 32>(k, grid, lds, st) : launch_nq<T, 1, 64>(k, grid, lds, st); \
    }
    if (p.dtype == STL_BF16) {
        DISPATCH(__bf16)
    } else {
        DISPATCH(float)
    }
#undef DISPATCH
}
