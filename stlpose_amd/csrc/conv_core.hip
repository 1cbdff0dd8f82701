// Implicit-GEMM convolution for gfx950: NHWC, im2col-free, MFMA 16x16 tiles, LDS-staged input
// halo tile and filter slice, "normalise on load" prologue and a fused epilogue.
// One kernel serves the forward conv and (with transposed/flipped filters) the data gradient.
//
//   GEMM view:  M = output pixels (tile of <=128, 8 MFMA row tiles), N = BN output channels,
//               K = taps x input channels, walked as 64-byte channel chunks (32 bf16 / 16 f32).
//   Block = 256 threads = 4 waves; wave w owns row tiles {2w, 2w+1} x all BN/16 column tiles.
//   Pixel tiles are taken from "virtual rows": the batch is stacked along y with one zero row
//   between images, so small feature maps (12x9, 8x6) still fill 128-pixel tiles.
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

template <typename T, int KS, int BN>
__global__ __launch_bounds__(256) void conv_core_kernel(const ConvK k) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KV = ET<T>::KV, CK = ET<T>::CK, TAPS = KS * KS, NT = BN / 16;
    constexpr int ROWB = TAPS * 64 + 16;
    constexpr int OSTR = BN + 4;
    constexpr int VPP = BN / 8;
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

    auto stage_b = [&](int k0) {
        const int nvec = BN * TAPS * 4;
        for (int v = tid; v < nvec; v += 256) {
            const int n = v / (TAPS * 4), r = v - n * (TAPS * 4), tap = r >> 2, part = r & 3;
            const int co = n0 + n, ch = k0 + part * KV;
            V16 val = zero16();
            if (co < p.Co && ch < p.Ci)
                val = ldg16((const char*)p.w + ((size_t)(co * TAPS + tap) * p.Ci + ch) * sizeof(T));
            *reinterpret_cast<V16*>(sB + n * ROWB + tap * 64 + part * 16) = val;
        }
    };
    if (k.wres) stage_b(0);

    // statistics accumulators (this thread always handles channel group tid % VPP)
    float acc_s0[8], acc_s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc_s0[j] = acc_s1[j] = 0.f;

    const int xcd = blockIdx.x & 7, lx = blockIdx.x >> 3, nx = gridDim.x >> 3;
    const int T8 = (k.npt + 7) >> 3;
    const int tilepx = p.TH * p.TW;
    const int vpitch = p.Ho + 1;

    for (int it = lx; it < T8; it += nx) {
        const int t = xcd * T8 + it;
        if (t >= k.npt) break;
        const int tr = t / k.tiles_c, tc = t - tr * k.tiles_c;
        const int vr0 = tr * p.TH, c0 = tc * p.TW;

        // lane's A row offsets for its two row tiles
        int aoff[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            int m = (wave * 2 + mi) * 16 + r16;
            if (m >= tilepx) m = 0;
            const int ty = m / p.TW, tx = m - ty * p.TW;
            aoff[mi] = ((ty * k.seff) * k.HC + tx * k.seff) * PSA + g * 16;
        }
        const int boff = r16 * ROWB + g * 16;

        f32x4 acc[2][NT];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mi][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ch0 = 0; ch0 < k.nchunks; ++ch0) {
            const int k0 = ch0 * CK;
            __syncthreads();  // previous users of sA/sB/sOut are done
            // ---- stage the input halo tile for channels [k0, k0+CK), transformed on load
            for (int v = tid; v < k.HP * 4; v += 256) {
                const int hp = v >> 2, part = v & 3;
                const int hr = hp / k.HC, hc = hp - hr * k.HC;
                const int ch = k0 + part * KV;
                const int vri = vr0 * k.seff - k.pad + hr;
                int ix = c0 * k.seff - k.pad + hc;
                bool ok = (vri >= 0) && (ix >= 0) && (ch < p.Ci);
                int b = 0, iy = 0;
                if (ok) {
                    b = vri / k.PI;
                    iy = vri - b * k.PI;
                    if (p.stuff) {
                        ok = (b < p.B) && !((iy | ix) & 1) && ((iy >> 1) < p.Hi) && ((ix >> 1) < p.Wi);
                        iy >>= 1;
                        ix >>= 1;
                    } else {
                        ok = (b < p.B) && (iy < p.Hi) && (ix < p.Wi);
                    }
                }
                V16 val = zero16();
                if (ok) {
                    const size_t off = (((size_t)b * p.Hi + iy) * p.Wi + ix) * p.Ci + ch;
                    val = ldg16((const char*)p.src.x + off * sizeof(T));
                    if (p.src.mode != STL_SRC_PLAIN) {
                        float f[KV];
                        unpack<T>(val, f);
                        if (p.src.mode == STL_SRC_BN) {
#pragma unroll
                            for (int j = 0; j < KV; ++j) {
                                float u = cs[ch + j] * f[j] + cs[k.cipad + ch + j];
                                f[j] = p.src.relu ? fmaxf(u, 0.f) : u;
                            }
                        } else {
                            float q[KV];
                            V16 qv = ldg16((const char*)p.src.y + off * sizeof(T));
                            unpack<T>(qv, q);
#pragma unroll
                            for (int j = 0; j < KV; ++j)
                                f[j] = cs[ch + j] * f[j] + cs[k.cipad + ch + j] * q[j] + cs[2 * k.cipad + ch + j];
                        }
                        val = pack<T>(f);
                    }
                }
                *reinterpret_cast<V16*>(sA + hp * PSA + part * 16) = val;
            }
            if (!k.wres) stage_b(k0);
            __syncthreads();
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
        }
        // ---- epilogue: accumulators -> LDS [pixel][channel] -> fused elementwise -> global
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sOut[((wave * 2 + mi) * 16 + 4 * g + r) * OSTR + nt * 16 + r16] = acc[mi][nt][r];
        __syncthreads();
        const int cg = tid % VPP;
        const int co = n0 + cg * 8;
        if (co < p.Co) {
            for (int v = tid; v < 128 * VPP; v += 256) {
                const int m = v / VPP;
                if (m >= tilepx) break;
                const int ty = m / p.TW, tx = m - ty * p.TW;
                const int vr = vr0 + ty, c = c0 + tx;
                const int b = vr / vpitch, oy = vr - b * vpitch;
                if (b >= p.B || oy >= p.Ho || c >= p.Wo) continue;
                float f[8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(sOut + m * OSTR + cg * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(sOut + m * OSTR + cg * 8 + 4);
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

template <typename T, int KS, int BN>
int launch(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_core_kernel<T, KS, BN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_core_kernel<T, KS, BN>), grid, dim3(256), lds, st, k);
    STL_LAUNCH_CHECK("conv_core");
    return 0;
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
    int szMain = szA + szB;
    if (szOut > szMain) szMain = szOut;
    if (szRed > szMain) szMain = szRed;
    k.off_cs = 0;
    k.off_cm = 3 * k.cipad * 4;
    k.off_main = (k.off_cm + 4 * BN * 4 + 15) & ~15;
    k.off_b = k.off_main + szA;
    const size_t lds = (size_t)k.off_main + szMain;
    STL_CHECK(lds <= 160 * 1024, "conv: tile needs %zu B of LDS (>160 KiB); shrink TH/TW", lds);
    int gx = ceil_div(k.npt, 8) * 8;
    const int cap = (p.out_stats || p.red) ? 512 : 1024;
    if (gx > cap) gx = cap;
    dim3 grid(gx, ceil_div(p.Co, BN));
    hipStream_t st = (hipStream_t)stream;
#define DISPATCH(T)                                                      \
    if (p.ks == 3) {                                                     \
        return BN == 32 ? launch<T, 3, 32>(k, grid, lds, st) : launch<T, 3, 64>(k, grid, lds, st); \
    } else {                                                             \
        return BN == 32 ? launch<T, 1, 32>(k, grid, lds, st) : launch<T, 1, 64>(k, grid, lds, st); \
    }
    if (p.dtype == STL_BF16) {
        DISPATCH(__bf16)
    } else {
        DISPATCH(float)
    }
#undef DISPATCH
}
