// HBM-bound kernels of the HRNet step on gfx950: multi-term sum (+upsample, +BN on load, +ReLU),
// its backward reductions, stem patches, heatmap head, masked MSE, heatmap decode.
// All activation traffic is 16-byte vectors along the NHWC channel axis.
#include "common.cuh"

namespace {

template <typename T>
__device__ __forceinline__ void load8(const void* base, size_t elem, float* f) {
    if constexpr (sizeof(T) == 2) {
        unpack<T>(ldg16((const char*)base + elem * 2), f);
    } else {
        unpack<float>(ldg16((const char*)base + elem * 4), f);
        unpack<float>(ldg16((const char*)base + elem * 4 + 16), f + 4);
    }
}
template <typename T>
__device__ __forceinline__ void store8(void* base, size_t elem, const float* f) {
    if constexpr (sizeof(T) == 2) {
        stg16((char*)base + elem * 2, pack<T>(f));
    } else {
        stg16((char*)base + elem * 4, pack<float>(f));
        stg16((char*)base + elem * 4 + 16, pack<float>(f + 4));
    }
}

// Block-level flush of per-thread channel-group sums: thread t holds NS sets of 8 channel sums for
// channel group (t % VPC).  Deterministic tree in LDS, then fp64 atomics into shard (block & 7).
template <int NS>
__device__ __forceinline__ void flush_sets(float (*acc)[8], float* red, int C, int VPC, double* const* dst,
                                           const int* which, int nsets) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(tid * NS + s) * 8 + j] = acc[s][j];
    __syncthreads();
    for (int e = tid; e < nsets * C; e += nthr) {
        const int s = e / C, c = e - s * C, cg = c >> 3, j = c & 7;
        float sum = 0.f;
        for (int q = cg; q < nthr; q += VPC) sum += red[(q * NS + s) * 8 + j];
        atomicAdd(dst[s] + (size_t)(blockIdx.x & (STL_NSHARD - 1)) * 2 * C + which[s] * C + c, (double)sum);
    }
}

// ---------------------------------------------------------------- fuse forward
template <typename T>
__global__ __launch_bounds__(256) void fuse_fwd_kernel(const stl_fuse p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cs = reinterpret_cast<float*>(smem);  // [nterms][2][C]
    const int C = p.C, VPC = C >> 3;
    for (int e = threadIdx.x; e < p.nterms * C; e += blockDim.x) {
        const int t = e / C, c = e - t * C;
        float a, b, cc;
        src_consts(p.t[t].src, c, C, a, b, cc);
        cs[(t * 2) * C + c] = a, cs[(t * 2 + 1) * C + c] = b;
    }
    __syncthreads();
    // 32-bit index math (tensors hold < 2^31 elements, checked on the host); sums whose terms all have
    // the output's resolution (every residual block end) need no pixel coordinates at all
    const uint32_t total = (uint32_t)p.B * p.H * p.W * VPC;
    bool flat = true;
    for (int t = 0; t < p.nterms; ++t) flat = flat && p.t[t].shift == 0;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < total; v += gridDim.x * blockDim.x) {
        const uint32_t pi = v / (uint32_t)VPC;
        const int c0 = (int)(v - pi * VPC) * 8;
        int x = 0, y = 0, b = 0;
        if (!flat) {
            const uint32_t by = pi / (uint32_t)p.W;
            x = (int)(pi - by * p.W);
            b = (int)(by / (uint32_t)p.H);
            y = (int)(by - (uint32_t)b * p.H);
        }
        float s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.f;
        for (int t = 0; t < p.nterms; ++t) {
            const stl_term& tm = p.t[t];
            const int hs = p.H >> tm.shift, ws = p.W >> tm.shift;
            const size_t off = flat ? (size_t)v * 8 : (((size_t)b * hs + (y >> tm.shift)) * ws + (x >> tm.shift)) * C + c0;
            float f[8];
            load8<T>(tm.src.x, off, f);
            if (tm.src.mode == STL_SRC_BN) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float u = fmaf(cs[(t * 2) * C + c0 + j], f[j], cs[(t * 2 + 1) * C + c0 + j]);   // one rounding, as conv_common.inc's fma2
                    // BN terms are rounded to the storage type like a materialised BN output would be
                    f[j] = tm.src.relu ? fmaxf(u, 0.f) : u;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += f[j];
        }
        if (p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = fmaxf(s[j], 0.f);
        }
        store8<T>(p.out, (size_t)v * 8, s);
    }
}

// One- / two-term sums of the output's resolution over LARGE tensors (the 113 MB residual block ends of layer1; round 5): the
// thread's channel group never changes (the grid's thread count is a multiple of the vectors per pixel), so its BatchNorm constants
// live in registers -- no division, no LDS read per element -- and two vectors per thread are in flight: 80 -> 70 us where a plain
// 2-read-1-write kernel reaches 6.1 TB/s (tools/hbm_ceiling.py).  A kernel of its OWN: at 73 registers the sum kernel loses a
// quarter of its resident waves, and the 66 small two-term sums of a step (14 - 42 MB) ran 2 us SLOWER each with this code inline.
// Same arithmetic, same order as fuse_fwd_kernel.
template <typename T>
__global__ __launch_bounds__(256) void fuse_flat_big_kernel(const stl_fuse p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cs = reinterpret_cast<float*>(smem);  // [nterms][2][C]
    const int C = p.C, VPC = C >> 3;
    for (int e = threadIdx.x; e < p.nterms * C; e += blockDim.x) {
        const int t = e / C, c = e - t * C;
        float a, b, cc;
        src_consts(p.t[t].src, c, C, a, b, cc);
        cs[(t * 2) * C + c] = a, cs[(t * 2 + 1) * C + c] = b;
    }
    __syncthreads();
    const uint32_t total = (uint32_t)p.B * p.H * p.W * VPC;
    const uint32_t nthr = gridDim.x * blockDim.x;
    // Residual block ends (one or two terms of the output's resolution; round 5): the thread's channel group never changes
    // (the grid's thread count is a multiple of the vectors per pixel), so its BatchNorm constants live in registers -- no
    // division, no LDS read per element -- and two vectors per thread are in flight (the 113 MB sums of layer1 ran at 3.8 TB/s
    // where a plain 2-read-1-write kernel reaches 6.1 on this chip, tools/hbm_ceiling.py).  Same arithmetic, same order.
    const uint32_t v0 = blockIdx.x * blockDim.x + threadIdx.x;
    const int c0 = (int)(v0 % (uint32_t)VPC) * 8;
    float ca[2][8], cb[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ca[t][j] = t < p.nterms ? cs[(t * 2) * C + c0 + j] : 0.f;
            cb[t][j] = t < p.nterms ? cs[(t * 2 + 1) * C + c0 + j] : 0.f;
        }
    const bool two = p.nterms == 2;
    const bool bn0 = p.t[0].src.mode == STL_SRC_BN, bn1 = two && p.t[1].src.mode == STL_SRC_BN;
    const bool r0 = p.t[0].src.relu, r1 = two && p.t[1].src.relu;
    const void* x0 = p.t[0].src.x;
    const void* x1 = two ? p.t[1].src.x : p.t[0].src.x;
    auto term = [&](float* f, const float* a, const float* b, bool bn, bool relu) __attribute__((always_inline)) {
        if (bn) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float u = fmaf(a[j], f[j], b[j]);
                f[j] = relu ? fmaxf(u, 0.f) : u;
            }
        }
    };
    for (uint32_t v = v0; v < total; v += 2 * nthr) {
        const uint32_t w = v + nthr;
        const bool hw = w < total;
        float fa[8], fb[8], ga[8], gb[8];
        load8<T>(x0, (size_t)v * 8, fa);
        if (two) load8<T>(x1, (size_t)v * 8, fb);
        if (hw) {
            load8<T>(x0, (size_t)w * 8, ga);
            if (two) load8<T>(x1, (size_t)w * 8, gb);
        }
        float s[8];
        term(fa, ca[0], cb[0], bn0, r0);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.f + fa[j];
        if (two) {
            term(fb, ca[1], cb[1], bn1, r1);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += fb[j];
        }
        if (p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = fmaxf(s[j], 0.f);
        }
        store8<T>(p.out, (size_t)v * 8, s);
        if (hw) {
            term(ga, ca[0], cb[0], bn0, r0);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = 0.f + ga[j];
            if (two) {
                term(gb, ca[1], cb[1], bn1, r1);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += gb[j];
            }
            if (p.relu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] = fmaxf(s[j], 0.f);
            }
            store8<T>(p.out, (size_t)w * 8, s);
        }
    }
}

// ---------------------------------------------------------------- fuse backward
// T: the gradients dz / du; TY: the forward tensors z and bn[].x (f16 in the mixed mode)
template <typename T, typename TY = T>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 2 : 4) void fuse_bwd_kernel(const stl_fuse_bwd p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = p.C, VPC = C >> 3;
    float* mu = reinterpret_cast<float*>(smem);  // [nbn][2][C] mean, rstd
    float* red = mu + (size_t)(p.nbn > 0 ? p.nbn : 1) * 2 * C;
    for (int e = threadIdx.x; e < p.nbn * C; e += blockDim.x) {
        const int t = e / C, c = e - t * C;
        float m, r;
        bn_mean_rstd(p.bn[t], c, C, m, r);
        mu[(t * 2) * C + c] = m, mu[(t * 2 + 1) * C + c] = r;
    }
    __syncthreads();
    float acc[5][8];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[s][j] = 0.f;
    const size_t total = (size_t)p.B * p.H * p.W * VPC;
    const int cg = threadIdx.x % VPC, c0 = cg * 8;  // (gridDim*blockDim) % VPC == 0 by construction
    // Every tensor of an element is requested UP FRONT, unconditionally (absent operands alias dz[0]: same element size, the values
    // are never used): written as "load, add, next load" under run-time operand counts the loop paid one memory round trip per
    // operand -- up to nine in sequence, with four waves per CU -- and a 70 MB sum ran at 2.8 TB/s (round 5: 24.6 -> 18.6 us on the 96 x 72 x 32 branch tensor).
    const void* dzp[4];
    const void* yp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) dzp[k] = k < p.ngrads ? p.dz[k] : p.dz[0];
#pragma unroll
    for (int t = 0; t < 4; ++t) yp[t] = t < p.nbn ? p.bn[t].x : p.dz[0];
    const void* zp = p.relu ? p.z : p.dz[0];
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (size_t)gridDim.x * blockDim.x) {
        const size_t off = v * 8;
        float d[8], e1[8], e2[8], e3[8], z[8], y[4][8];
        load8<T>(dzp[0], off, d);
        load8<T>(dzp[1], off, e1);
        load8<T>(dzp[2], off, e2);
        load8<T>(dzp[3], off, e3);
        load8<TY>(zp, off, z);
#pragma unroll
        for (int t = 0; t < 4; ++t) load8<TY>(yp[t], off, y[t]);
        if (p.ngrads > 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] += e1[j];
        }
        if (p.ngrads > 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] += e2[j];
        }
        if (p.ngrads > 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] += e3[j];
        }
        if (p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = z[j] > 0.f ? d[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = round_to<T>(d[j]);
        store8<T>(p.du, off, d);
        if (p.nbn > 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[0][j] += d[j];
            // fully unrolled with static indices: a runtime index into acc[][] would put it in scratch memory
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < p.nbn) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        acc[1 + t][j] += d[j] * (y[t][j] - mu[(t * 2) * C + c0 + j]) * mu[(t * 2 + 1) * C + c0 + j];
                }
            }
        }
    }
    if (p.nbn > 0) {
        // sets: for each term t: r1 (set 0) -> rstats[t] slot 0 ; r2 (set 1+t) -> rstats[t] slot 1
        double* dst[5];
        int which[5];
        // first flush r2 sets + one r1; r1 is shared by all terms, so add it to each term's buffer
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < p.nbn) {
                dst[0] = p.rstats[t], which[0] = 0;
                dst[1] = p.rstats[t], which[1] = 1;
                float two[2][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) two[0][j] = acc[0][j], two[1][j] = acc[1 + t][j];
                flush_sets<2>(two, red, C, VPC, dst, which, 2);
            }
        }
    }
}

// ---------------------------------------------------------------- upsample backward
template <typename T, typename TY, int SHIFT>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const stl_upbwd p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = p.C, VPC = C >> 3;
    float* mu = reinterpret_cast<float*>(smem);  // [2][C]
    float* red = mu + 2 * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float m, r;
        bn_mean_rstd(p.bn, c, C, m, r);
        mu[c] = m, mu[C + c] = r;
    }
    __syncthreads();
    float acc[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[s][j] = 0.f;
    // SHIFT is a template argument and the index arithmetic 32-bit (round 5): with run-time loop bounds the f x f source vectors of an
    // output element were fetched ONE AT A TIME -- 4 / 16 / 64 memory round trips in sequence -- behind 64-bit divisions.  Now one row
    // of f vectors (and the BatchNorm input) is in flight at a time; the summation order (rows outer, columns inner) is unchanged.
    constexpr int F = 1 << SHIFT;
    const uint32_t Wh = (uint32_t)p.W << SHIFT, Hh = (uint32_t)p.H << SHIFT;
    const uint32_t total = (uint32_t)p.B * p.H * p.W * VPC;
    const int cg = threadIdx.x % VPC, c0 = cg * 8;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < total; v += gridDim.x * blockDim.x) {
        const uint32_t pi = v / (uint32_t)VPC;
        const uint32_t by = pi / (uint32_t)p.W, x = pi - by * p.W;
        const uint32_t b = by / (uint32_t)p.H, y = by - b * p.H;
        float yv[8];
        load8<TY>(p.bn.x, (size_t)pi * C + c0, yv);
        float s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.f;
        const size_t row0 = ((size_t)(b * Hh + y * F) * Wh + x * F) * C + c0;
#pragma unroll(SHIFT == 1 ? 2 : 1)
        for (int dy = 0; dy < F; ++dy) {   // (SHIFT >= 2: one row at a time -- unrolled, the compiler requests all F x F vectors at once: 150 - 210 registers)
            float d[F][8];
#pragma unroll
            for (int dx = 0; dx < F; ++dx) load8<T>(p.du, row0 + ((size_t)dy * Wh + dx) * C, d[dx]);
#pragma unroll
            for (int dx = 0; dx < F; ++dx)
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += d[dx][j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = round_to<T>(s[j]);
        store8<T>(p.dt, (size_t)pi * C + c0, s);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[0][j] += s[j];
            acc[1][j] += s[j] * (yv[j] - mu[c0 + j]) * mu[C + c0 + j];
        }
    }
    double* dst[2] = {p.rstats, p.rstats};
    int which[2] = {0, 1};
    flush_sets<2>(acc, red, C, VPC, dst, which, 2);
}

// ---------------------------------------------------------------- stem patches
template <typename T>
__global__ __launch_bounds__(256) void patch_kernel(const float* img, void* out, int B, int H, int W, int Ho, int Wo,
                                                    int stride, const float* mean3, const float* std3) {
    const size_t total = (size_t)B * Ho * Wo * 4;  // 4 vectors of 8 per pixel
    float mean[3] = {0.f, 0.f, 0.f}, istd[3] = {1.f, 1.f, 1.f};
    if (mean3) {
#pragma unroll
        for (int c = 0; c < 3; ++c) mean[c] = mean3[c], istd[c] = 1.f / std3[c];
    }
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (size_t)gridDim.x * 256) {
        const size_t pi = v >> 2;
        const int part = (int)(v & 3);
        const int ox = (int)(pi % Wo);
        const size_t by = pi / Wo;
        const int oy = (int)(by % Ho), b = (int)(by / Ho);
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = part * 8 + j;
            float val = 0.f;
            if (kk < 27) {
                const int tap = kk / 3, c = kk - tap * 3;
                const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    val = img[(((size_t)b * 3 + c) * H + iy) * W + ix];
                    if (mean3) val = (val - mean[c]) / std3[c];
                }
            }
            f[j] = val;
        }
        (void)istd;
        store8<T>(out, pi * 32 + part * 8, f);
    }
}

// ---------------------------------------------------------------- head
template <typename T, int J>
__global__ __launch_bounds__(256) void head_fwd_kernel(const void* x, const float* w, const float* bias, float* out,
                                                       int B, int HW, int Ci) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sw = reinterpret_cast<float*>(smem);  // [J][Ci]
    for (int e = threadIdx.x; e < J * Ci; e += 256) sw[e] = w[e];
    __syncthreads();
    const size_t P = (size_t)B * HW;
    for (size_t pi = (size_t)blockIdx.x * 256 + threadIdx.x; pi < P; pi += (size_t)gridDim.x * 256) {
        float o[J];
#pragma unroll
        for (int j = 0; j < J; ++j) o[j] = bias ? bias[j] : 0.f;
        for (int c0 = 0; c0 < Ci; c0 += 8) {
            float f[8];
            load8<T>(x, pi * Ci + c0, f);
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q) o[j] = fmaf(f[q], sw[j * Ci + c0 + q], o[j]);
        }
        const size_t b = pi / HW, hw = pi - b * HW;
#pragma unroll
        for (int j = 0; j < J; ++j) out[(b * J + j) * HW + hw] = o[j];
    }
}

// dx = dout . w per pixel (17 FMAs per element, VALU), and the weight / bias gradient of the block's pixels:
// dw[j][c] = sum_px d[px][j] * x[px][c] is a [J x 256] . [256 x Ci] product per 256-pixel chunk -- on the fp32 matrix
// cores (v_mfma_f32_16x16x4_f32, operands straight out of the LDS copies the chunk needs anyway: each wave a quarter
// of the pixels, accumulators kept across the block's chunks, one cross-wave reduction at the end).  The scalar LDS
// loop this replaces (561 outputs x 256 pixels x 2 reads per chunk) made the head the slowest launch per byte of the
// step: 78 us at the very start of backward, where nothing else can run.
template <typename T, int J, typename TY = T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const void* x, const float* w, const float* dout, void* dx,
                                                       float* partial, int B, int HW, int Ci) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SDP = 260;                     // row pitch of sd in floats: 16-byte aligned rows, conflict-free 16-row column reads
    constexpr int NTMAX = 4;                     // Ci <= 64
    float* sw = reinterpret_cast<float*>(smem);  // [J][Ci]
    float* sd = sw + ((J * Ci + 3) & ~3);        // [J][SDP]
    float* sx = sd + J * SDP;                    // [256][Ci+1]; reused for the cross-wave reduction at the end
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
    const int nt_n = Ci >> 4;
    for (int e = tid; e < J * Ci; e += 256) sw[e] = w[e];
    const int nel = J * Ci + J;
    f32x4 acc[2][NTMAX];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTMAX; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                            // thread (j = tid >> 2, quarter = tid & 3): bias gradient partial
    const size_t P = (size_t)B * HW;
    const size_t nchunk = (P + 255) / 256;
    for (size_t ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
        const size_t pi = ch * 256 + tid;
        __syncthreads();
        float d[J];
        if (pi < P) {
            const size_t b = pi / HW, hw = pi - b * HW;
#pragma unroll
            for (int j = 0; j < J; ++j) d[j] = dout[(b * J + j) * HW + hw];
        } else {
#pragma unroll
            for (int j = 0; j < J; ++j) d[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < J; ++j) sd[j * SDP + tid] = d[j];
        for (int c0 = 0; c0 < Ci; c0 += 8) {   // (all of a pixel's vectors up front, round 5: 214 registers + 38 AGPRs, 42 -> 56 us: not kept)
            float f[8], gx[8];
            if (pi < P) {
                load8<TY>(x, pi * Ci + c0, f);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                sx[tid * (Ci + 1) + c0 + q] = f[q];
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < J; ++j) s = fmaf(d[j], sw[j * Ci + c0 + q], s);
                gx[q] = s;
            }
            if (pi < P) store8<T>(dx, pi * Ci + c0, gx);
        }
        __syncthreads();
        // ---- dw += d^T . x over this wave's 64 pixels: A[i = j][k = pixel] from sd, B[k = pixel][n = c] from sx
        const int px0 = wave * 64 + g;
        const float* arow0 = sd + r16 * SDP + px0;                      // j = r16       (always < J)
        const float* arow1 = sd + (16 + r16 < J ? 16 + r16 : 0) * SDP + px0;   // j = 16 + r16  (zeroed when >= J)
        const bool a1ok = 16 + r16 < J;
        const float* brow = sx + (size_t)px0 * (Ci + 1) + r16;
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const float a0 = arow0[4 * ks];
            const float a1 = a1ok ? arow1[4 * ks] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NTMAX; ++nt) {
                if (nt < nt_n) {
                    const float bv = brow[(size_t)4 * ks * (Ci + 1) + nt * 16];
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][nt], 0, 0, 0);
                }
            }
        }
        // ---- db: 4 threads per joint, 64 pixels each
        if (tid < 4 * J) {
            const float* r = sd + (tid >> 2) * SDP + (tid & 3) * 64;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(r + 4 * q);
                s += (v.x + v.y) + (v.z + v.w);
            }
            bsum += s;
        }
    }
    // ---- cross-wave reduction through LDS (fixed order: deterministic), one slab per block
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(sx);   // [wave][mt * NTMAX + nt][lane]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTMAX; ++nt)
            if (nt < nt_n) red[(wave * 2 * NTMAX + mt * NTMAX + nt) * 64 + lane] = acc[mt][nt];
    float* bred = reinterpret_cast<float*>(red + 4 * 2 * NTMAX * 64);   // [4 J]
    if (tid < 4 * J) bred[tid] = bsum;
    __syncthreads();
    for (int e = tid; e < nel; e += 256) {
        float s;
        if (e < J * Ci) {
            // D[i][n]: lane = 16 * (i / 4) + n, register i % 4;  i = j % 16, n = c % 16
            const int j = e / Ci, c = e - j * Ci;
            const int tile = (j >> 4) * NTMAX + (c >> 4), ln = 16 * ((j & 15) >> 2) + (c & 15), rr = j & 3;
            s = (red[(0 * 2 * NTMAX + tile) * 64 + ln][rr] + red[(1 * 2 * NTMAX + tile) * 64 + ln][rr]) +
                (red[(2 * 2 * NTMAX + tile) * 64 + ln][rr] + red[(3 * 2 * NTMAX + tile) * 64 + ln][rr]);
        } else {
            const int j = e - J * Ci;
            s = (bred[4 * j] + bred[4 * j + 1]) + (bred[4 * j + 2] + bred[4 * j + 3]);
        }
        partial[(size_t)blockIdx.x * nel + e] = s;
    }
}

// ---------------------------------------------------------------- loss
__global__ __launch_bounds__(256) void mse_kernel(const float* o, const float* t, const float* tw, float* dout,
                                                  double* partial, size_t n, int HW, float gs) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float w = tw[i / HW];
        const float d = (o[i] - t[i]) * w;
        acc += (double)d * (double)d;
        if (dout) dout[i] = d * w * gs;
    }
    __shared__ double sred[4];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sred[0] + sred[1] + sred[2] + sred[3];
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* partial, int n, double scale, float* out, int accumulate) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    __shared__ double sred[4];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = (sred[0] + sred[1] + sred[2] + sred[3]) * scale;
        out[0] = (float)((accumulate ? (double)out[0] : 0.0) + s);
    }
}

// ---------------------------------------------------------------- decode
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
    // np.argmax order: NaN beats everything, then larger value, ties -> smaller index
    const bool vn = v != v, bn = bv != bv;
    if (vn != bn) return vn;
    if (vn) return i < bi;
    return v > bv || (v == bv && i < bi);
}
__device__ __forceinline__ void block_argmax(const float* row, int n, float& bv, int& bi, float* sv, int* si) {
    bv = -INFINITY, bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = row[i];
        if (bi == 0x7fffffff || better(v, i, bv, bi)) bv = v, bi = i;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || better(ov, oi, bv, bi))) bv = ov, bi = oi;
    }
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = bv, si[threadIdx.x >> 6] = bi;
    __syncthreads();
    bv = sv[0], bi = si[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
        if (si[w] != 0x7fffffff && (bi == 0x7fffffff || better(sv[w], si[w], bv, bi))) bv = sv[w], bi = si[w];
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* hm, int32_t* idx, float* maxval, float* preds, int H, int W) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const float* row = hm + (size_t)blockIdx.x * H * W;
    float bv;
    int bi;
    block_argmax(row, H * W, bv, bi, sv, si);
    if (threadIdx.x == 0) {
        if (idx) idx[blockIdx.x] = bi;
        maxval[blockIdx.x] = bv;
        const float m = bv > 0.f ? 1.f : 0.f;
        preds[2 * blockIdx.x] = (float)(bi % W) * m;
        preds[2 * blockIdx.x + 1] = (float)(bi / W) * m;
    }
}

__global__ __launch_bounds__(256) void final_preds_kernel(const float* hm, const float* center, const float* scale,
                                                          float* preds, float* maxval, int J, int H, int W) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const float* row = hm + (size_t)blockIdx.x * H * W;
    float bv;
    int bi;
    block_argmax(row, H * W, bv, bi, sv, si);
    if (threadIdx.x == 0) {
        const float m = bv > 0.f ? 1.f : 0.f;
        float cx = (float)(bi % W) * m, cy = (float)(bi / W) * m;
        const int px = (int)floorf(cx + 0.5f), py = (int)floorf(cy + 0.5f);
        if (1 < px && px < W - 1 && 1 < py && py < H - 1) {
            const float dx = row[py * W + px + 1] - row[py * W + px - 1];
            const float dy = row[(py + 1) * W + px] - row[(py - 1) * W + px];
            cx += (dx > 0.f) ? 0.25f : (dx < 0.f ? -0.25f : 0.f);
            cy += (dy > 0.f) ? 0.25f : (dy < 0.f ? -0.25f : 0.f);
        }
        const int b = blockIdx.x / J;
        // inverse of the crop transform (rot = 0): scale s = W / (scale_x * 200) on both axes
        const double s = (double)W / ((double)scale[2 * b] * 200.0);
        preds[2 * blockIdx.x] = (float)(((double)cx - 0.5 * W) / s + (double)center[2 * b]);
        preds[2 * blockIdx.x + 1] = (float)(((double)cy - 0.5 * H) / s + (double)center[2 * b + 1]);
        maxval[blockIdx.x] = bv;
    }
}

__global__ __launch_bounds__(256) void flip_merge_kernel(const float* a, const float* bf, float* out, const int32_t* perm,
                                                         int J, int H, int W, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % W);
        const size_t r = i / W;
        const int y = (int)(r % H);
        const size_t bj = r / H;
        const int j = (int)(bj % J);
        const size_t b = bj / J;
        const int xs = x > 0 ? x - 1 : 0;  // 1-px right shift, column 0 keeps its own value
        const float f = bf[((b * J + perm[j]) * H + y) * W + (W - 1 - xs)];
        out[i] = 0.5f * (a[i] + f);
    }
}

// ---------------------------------------------------------------- batched tables
__device__ __forceinline__ int find_entry(const int* blk0_first, int stride_ints, int n, int blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blk0_first[(size_t)mid * stride_ints] <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// 4 consecutive elements as ONE 8-byte (bf16) / 16-byte (fp32) store; offsets are multiples of 4 elements
__device__ __forceinline__ void store_vec4(__bf16* dst, const __bf16* v) {
    uint2 u;
    u.x = (uint32_t)__builtin_bit_cast(unsigned short, v[0]) | ((uint32_t)__builtin_bit_cast(unsigned short, v[1]) << 16);
    u.y = (uint32_t)__builtin_bit_cast(unsigned short, v[2]) | ((uint32_t)__builtin_bit_cast(unsigned short, v[3]) << 16);
    *reinterpret_cast<uint2*>(dst) = u;
}
__device__ __forceinline__ void store_vec4(f16* dst, const f16* v) {
    uint2 u;
    u.x = (uint32_t)__builtin_bit_cast(unsigned short, v[0]) | ((uint32_t)__builtin_bit_cast(unsigned short, v[1]) << 16);
    u.y = (uint32_t)__builtin_bit_cast(unsigned short, v[2]) | ((uint32_t)__builtin_bit_cast(unsigned short, v[3]) << 16);
    *reinterpret_cast<uint2*>(dst) = u;
}
__device__ __forceinline__ void store_vec4(float* dst, const float* v) {
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
}

// T: element type of the data-gradient layouts; TF: of the forward layouts (f16 in the mixed mode, else T)
template <typename T, typename TF = T>
__global__ __launch_bounds__(256) void weight_prep_kernel(const float* master, T* wk, const stl_wprep* tab, int n, int blk_base) {
    static_assert(sizeof(T) == sizeof(TF), "both layouts live in one buffer of equal-width elements");
    TF* wkf = reinterpret_cast<TF*>(wk);
    const int bx = blockIdx.x + blk_base;  // tab points at the first entry of the range, blk0 values are table-absolute
    const int ei = find_entry(&tab[0].blk0, sizeof(stl_wprep) / 4, n, bx);
    const stl_wprep e = tab[ei];
    const int t = e.ks * e.ks;
    const int64_t tot = (int64_t)e.Co * e.Ci * t;
    const int64_t base = (int64_t)(bx - e.blk0) * 1024;
    if (!e.patch && (e.Ci & 3) == 0 && (e.Co & 3) == 0 && t <= 9 && e.Cip == e.Ci && (e.fwd_off & 3) == 0 && (e.bwd_off < 0 || (e.bwd_off & 3) == 0)) {   // row pitch Ci, 8/16-byte aligned vector stores
        // 32 x 32 channel tiles through LDS: the OIHW master is read ONCE in contiguous (32 ci x taps) row segments,
        // both kernel layouts are written in 4-element vectors that form 64-byte runs ([co][tap][ci0..ci0+32) and
        // [ci][flipped tap][co0..co0+32)).  The table still counts 1024-element blocks per conv; the blocks of a conv
        // share its tiles round robin.  (Output-major gathers -- 4-byte reads at stride taps resp. Ci*taps -- moved
        // 8x the bytes of the master: 158 us at the head of every step.)
        __shared__ float st[32 * (32 * 9 + 1)];
        const int pitch = 32 * t + 1;
        const int tci = (e.Ci + 31) >> 5, ntiles = ((e.Co + 31) >> 5) * tci;
        const int nblk = (int)((tot + 1023) / 1024);
        const float* src0 = master + e.src_off;
        for (int tile = bx - e.blk0; tile < ntiles; tile += nblk) {
            const int co0 = (tile / tci) * 32, ci0 = (tile % tci) * 32;
            const int nco = min(32, e.Co - co0), nci = min(32, e.Ci - ci0), seg = nci * t;
            __syncthreads();
            if (((e.src_off & 3) | (int)((uintptr_t)master & 15)) == 0) {
                // 16-byte loads, all of a thread's (up to nine) in flight at once: (row, quad) tasks over the block.  (Round 5: the
                // 4-byte row walk had one small load per lane in flight -- 94 us for 114 MB in + 114 MB out.)  Rows start on
                // 16-byte boundaries: Ci and ci0 * taps are multiples of 4.
                const int nv = seg >> 2, ntask = nco * nv;
                float4 buf[9];
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    const int i = threadIdx.x + u * 256;
                    const bool ok = i < ntask;
                    const int r = ok ? i / nv : 0, q = ok ? i - r * nv : 0;
                    buf[u] = *reinterpret_cast<const float4*>(src0 + ((int64_t)(co0 + r) * e.Ci + ci0) * t + 4 * q);
                }
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    const int i = threadIdx.x + u * 256;
                    if (i < ntask) {
                        const int r = i / nv, q = i - r * nv;
                        float* d = st + r * pitch + 4 * q;
                        d[0] = buf[u].x, d[1] = buf[u].y, d[2] = buf[u].z, d[3] = buf[u].w;
                    }
                }
            } else {
                for (int r = threadIdx.x >> 6; r < nco; r += 4) {   // one wave per co row: contiguous, division-free
                    const float* row = src0 + ((int64_t)(co0 + r) * e.Ci + ci0) * t;
                    for (int c = threadIdx.x & 63; c < seg; c += 64) st[r * pitch + c] = row[c];
                }
            }
            __syncthreads();
            T v[4];
            TF vf[4];
            // forward layout [co][tap][ci]: task = (row = co_l * t + tap, 4 ci); taps are 1 or 9 (constant divisors)
            const int q4 = nci >> 2;
            for (int i = threadIdx.x; i < nco * t * q4; i += 256) {
                const int row = q4 == 8 ? i >> 3 : i / q4, c4 = (i - row * q4) * 4;
                const int r = t == 9 ? row / 9 : (t == 1 ? row : row / t), tap = row - r * t;
#pragma unroll
                for (int j = 0; j < 4; ++j) vf[j] = (TF)st[r * pitch + (c4 + j) * t + tap];
                store_vec4(wkf + e.fwd_off + ((int64_t)(co0 + r) * t + tap) * e.Ci + ci0 + c4, vf);
            }
            if (e.bwd_off >= 0) {   // data-gradient layout [ci][flipped tap][co]: task = (row = ci_l * t + tapf, 4 co)
                const int p4 = nco >> 2;
                for (int i = threadIdx.x; i < nci * t * p4; i += 256) {
                    const int row = p4 == 8 ? i >> 3 : i / p4, r4 = (i - row * p4) * 4;
                    const int c = t == 9 ? row / 9 : (t == 1 ? row : row / t), tapf = row - c * t;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (T)st[(r4 + j) * pitch + c * t + (t - 1 - tapf)];
                    store_vec4(wk + e.bwd_off + ((int64_t)(ci0 + c) * t + tapf) * e.Co + co0 + r4, v);
                }
            }
        }
        return;
    }
    for (int r = 0; r < 4; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        if (i >= tot) return;
        const int co = (int)(i / (e.Ci * t)), rem = (int)(i - (int64_t)co * e.Ci * t), ci = rem / t, tap = rem - ci * t;
        const float v = master[e.src_off + i];
        if (e.patch)
            wkf[e.fwd_off + (int64_t)co * e.Cip + tap * e.Ci + ci] = (TF)v;
        else
            wkf[e.fwd_off + ((int64_t)co * t + tap) * e.Cip + ci] = (TF)v;
        if (e.bwd_off >= 0) wk[e.bwd_off + ((int64_t)ci * t + (t - 1 - tap)) * e.Co + co] = (T)v;
    }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* partials, float* grads, const stl_slab* tab, int n, int blk_base) {
    const int bx = blockIdx.x + blk_base;  // tab points at the first entry of the range, blk0 values are table-absolute
    const int ei = find_entry(&tab[0].blk0, sizeof(stl_slab) / 4, n, bx);
    const stl_slab e = tab[ei];
    const int t = e.ks * e.ks;
    const int64_t tot = (int64_t)e.Co * e.Ci * t;
    const int tk = e.patch ? 1 : t, cik = e.patch ? e.Cip : e.Ci;
    const int64_t slab = e.pad ? (int64_t)e.pad : (int64_t)e.Co * tk * cik;
    const int64_t base = (int64_t)(bx - e.blk0) * 1024;
    if (!e.patch && (e.Ci & 3) == 0) {
        // fast path: walk the slabs in THEIR order ([Co][taps][Ci], 16-byte coalesced reads, four slabs
        // in flight), scatter the four sums into the OIHW gradient (1/nsplit of the traffic)
        const int64_t j = base + 4 * threadIdx.x;
        if (j >= tot) return;
        const float* src = partials + e.part_off + j;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        // eight slabs in flight, added in slab order (same sums as four at a time, half the dependent memory round trips: the
        // bucket that closes the step sums 125 slabs per element on 110 blocks, alone on the chip)
        for (; k + 8 <= e.nsplit; k += 8) {
            float4 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const float4*>(src + (int64_t)(k + u) * slab);
#pragma unroll
            for (int u = 0; u < 8; ++u) s.x += a[u].x, s.y += a[u].y, s.z += a[u].z, s.w += a[u].w;
        }
        for (; k + 4 <= e.nsplit; k += 4) {
            const float4 a0 = *reinterpret_cast<const float4*>(src + (int64_t)(k + 0) * slab);
            const float4 a1 = *reinterpret_cast<const float4*>(src + (int64_t)(k + 1) * slab);
            const float4 a2 = *reinterpret_cast<const float4*>(src + (int64_t)(k + 2) * slab);
            const float4 a3 = *reinterpret_cast<const float4*>(src + (int64_t)(k + 3) * slab);
            s.x += a0.x, s.y += a0.y, s.z += a0.z, s.w += a0.w;
            s.x += a1.x, s.y += a1.y, s.z += a1.z, s.w += a1.w;
            s.x += a2.x, s.y += a2.y, s.z += a2.z, s.w += a2.w;
            s.x += a3.x, s.y += a3.y, s.z += a3.z, s.w += a3.w;
        }
        for (; k < e.nsplit; ++k) {
            const float4 a0 = *reinterpret_cast<const float4*>(src + (int64_t)k * slab);
            s.x += a0.x, s.y += a0.y, s.z += a0.z, s.w += a0.w;
        }
        const int co = (int)(j / (t * e.Ci)), rem = (int)(j - (int64_t)co * t * e.Ci), tap = rem / e.Ci, ci = rem - tap * e.Ci;
        float* d = grads + e.grad_off + ((int64_t)co * e.Ci + ci) * t + tap;
        d[0] = s.x, d[t] = s.y, d[2 * t] = s.z, d[3 * t] = s.w;
        return;
    }
    for (int r = 0; r < 4; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        if (i >= tot) return;
        const int co = (int)(i / (e.Ci * t)), rem = (int)(i - (int64_t)co * e.Ci * t), ci = rem / t, tap = rem - ci * t;
        const int64_t src = e.patch ? ((int64_t)co * cik + tap * e.Ci + ci) : (((int64_t)co * t + tap) * cik + ci);
        // eight slabs in flight (a serial loop is one memory round trip per slab: 124 us for the 128 slabs of the
        // stem's patch conv, at the very end of the step); fixed summation order -> deterministic
        const float* pp = partials + e.part_off + src;
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= e.nsplit; k += 8) {
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = pp[(int64_t)(k + u) * slab];
            s += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        }
        for (; k < e.nsplit; ++k) s += pp[(int64_t)k * slab];
        grads[e.grad_off + i] = s;
    }
}

__global__ __launch_bounds__(256) void bn_running_kernel(const double* stats, float* buffers, int64_t* nbt,
                                                         const stl_bnrec* tab, float momentum, int32_t* overflow) {
    const stl_bnrec e = tab[blockIdx.x];
    const double n = 1.0 / (double)e.inv_count;
    bool bad = false;
    for (int c = threadIdx.x; c < e.C; c += 256) {
        double s0 = 0.0, s1 = 0.0;
        for (int k = 0; k < STL_NSHARD; ++k) {
            s0 += stats[e.stats_off + (int64_t)k * 2 * e.C + c];
            s1 += stats[e.stats_off + (int64_t)k * 2 * e.C + e.C + c];
        }
        // A raw conv output beyond the range of its storage type (f16 forward tensors of the mixed mode: |y| > 65504) was stored
        // as infinity, and the sums of the values AS STORED carry it: the layer's running statistics stay as they are and the
        // FIRST such layer (smallest index) is reported through `overflow` (engine.Engine.check_forward_range).
        if (!(fabs(s0) <= 1.7e308 && fabs(s1) <= 1.7e308)) {
            bad = true;
            continue;
        }
        const double mean = s0 / n;
        double var = s1 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
        float* rm = buffers + e.buf_off;
        float* rv = rm + e.C;
        rm[c] = (float)((1.0 - momentum) * (double)rm[c] + momentum * mean);
        rv[c] = (float)((1.0 - momentum) * (double)rv[c] + momentum * unb);
    }
    if (threadIdx.x == 0 && nbt) nbt[blockIdx.x] += 1;
    if (bad && overflow) atomicMin(overflow, (int32_t)blockIdx.x);
}

__global__ __launch_bounds__(256) void bn_param_grads_kernel(const double* rstats, float* grads, const stl_bnrec* tab) {
    const stl_bnrec e = tab[blockIdx.x];
    for (int c = threadIdx.x; c < e.C; c += 256) {
        double r1 = 0.0, r2 = 0.0;
        for (int k = 0; k < STL_NSHARD; ++k) {
            r1 += rstats[e.stats_off + (int64_t)k * 2 * e.C + c];
            r2 += rstats[e.stats_off + (int64_t)k * 2 * e.C + e.C + c];
        }
        grads[e.param_off + c] = (float)r2;        // dgamma
        grads[e.param_off + e.C + c] = (float)r1;  // dbeta
    }
}

// ---------------------------------------------------------------- optimisers
// step counter of the fused optimisers.  A NEGATIVE value means "skip this step": the forward pass of the step stored a
// non-finite forward tensor (stl_bn_running_update's overflow word; ReLU(NaN) = 0 lets the loss and the gradients come out
// finite and wrong, so the gradient alone cannot tell) -- the weights must not move.  The skipped step is not counted.
__global__ void inc_step_kernel(int32_t* step, const int32_t* overflow) {
    const int t = step[0] < 0 ? -step[0] - 1 : step[0];   // steps taken so far (a skipped step is stored as -t - 1)
    step[0] = (overflow && overflow[0] != 0x7fffffff) ? -t - 1 : t + 1;
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                   const float* hyper, const int32_t* step) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], gs = hyper[7];
    const int t = step[0];
    if (t < 0) return;   // skipped step (inc_step_kernel)
    const float bc1 = 1.f - powf(b1, (float)t), bc2 = 1.f - powf(b2, (float)t);
    const float step_size = lr / bc1, isq2 = 1.f / sqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float gi = g[i] * gs;
        if (!(fabsf(gi) <= 3.4e38f)) continue;   // non-finite gradient (forward overflow, see bn_running_kernel): keep the weights
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi, v[i] = vi;
        p[i] = pi - step_size * mi / (sqrtf(vi) * isq2 + eps);
    }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* p, const float* g, float* mom, int64_t n, const float* hyper,
                                                  const int32_t* step) {
    const float lr = hyper[0], wd = hyper[4], mu = hyper[5], nest = hyper[6], gs = hyper[7];
    if (step[0] < 0) return;   // skipped step (inc_step_kernel)
    const bool first = step[0] <= 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float gi = g[i] * gs;
        if (!(fabsf(gi) <= 3.4e38f)) continue;   // non-finite gradient: keep the weights (adam_kernel)
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        if (mu != 0.f) {
            const float b = first ? gi : mu * mom[i] + gi;
            mom[i] = b;
            gi = nest != 0.f ? gi + mu * b : b;
        }
        p[i] = pi - lr * gi;
    }
}


// ---------------------------------------------------------------- affine crop (data/JointsDataset.py:189-200)
// out[b, c, y, x] = (bilinear(src_b, Minv_b * (x, y, 1)) / 255 - mean[c]) / std[c]; taps outside the source image
// contribute 0 (cv2.warpAffine's default BORDER_CONSTANT); flip: the source is read mirrored left-right
// (data_numpy[:, ::-1, :]).  Source images are uint8 HWC (RGB) packed in one buffer.
__global__ __launch_bounds__(256) void affine_crop_kernel(const uint8_t* src, const int64_t* src_off, const int32_t* src_hw,
                                                          const float* minv, const int32_t* flip, float* out, int Ho, int Wo,
                                                          const float* mean3, const float* std3) {
    const int b = blockIdx.y;
    const int H = src_hw[2 * b], W = src_hw[2 * b + 1];
    const uint8_t* img = src + src_off[b];
    const float m0 = minv[6 * b], m1 = minv[6 * b + 1], m2 = minv[6 * b + 2];
    const float m3 = minv[6 * b + 3], m4 = minv[6 * b + 4], m5 = minv[6 * b + 5];
    const bool fl = flip && flip[b];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Ho * Wo; i += gridDim.x * 256) {
        const int y = i / Wo, x = i - y * Wo;
        const float sx = m0 * x + m1 * y + m2, sy = m3 * x + m4 * y + m5;
        const float fx = floorf(sx), fy = floorf(sy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float ax = sx - fx, ay = sy - fy;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int xx = x0 + dx, yy = y0 + dy;
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                    const float w = (dx ? ax : 1.f - ax) * (dy ? ay : 1.f - ay);
                    const uint8_t* px = img + ((size_t)yy * W + (fl ? W - 1 - xx : xx)) * 3;
                    acc[0] += w * px[0], acc[1] += w * px[1], acc[2] += w * px[2];
                }
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = acc[c] * (1.f / 255.f);
            if (mean3) v = (v - mean3[c]) / std3[c];
            out[(((size_t)b * 3 + c) * Ho + y) * Wo + x] = v;
        }
    }
}

// ---------------------------------------------------------------- VGG helpers
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const void* x, void* out, int B, int H, int W, int C) {
    const int Ho = H >> 1, Wo = W >> 1, VPC = C >> 3;
    const size_t total = (size_t)B * Ho * Wo * VPC;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < total; v += (size_t)gridDim.x * 256) {
        const size_t pi = v / VPC;
        const int c0 = (int)(v - pi * VPC) * 8;
        const int ox = (int)(pi % Wo);
        const size_t by = pi / Wo;
        const int oy = (int)(by % Ho), b = (int)(by / Ho);
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                float f[8];
                load8<T>(x, (((size_t)b * H + 2 * oy + dy) * W + 2 * ox + dx) * C + c0, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], f[j]);
            }
        store8<T>(out, pi * C + c0, m);
    }
}

template <typename T, bool SQ>
__global__ __launch_bounds__(256) void l1_kernel(const void* a, const void* b, size_t nvec, double* partial) {
    double acc = 0.0;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) {
        float x[8], y[8];
        load8<T>(a, v * 8, x);
        load8<T>(b, v * 8, y);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += SQ ? (x[j] - y[j]) * (x[j] - y[j]) : fabsf(x[j] - y[j]);
        acc += (double)s;
    }
    __shared__ double sred[4];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sred[0] + sred[1] + sred[2] + sred[3];
}

__global__ __launch_bounds__(256) void bilinear_kernel(const float* in, float* out, int BC, int H, int W, int Ho, int Wo) {
    const float sy = (float)H / Ho, sx = (float)W / Wo;
    const size_t total = (size_t)BC * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % Wo);
        const size_t r = i / Wo;
        const int oy = (int)(r % Ho);
        const size_t bc = r / Ho;
        float fy = sy * (oy + 0.5f) - 0.5f, fx = sx * (ox + 0.5f) - 0.5f;
        if (fy < 0.f) fy = 0.f;
        if (fx < 0.f) fx = 0.f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0;
        const float* p = in + bc * H * W;
        out[i] = (1.f - ly) * ((1.f - lx) * p[y0 * W + x0] + lx * p[y0 * W + x1]) +
                 ly * ((1.f - lx) * p[y1 * W + x0] + lx * p[y1 * W + x1]);
    }
}

// ---------------------------------------------------------------- layout
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* in, T* out, int B, int C, int HW) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t r = i / C;
        const size_t hw = r % HW, b = r / HW;
        out[i] = (T)in[(b * C + c) * HW + hw];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* in, float* out, int B, int C, int HW) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t hw = i % HW;
        const size_t r = i / HW;
        const int c = (int)(r % C);
        const size_t b = r / C;
        out[i] = (float)in[(b * HW + hw) * C + c];
    }
}

// ---------------------------------------------------------------- self test
__global__ void selftest_kernel(float* out) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[32 * 16];  // [32 k][16 cols] bf16
    __shared__ float ref[16 * 16];
    const int lane = threadIdx.x, r16 = lane & 15, g = lane >> 4;
    // A[m][k] = ((m*3 + k*5) % 7) - 3, B[k][n] = ((k*2 + n*7) % 5) - 2  (exact in bf16; asymmetric)
    float err_bf = 0.f, err_f32 = 0.f;
    {
        V16 a, b;
        float fa[8], fb[8];
        for (int j = 0; j < 8; ++j) {
            const int kk = 8 * g + j;
            fa[j] = (float)(((r16 * 3 + kk * 5) % 7) - 3);
            fb[j] = (float)(((kk * 2 + r16 * 7) % 5) - 2);
        }
        a = pack<__bf16>(fa), b = pack<__bf16>(fb);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mma16<__bf16>(acc, a, b);
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * g + r, nn = r16;
            float s = 0.f;
            for (int kk = 0; kk < 32; ++kk) s += (float)(((m * 3 + kk * 5) % 7) - 3) * (float)(((kk * 2 + nn * 7) % 5) - 2);
            err_bf = fmaxf(err_bf, fabsf(s - acc[r]));
        }
    }
    {
        V16 a, b;
        float fa[4], fb[4];
        for (int s = 0; s < 4; ++s) {
            const int kk = 4 * g + s;
            fa[s] = (float)(((r16 * 3 + kk * 5) % 7) - 3);
            fb[s] = (float)(((kk * 2 + r16 * 7) % 5) - 2);
        }
        a = pack<float>(fa), b = pack<float>(fb);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        mma16<float>(acc, a, b);
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * g + r, nn = r16;
            float s = 0.f;
            for (int kk = 0; kk < 16; ++kk) s += (float)(((m * 3 + kk * 5) % 7) - 3) * (float)(((kk * 2 + nn * 7) % 5) - 2);
            err_f32 = fmaxf(err_f32, fabsf(s - acc[r]));
        }
    }
    // transposed LDS read: tile[k][col] = k*16 + col ; expect lane (col = r16, group g) to get
    // rows 8g+4h+{0..3} of column r16
    for (int e = lane; e < 32 * 16; e += 64) tile[e] = (unsigned short)e;
    __syncthreads();
    float err_tr = 0.f;
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * g + 4 * h + (r16 >> 2);
        const char* ptr = (const char*)tile + row * 32 + (lane & 3) * 8;
        s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(uintptr_t)(uint32_t)(uintptr_t)ptr);
        for (int q = 0; q < 4; ++q) {
            const int expect = (8 * g + 4 * h + q) * 16 + r16;
            err_tr = fmaxf(err_tr, fabsf((float)(unsigned short)r[q] - (float)expect));
        }
    }
    (void)ref;
    err_bf = wave_sum(err_bf), err_f32 = wave_sum(err_f32), err_tr = wave_sum(err_tr);
    if (lane == 0) out[0] = err_bf, out[1] = err_f32, out[2] = err_tr;
}
__global__ void selftest_atomic_kernel(double* d) { atomicAdd(d + (threadIdx.x & 1), 0.5); }
__global__ void selftest_fin_kernel(const double* d, float* out) { out[3] = (float)fabs(d[0] - 64.0) + (float)fabs(d[1] - 64.0); }

inline int nblocks_for(size_t work, int per_block = 256, int cap = 2048) {
    size_t b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int stl_fuse_forward(const stl_fuse* pp, void* stream) {
    const stl_fuse& p = *pp;
    STL_CHECK(p.C % 8 == 0 && p.C > 0 && p.C <= 1024, "fuse: C=%d must be a multiple of 8 (<=1024)", p.C);
    STL_CHECK(p.nterms >= 1 && p.nterms <= 4, "fuse: nterms %d", p.nterms);
    for (int t = 0; t < p.nterms; ++t) {
        STL_CHECK(p.t[t].src.x, "fuse: null term %d", t);
        STL_CHECK(p.t[t].src.mode == STL_SRC_PLAIN || p.t[t].src.mode == STL_SRC_BN, "fuse: term mode");
        STL_CHECK(p.t[t].shift >= 0 && p.t[t].shift <= 4 && (p.H % (1 << p.t[t].shift)) == 0 && (p.W % (1 << p.t[t].shift)) == 0,
                  "fuse: %dx%d not divisible by 2^%d", p.H, p.W, p.t[t].shift);
    }
    const size_t total = (size_t)p.B * p.H * p.W * (p.C / 8);
    STL_CHECK(total * 8 < (1ull << 31), "fuse: tensors of 2^31 or more elements are not supported");
    const size_t lds = (size_t)p.nterms * 2 * p.C * 4;
    const dim3 grid(nblocks_for(total, 256, 2048));
    bool flat = p.nterms <= 2;
    for (int t = 0; t < p.nterms; ++t) flat = flat && p.t[t].shift == 0;
    // large one- / two-term sums of the output's resolution: the register-resident form (see fuse_flat_big_kernel)
    const bool big = flat && total >= ((size_t)1 << 21) && ((size_t)grid.x * 256) % (size_t)(p.C / 8) == 0;
#define STL_FUSE(TT)                                                                   \
    do {                                                                               \
        if (big) STL_LAUNCH(fuse_flat_big_kernel<TT>, grid, dim3(256), lds, ST, p);    \
        else STL_LAUNCH(fuse_fwd_kernel<TT>, grid, dim3(256), lds, ST, p);             \
    } while (0)
    if (p.dtype == STL_BF16)
        STL_FUSE(__bf16);
    else if (p.dtype == STL_F16)
        STL_FUSE(f16);
    else
        STL_FUSE(float);
#undef STL_FUSE
    STL_LAUNCH_CHECK("fuse_forward");
    return 0;
}

static int stat_block(int C) {
    const int vpc = C / 8;
    return vpc * (256 / vpc);
}

extern "C" int stl_fuse_backward(const stl_fuse_bwd* pp, void* stream) {
    const stl_fuse_bwd& p = *pp;
    STL_CHECK(p.C % 8 == 0 && p.C > 0 && p.C <= 1024, "fuse_bwd: C=%d", p.C);
    STL_CHECK(p.ngrads >= 1 && p.ngrads <= 4 && p.nbn >= 0 && p.nbn <= 4, "fuse_bwd: ngrads/nbn");
    STL_CHECK(!p.relu || p.z, "fuse_bwd: relu needs z");
    for (int t = 0; t < p.nbn; ++t) STL_CHECK(p.bn[t].x && p.bn[t].stats && p.rstats[t], "fuse_bwd: bn term %d incomplete", t);
    const int bd = stat_block(p.C);
    const size_t total = (size_t)p.B * p.H * p.W * (p.C / 8);
    // grid: 256 blocks measured best end to end for the branch tensors (fewer statistics atomics, less
    // contention with co-running kernels); the 113 MB layer1 tensors need more loads in flight
    int cap = (int)(total / (size_t)(bd * 16));   // (8 / 4 vectors per thread, i.e. 2 / 4 x the blocks: 13.09 / 13.20 vs 13.07 ms per step, round 5)
    cap = cap < 256 ? 256 : (cap > 1024 ? 1024 : cap);
    int nb = nblocks_for(total, bd, cap);
    const size_t lds = (size_t)(p.nbn > 0 ? p.nbn : 1) * 2 * p.C * 4 + (size_t)bd * 2 * 8 * 4;
    STL_CHECK(p.ydtype == 0 || p.ydtype == p.dtype || (p.dtype == STL_BF16 && p.ydtype == STL_F16), "fuse_bwd: ydtype %d does not go with dtype %d", p.ydtype, p.dtype);
    if (p.dtype == STL_BF16 && p.ydtype == STL_F16)
        STL_LAUNCH((fuse_bwd_kernel<__bf16, f16>), dim3(nb), dim3(bd), lds, ST, p);
    else if (p.dtype == STL_BF16)
        STL_LAUNCH(fuse_bwd_kernel<__bf16>, dim3(nb), dim3(bd), lds, ST, p);
    else
        STL_LAUNCH(fuse_bwd_kernel<float>, dim3(nb), dim3(bd), lds, ST, p);
    STL_LAUNCH_CHECK("fuse_backward");
    return 0;
}

extern "C" int stl_upsample_backward(const stl_upbwd* pp, void* stream) {
    const stl_upbwd& p = *pp;
    STL_CHECK(p.C % 8 == 0 && p.C > 0 && p.C <= 1024, "upsample_bwd: C=%d", p.C);
    STL_CHECK(p.shift >= 1 && p.shift <= 3, "upsample_bwd: shift %d (1 .. 3)", p.shift);
    STL_CHECK(p.du && p.dt && p.bn.x && p.bn.stats && p.rstats, "upsample_bwd: null pointer");
    const int bd = stat_block(p.C);
    const size_t total = (size_t)p.B * p.H * p.W * (p.C / 8);
    // grid: 256 blocks measured best end to end for the branch tensors (fewer statistics atomics, less
    // contention with co-running kernels); the 113 MB layer1 tensors need more loads in flight
    int cap = (int)(total / (size_t)(bd * 16));
    cap = cap < 256 ? 256 : (cap > 1024 ? 1024 : cap);
    int nb = nblocks_for(total, bd, cap);
    const size_t lds = (size_t)2 * p.C * 4 + (size_t)bd * 2 * 8 * 4;
    STL_CHECK(p.ydtype == 0 || p.ydtype == p.dtype || (p.dtype == STL_BF16 && p.ydtype == STL_F16), "upsample_bwd: ydtype %d does not go with dtype %d", p.ydtype, p.dtype);
    STL_CHECK(total * 8 < (1ull << 31) && ((size_t)p.B * (p.H << p.shift) * (p.W << p.shift) * p.C) < (1ull << 32),
              "upsample_bwd: tensors of 2^31 or more elements are not supported (32-bit index arithmetic)");
#define STL_UP(TT, TYY)                                                                                         \
    do {                                                                                                        \
        if (p.shift == 1) STL_LAUNCH((upsample_bwd_kernel<TT, TYY, 1>), dim3(nb), dim3(bd), lds, ST, p);        \
        else if (p.shift == 2) STL_LAUNCH((upsample_bwd_kernel<TT, TYY, 2>), dim3(nb), dim3(bd), lds, ST, p);   \
        else if (p.shift == 3) STL_LAUNCH((upsample_bwd_kernel<TT, TYY, 3>), dim3(nb), dim3(bd), lds, ST, p);   \
        else return stl_set_error("upsample_bwd: shift %d has no instantiation (1 .. 3)", p.shift);             \
    } while (0)
    if (p.dtype == STL_BF16 && p.ydtype == STL_F16)
        STL_UP(__bf16, f16);
    else if (p.dtype == STL_BF16)
        STL_UP(__bf16, __bf16);
    else
        STL_UP(float, float);
#undef STL_UP
    STL_LAUNCH_CHECK("upsample_backward");
    return 0;
}

extern "C" int stl_patch3x3(int dtype, const float* img, void* out, int B, int H, int W, int stride, const float* mean3,
                            const float* std3, void* stream) {
    STL_CHECK(stride == 1 || stride == 2, "patch3x3: stride");
    STL_CHECK(img && out && B > 0 && H > 0 && W > 0, "patch3x3: bad args");
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const size_t total = (size_t)B * Ho * Wo * 4;
    if (dtype == STL_BF16)
        STL_LAUNCH(patch_kernel<__bf16>, dim3(nblocks_for(total, 256, 4096)), dim3(256), 0, ST, img, out, B, H, W, Ho, Wo, stride, mean3, std3);
    else if (dtype == STL_F16)
        STL_LAUNCH(patch_kernel<f16>, dim3(nblocks_for(total, 256, 4096)), dim3(256), 0, ST, img, out, B, H, W, Ho, Wo, stride, mean3, std3);
    else
        STL_LAUNCH(patch_kernel<float>, dim3(nblocks_for(total, 256, 4096)), dim3(256), 0, ST, img, out, B, H, W, Ho, Wo, stride, mean3, std3);
    STL_LAUNCH_CHECK("patch3x3");
    return 0;
}

extern "C" int stl_head_forward(int dtype, const void* x, const float* w, const float* bias, float* out, int B, int H, int W,
                                int Ci, int J, void* stream) {
    STL_CHECK(J == 17 || J == 16, "head: J=%d unsupported (16 or 17)", J);
    STL_CHECK(Ci % 8 == 0 && Ci <= 512, "head: Ci=%d", Ci);
    const size_t P = (size_t)B * H * W;
    const size_t lds = (size_t)J * Ci * 4;
    const dim3 grid(nblocks_for(P));
#define HF(T, JJ) STL_LAUNCH((head_fwd_kernel<T, JJ>), grid, dim3(256), lds, ST, x, w, bias, out, B, H * W, Ci)
    if (dtype == STL_BF16) { if (J == 17) HF(__bf16, 17); else HF(__bf16, 16); }
    else if (dtype == STL_F16) { if (J == 17) HF(f16, 17); else HF(f16, 16); }
    else { if (J == 17) HF(float, 17); else HF(float, 16); }
#undef HF
    STL_LAUNCH_CHECK("head_forward");
    return 0;
}

extern "C" int stl_head_backward(int dtype2, const void* x, const float* w, const float* dout, void* dx, float* partial,
                                 int nblk, int B, int H, int W, int Ci, int J, void* stream) {
    const int dtype = dtype2 & 0xff, ydtype = (dtype2 >> 8) & 0xff;   // STL_DT2(type of dx, type of x)
    STL_CHECK(ydtype == 0 || ydtype == dtype || (dtype == STL_BF16 && ydtype == STL_F16), "head_bwd: x type %d does not go with dx type %d", ydtype, dtype);
    STL_CHECK(J == 17 || J == 16, "head_bwd: J=%d unsupported", J);
    STL_CHECK(Ci % 8 == 0 && J * Ci + J <= 1024, "head_bwd: Ci=%d", Ci);
    STL_CHECK(nblk >= 1, "head_bwd: nblk");
    STL_CHECK(Ci % 16 == 0 && Ci <= 64, "head_bwd: Ci=%d must be a multiple of 16, at most 64", Ci);
    size_t lds = (size_t)(((J * Ci + 3) & ~3) + J * 260 + 256 * (Ci + 1)) * 4;
    const size_t lds_red = (size_t)(((J * Ci + 3) & ~3) + J * 260) * 4 + (size_t)4 * 2 * 4 * 64 * 16 + 4 * J * 4;   // reduction buffers alias sx
    if (lds_red > lds) lds = lds_red;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel<__bf16, 17>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel<__bf16, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel<float, 17>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel<float, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel<__bf16, 17, f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_kernel<__bf16, 16, f16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
#define HB(T, JJ, TY) STL_LAUNCH((head_bwd_kernel<T, JJ, TY>), dim3(nblk), dim3(256), lds, ST, x, w, dout, dx, partial, B, H * W, Ci)
    if (dtype == STL_BF16 && ydtype == STL_F16) { if (J == 17) HB(__bf16, 17, f16); else HB(__bf16, 16, f16); }
    else if (dtype == STL_BF16) { if (J == 17) HB(__bf16, 17, __bf16); else HB(__bf16, 16, __bf16); }
    else { if (J == 17) HB(float, 17, float); else HB(float, 16, float); }
#undef HB
    STL_LAUNCH_CHECK("head_backward");
    return 0;
}

extern "C" int stl_sum_partials(const double* partial, int n, double scale, float* out, int accumulate, void* stream) {
    STL_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, ST, partial, n, scale, out, accumulate);
    STL_LAUNCH_CHECK("sum_partials");
    return 0;
}

extern "C" int stl_mse_loss(const float* out, const float* target, const float* tweight, float* dout, double* partial,
                            int nblk, float* loss, int B, int J, int HW, float gscale, void* stream) {
    STL_CHECK(out && target && tweight && partial && nblk >= 1, "mse: null pointer");
    const size_t n = (size_t)B * J * HW;
    STL_LAUNCH(mse_kernel, dim3(nblk), dim3(256), 0, ST, out, target, tweight, dout, partial, n, HW, gscale / (float)n);
    STL_LAUNCH_CHECK("mse_loss");
    if (!loss) return 0;   // the caller sums the partials later: stl_sum_partials(partial, nblk, 0.5 / (B * J * HW), loss, 0, stream)
    return stl_sum_partials(partial, nblk, 0.5 / (double)n, loss, 0, stream);
}

extern "C" int stl_heatmap_argmax(const float* hm, int32_t* idx, float* maxval, float* preds, int BJ, int H, int W, void* stream) {
    if (BJ == 0) return 0;
    STL_CHECK(hm && maxval && preds && H > 0 && W > 0, "argmax: bad args");
    STL_LAUNCH(argmax_kernel, dim3(BJ), dim3(256), 0, ST, hm, idx, maxval, preds, H, W);
    STL_LAUNCH_CHECK("heatmap_argmax");
    return 0;
}

extern "C" int stl_final_preds(const float* hm, const float* center, const float* scale, float* preds, float* maxval, int B,
                               int J, int H, int W, void* stream) {
    if (B * J == 0) return 0;
    STL_LAUNCH(final_preds_kernel, dim3(B * J), dim3(256), 0, ST, hm, center, scale, preds, maxval, J, H, W);
    STL_LAUNCH_CHECK("final_preds");
    return 0;
}

extern "C" int stl_flip_merge(const float* a, const float* bflip, float* out, const int32_t* perm, int B, int J, int H, int W,
                              void* stream) {
    const size_t n = (size_t)B * J * H * W;
    if (n == 0) return 0;
    STL_LAUNCH(flip_merge_kernel, dim3(nblocks_for(n)), dim3(256), 0, ST, a, bflip, out, perm, J, H, W, n);
    STL_LAUNCH_CHECK("flip_merge");
    return 0;
}

extern "C" int stl_weight_prep(int dtype2, const float* master, void* wk, const stl_wprep* tab, int n, int nblocks, void* stream) {
    if (n == 0) return 0;
    return stl_weight_prep_range(dtype2, master, wk, tab, n, 0, nblocks, stream);
}

extern "C" int stl_weight_prep_range(int dtype2, const float* master, void* wk, const stl_wprep* tab, int n, int blk_base, int nblocks,
                                     void* stream) {
    if (n == 0 || nblocks == 0) return 0;
    STL_CHECK(master && wk && tab && n > 0 && blk_base >= 0 && nblocks > 0, "weight_prep_range: bad arguments");
    const int dtype = dtype2 & 0xff, fdtype = (dtype2 >> 8) & 0xff;   // STL_DT2(type of the data-gradient layouts, type of the forward layouts)
    STL_CHECK(fdtype == 0 || fdtype == dtype || (dtype == STL_BF16 && fdtype == STL_F16), "weight_prep: forward type %d does not go with %d", fdtype, dtype);
    if (dtype == STL_BF16 && fdtype == STL_F16)
        STL_LAUNCH((weight_prep_kernel<__bf16, f16>), dim3(nblocks), dim3(256), 0, ST, master, (__bf16*)wk, tab, n, blk_base);
    else if (dtype == STL_F16)
        STL_LAUNCH((weight_prep_kernel<f16, f16>), dim3(nblocks), dim3(256), 0, ST, master, (f16*)wk, tab, n, blk_base);
    else if (dtype == STL_BF16)
        STL_LAUNCH(weight_prep_kernel<__bf16>, dim3(nblocks), dim3(256), 0, ST, master, (__bf16*)wk, tab, n, blk_base);
    else
        STL_LAUNCH(weight_prep_kernel<float>, dim3(nblocks), dim3(256), 0, ST, master, (float*)wk, tab, n, blk_base);
    STL_LAUNCH_CHECK("weight_prep_range");
    return 0;
}

extern "C" int stl_reduce_slabs(const float* partials, float* grads, const stl_slab* tab, int n, int nblocks, void* stream) {
    if (n == 0) return 0;
    STL_LAUNCH(reduce_slabs_kernel, dim3(nblocks), dim3(256), 0, ST, partials, grads, tab, n, 0);
    STL_LAUNCH_CHECK("reduce_slabs");
    return 0;
}

extern "C" int stl_reduce_slabs_range(const stl_reduce_range* r, void* stream) {
    STL_CHECK(r && r->partials && r->grads && r->tab && r->n >= 0 && r->nblocks >= 0, "reduce_slabs_range: bad arguments");
    if (r->n == 0 || r->nblocks == 0) return 0;
    STL_LAUNCH(reduce_slabs_kernel, dim3(r->nblocks), dim3(256), 0, ST, r->partials, r->grads, r->tab, r->n, r->blk_base);
    STL_LAUNCH_CHECK("reduce_slabs_range");
    return 0;
}

extern "C" int stl_bn_running_update(const double* stats, float* buffers, int64_t* nbt, const stl_bnrec* tab, int n,
                                     float momentum, int32_t* overflow, void* stream) {
    if (n == 0) return 0;
    STL_LAUNCH(bn_running_kernel, dim3(n), dim3(256), 0, ST, stats, buffers, nbt, tab, momentum, overflow);
    STL_LAUNCH_CHECK("bn_running_update");
    return 0;
}

extern "C" int stl_bn_param_grads(const double* rstats, float* grads, const stl_bnrec* tab, int n, void* stream) {
    if (n == 0) return 0;
    STL_LAUNCH(bn_param_grads_kernel, dim3(n), dim3(256), 0, ST, rstats, grads, tab);
    STL_LAUNCH_CHECK("bn_param_grads");
    return 0;
}

extern "C" int stl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, int32_t* step, const int32_t* overflow,
                             void* stream) {
    STL_LAUNCH(inc_step_kernel, dim3(1), dim3(1), 0, ST, step, overflow);
    STL_LAUNCH(adam_kernel, dim3(nblocks_for((size_t)n, 256, 4096)), dim3(256), 0, ST, p, g, m, v, n, hyper, step);
    STL_LAUNCH_CHECK("adam_step");
    return 0;
}

extern "C" int stl_sgd_step(float* p, const float* g, float* mom, int64_t n, const float* hyper, int32_t* step, const int32_t* overflow, void* stream) {
    STL_LAUNCH(inc_step_kernel, dim3(1), dim3(1), 0, ST, step, overflow);
    STL_LAUNCH(sgd_kernel, dim3(nblocks_for((size_t)n, 256, 4096)), dim3(256), 0, ST, p, g, mom, n, hyper, step);
    STL_LAUNCH_CHECK("sgd_step");
    return 0;
}

extern "C" int stl_optim_begin_step(int32_t* step, const int32_t* overflow, void* stream) {
    STL_LAUNCH(inc_step_kernel, dim3(1), dim3(1), 0, ST, step, overflow);
    STL_LAUNCH_CHECK("optim_begin_step");
    return 0;
}

extern "C" int stl_adam_slice(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, const int32_t* step, void* stream) {
    if (n <= 0) return 0;
    STL_LAUNCH(adam_kernel, dim3(nblocks_for((size_t)n, 256, 4096)), dim3(256), 0, ST, p, g, m, v, n, hyper, step);
    STL_LAUNCH_CHECK("adam_slice");
    return 0;
}

extern "C" int stl_sgd_slice(float* p, const float* g, float* mom, int64_t n, const float* hyper, const int32_t* step, void* stream) {
    if (n <= 0) return 0;
    STL_LAUNCH(sgd_kernel, dim3(nblocks_for((size_t)n, 256, 4096)), dim3(256), 0, ST, p, g, mom, n, hyper, step);
    STL_LAUNCH_CHECK("sgd_slice");
    return 0;
}

// ---------------------------------------------------------------- gaussian heatmap targets
// One thread per heatmap pixel; mirrors data/JointsDataset.py:230-286 (generate_target).
__global__ __launch_bounds__(256) void gaussian_targets_kernel(const float* __restrict__ joints, const float* __restrict__ vis,
                                                               float* __restrict__ target, float* __restrict__ tw, int BJ, int Hh,
                                                               int Wh, double sx, double sy, float sigma) {
    const size_t total = (size_t)BJ * Hh * Wh;
    const float r = sigma * 3.f;
    const float inv2s2 = 1.f / (2.f * sigma * sigma);
    const int c = (int)(2.f * r + 1.f) / 2;  // size // 2
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wh);
        const size_t t = i / Wh;
        const int y = (int)(t % Hh), k = (int)(t / Hh);
        const int mx = (int)((double)joints[2 * k] / sx + 0.5), my = (int)((double)joints[2 * k + 1] / sy + 0.5);
        const int ulx = (int)((float)mx - r), uly = (int)((float)my - r);
        const int brx = (int)((float)mx + r + 1.f), bry = (int)((float)my + r + 1.f);
        const bool inb = !(ulx >= Wh || uly >= Hh || brx < 0 || bry < 0);
        const float v = vis[k];
        float val = 0.f;
        if (inb && v > 0.5f && x >= max(0, ulx) && x < min(brx, Wh) && y >= max(0, uly) && y < min(bry, Hh)) {
            const float dx = (float)(x - ulx - c), dy = (float)(y - uly - c);
            val = expf(-(dx * dx + dy * dy) * inv2s2);
        }
        target[i] = val;
        if (x == 0 && y == 0) tw[k] = inb ? v : 0.f;
    }
}

extern "C" int stl_gaussian_targets(const float* joints_xy, const float* vis, float* target, float* tweight, int B, int J, int Hh,
                                    int Wh, float stride_x, float stride_y, float sigma, void* stream) {
    STL_CHECK(joints_xy && vis && target && tweight && B > 0 && J > 0 && Hh > 0 && Wh > 0, "gaussian_targets: bad arguments");
    STL_CHECK(sigma > 0.f && stride_x > 0.f && stride_y > 0.f, "gaussian_targets: sigma / stride must be positive");
    const size_t total = (size_t)B * J * Hh * Wh;
    STL_LAUNCH(gaussian_targets_kernel, dim3(nblocks_for(total, 256, 4096)), dim3(256), 0, ST, joints_xy, vis, target, tweight,
                       B * J, Hh, Wh, (double)stride_x, (double)stride_y, sigma);
    STL_LAUNCH_CHECK("gaussian_targets");
    return 0;
}

extern "C" int stl_affine_crop(const uint8_t* src, const int64_t* src_off, const int32_t* src_hw, const float* minv, const int32_t* flip,
                               float* out, int B, int Ho, int Wo, const float* mean3, const float* std3, void* stream) {
    if (B == 0) return 0;
    STL_CHECK(src && src_off && src_hw && minv && out && Ho > 0 && Wo > 0, "affine_crop: bad arguments");
    STL_CHECK((mean3 == nullptr) == (std3 == nullptr), "affine_crop: mean and std go together");
    const int gx = (Ho * Wo + 255) / 256;
    STL_LAUNCH(affine_crop_kernel, dim3(gx < 64 ? gx : 64, B), dim3(256), 0, ST, src, src_off, src_hw, minv, flip, out, Ho, Wo, mean3, std3);
    STL_LAUNCH_CHECK("affine_crop");
    return 0;
}

extern "C" int stl_maxpool2x2(int dtype, const void* x, void* out, int B, int H, int W, int C, void* stream) {
    STL_CHECK(C % 8 == 0 && H >= 2 && W >= 2, "maxpool: C%%8 == 0 and H, W >= 2 required");  // odd H/W: floor, like nn.MaxPool2d
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 8);
    if (dtype == STL_BF16)
        STL_LAUNCH(maxpool_kernel<__bf16>, dim3(nblocks_for(total, 256, 4096)), dim3(256), 0, ST, x, out, B, H, W, C);
    else
        STL_LAUNCH(maxpool_kernel<float>, dim3(nblocks_for(total, 256, 4096)), dim3(256), 0, ST, x, out, B, H, W, C);
    STL_LAUNCH_CHECK("maxpool2x2");
    return 0;
}

extern "C" int stl_l1_partial(int dtype, const void* a, const void* b, int64_t n, double* partial, int nblk, void* stream) {
    STL_CHECK(n % 8 == 0 && nblk >= 1, "l1: n%%8");
    if (dtype == STL_BF16)
        STL_LAUNCH((l1_kernel<__bf16, false>), dim3(nblk), dim3(256), 0, ST, a, b, (size_t)n / 8, partial);
    else
        STL_LAUNCH((l1_kernel<float, false>), dim3(nblk), dim3(256), 0, ST, a, b, (size_t)n / 8, partial);
    STL_LAUNCH_CHECK("l1_partial");
    return 0;
}

extern "C" int stl_l2_partial(int dtype, const void* a, const void* b, int64_t n, double* partial, int nblk, void* stream) {
    STL_CHECK(n % 8 == 0 && nblk >= 1, "l2: n%%8");
    if (dtype == STL_BF16)
        STL_LAUNCH((l1_kernel<__bf16, true>), dim3(nblk), dim3(256), 0, ST, a, b, (size_t)n / 8, partial);
    else
        STL_LAUNCH((l1_kernel<float, true>), dim3(nblk), dim3(256), 0, ST, a, b, (size_t)n / 8, partial);
    STL_LAUNCH_CHECK("l2_partial");
    return 0;
}

extern "C" int stl_bilinear_nchw(const float* in, float* out, int B, int C, int H, int W, int Ho, int Wo, void* stream) {
    const size_t total = (size_t)B * C * Ho * Wo;
    STL_LAUNCH(bilinear_kernel, dim3(nblocks_for(total, 256, 4096)), dim3(256), 0, ST, in, out, B * C, H, W, Ho, Wo);
    STL_LAUNCH_CHECK("bilinear");
    return 0;
}

extern "C" int stl_nchw_to_nhwc(int dtype, const float* in, void* out, int B, int C, int H, int W, void* stream) {
    const size_t total = (size_t)B * C * H * W;
    if (dtype == STL_BF16)
        STL_LAUNCH(nchw_to_nhwc_kernel<__bf16>, dim3(nblocks_for(total, 256, 4096)), dim3(256), 0, ST, in, (__bf16*)out, B, C, H * W);
    else
        STL_LAUNCH(nchw_to_nhwc_kernel<float>, dim3(nblocks_for(total, 256, 4096)), dim3(256), 0, ST, in, (float*)out, B, C, H * W);
    STL_LAUNCH_CHECK("nchw_to_nhwc");
    return 0;
}

extern "C" int stl_nhwc_to_nchw(int dtype, const void* in, float* out, int B, int C, int H, int W, void* stream) {
    const size_t total = (size_t)B * C * H * W;
    if (dtype == STL_BF16)
        STL_LAUNCH(nhwc_to_nchw_kernel<__bf16>, dim3(nblocks_for(total, 256, 4096)), dim3(256), 0, ST, (const __bf16*)in, out, B, C, H * W);
    else
        STL_LAUNCH(nhwc_to_nchw_kernel<float>, dim3(nblocks_for(total, 256, 4096)), dim3(256), 0, ST, (const float*)in, out, B, C, H * W);
    STL_LAUNCH_CHECK("nhwc_to_nchw");
    return 0;
}

extern "C" int stl_selftest_mfma(float* out, void* stream) {
    double* d = nullptr;
    if (hipMalloc(&d, 16) != hipSuccess) return stl_set_error("selftest: hipMalloc failed");
    hipMemsetAsync(d, 0, 16, ST);
    STL_LAUNCH(selftest_kernel, dim3(1), dim3(64), 0, ST, out);
    STL_LAUNCH(selftest_atomic_kernel, dim3(1), dim3(256), 0, ST, d);
    STL_LAUNCH(selftest_fin_kernel, dim3(1), dim3(1), 0, ST, d, out);
    hipStreamSynchronize(ST);
    hipFree(d);
    STL_LAUNCH_CHECK("selftest");
    return 0;
}
