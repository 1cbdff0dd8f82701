// Shared device helpers for the gfx950 kernels (wave64, MFMA 16x16 tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <initializer_list>
#include <tuple>
#include <type_traits>
#include <utility>
#include "../../include/stlpose_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef _Float16 f16;   // STL_F16: forward activations of the mixed 16-bit mode (10 mantissa bits; gradients stay bf16 for range)

struct alignas(16) V16 {
    uint32_t w[4];
};

int stl_set_error(const char* fmt, ...);
// Name of the kernel instantiation the calling thread launched last (stl_last_kernel(): bench.py groups its per-launch
// timings by it, so the roofline entry names what rocprofv3's kernel trace names).  Templated launchers note their
// instantiation ("conv_core_kernel<bf16,3,4,2,4,2,3,1,0,1,-1,0>": the template arguments in declaration order); every
// other launch site is named by its STL_LAUNCH_CHECK string.
void stl_note_kernel(const char* name, bool specific);
#define STL_CHECK(cond, ...)                 \
    do {                                     \
        if (!(cond)) return stl_set_error(__VA_ARGS__); \
    } while (0)
#define STL_LAUNCH_CHECK(name)                                                       \
    do {                                                                             \
        stl_note_kernel(name, false);                                                \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) return stl_set_error("%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------- element traits
template <typename T>
struct ET;
template <>
struct ET<float> {
    static constexpr int KV = 4;   // elements per 16 bytes
    static constexpr int CK = 16;  // channels per 64-byte K chunk
};
template <>
struct ET<__bf16> {
    static constexpr int KV = 8;
    static constexpr int CK = 32;
};
template <>
struct ET<f16> {
    static constexpr int KV = 8;
    static constexpr int CK = 32;
};
// dtype code (STL_F32 / STL_BF16 / STL_F16) of an element type
template <typename T>
constexpr int stl_code() { return sizeof(T) == 4 ? STL_F32 : (std::is_same<T, __bf16>::value ? STL_BF16 : STL_F16); }

__device__ __forceinline__ float bf16_to_f32(uint32_t bits16) { return __uint_as_float(bits16 << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
    return (uint32_t)__builtin_bit_cast(unsigned short, h);
}

// two 16-bit values of one dword <-> two floats (T = __bf16 or f16): every 16-bit conversion of the kernels goes through these
template <typename T>
__device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
    if constexpr (std::is_same<T, __bf16>::value) {
        lo = __uint_as_float(w << 16), hi = __uint_as_float(w & 0xFFFF0000u);
    } else {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        const h2 h = __builtin_bit_cast(h2, w);
        lo = (float)h[0], hi = (float)h[1];
    }
}
template <typename T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {   // round to nearest even, NaN-preserving
    if constexpr (std::is_same<T, __bf16>::value) {
        typedef __attribute__((ext_vector_type(2))) __bf16 b2;
        b2 r;
        r[0] = (__bf16)lo, r[1] = (__bf16)hi;   // v_cvt_pk_bf16_f32
        return __builtin_bit_cast(uint32_t, r);
    } else {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        h2 r;
        r[0] = (_Float16)lo, r[1] = (_Float16)hi;
        return __builtin_bit_cast(uint32_t, r);
    }
}

template <typename T>
__device__ __forceinline__ void unpack(const V16& v, float* f);
template <>
__device__ __forceinline__ void unpack<float>(const V16& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v.w[i]);
}
template <>
__device__ __forceinline__ void unpack<__bf16>(const V16& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2<__bf16>(v.w[i], f[2 * i], f[2 * i + 1]);
}
template <>
__device__ __forceinline__ void unpack<f16>(const V16& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2<f16>(v.w[i], f[2 * i], f[2 * i + 1]);
}
template <typename T>
__device__ __forceinline__ V16 pack(const float* f);
template <>
__device__ __forceinline__ V16 pack<float>(const float* f) {
    V16 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v.w[i] = __float_as_uint(f[i]);
    return v;
}
template <>
__device__ __forceinline__ V16 pack<__bf16>(const float* f) {
    V16 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v.w[i] = pack2<__bf16>(f[2 * i], f[2 * i + 1]);
    return v;
}
template <>
__device__ __forceinline__ V16 pack<f16>(const float* f) {
    V16 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v.w[i] = pack2<f16>(f[2 * i], f[2 * i + 1]);
    return v;
}
// value as stored (rounded to T) -- so statistics see exactly what consumers will read
template <typename T>
__device__ __forceinline__ float round_to(float f);
template <>
__device__ __forceinline__ float round_to<float>(float f) { return f; }
template <>
__device__ __forceinline__ float round_to<__bf16>(float f) { return bf16_to_f32(f32_to_bf16(f)); }
template <>
__device__ __forceinline__ float round_to<f16>(float f) { return (float)(_Float16)f; }

__device__ __forceinline__ V16 zero16() {
    V16 v;
    v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0u;
    return v;
}
__device__ __forceinline__ V16 ldg16(const void* p) { return *reinterpret_cast<const V16*>(p); }
__device__ __forceinline__ void stg16(void* p, const V16& v) { *reinterpret_cast<V16*>(p) = v; }

// ---------------------------------------------------------------- MFMA 16x16 step over one 64-byte K slice
// A fragment: lane (row = lane&15, g = lane>>4) holds the KV consecutive K elements g*KV..g*KV+KV-1
// B fragment: lane (col = lane&15, g) holds the same K elements.  C/D: col = lane&15, row = 4*g + reg.
template <typename T>
__device__ __forceinline__ void mma16(f32x4& acc, const V16& a, const V16& b);
template <>
__device__ __forceinline__ void mma16<__bf16>(f32x4& acc, const V16& a, const V16& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma16<f16>(f32x4& acc, const V16& a, const V16& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma16<float>(f32x4& acc, const V16& a, const V16& b) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w[s]), __uint_as_float(b.w[s]), acc, 0, 0, 0);
}

// 1/sqrt(x) in fp32: hardware estimate + one Newton step (full fp32 accuracy, ~6 instructions; the
// fp64 sqrt + divide it replaces costs ~60 in every kernel prologue).  The variance itself is still
// formed in fp64 (sum of squares minus squared mean cancels).
__device__ __forceinline__ float rsqrt_nr(float x) {
#pragma clang fp contract(off)   // (the same roundings wherever it is inlined: see bn_affine)
    const float r = __frsqrt_rn(x);
    const float h = 0.5f * x * r;
    return r * fmaf(-h, r, 1.5f);
}

// a, b of y -> a * y + b for a BatchNorm given mean and 1 / std (the ReLU-mask BatchNorm of a data gradient's epilogue)
__device__ __forceinline__ void bn_affine(float gamma, float beta, float mean, float rstd, float& a, float& b) {
#pragma clang fp contract(off)   // (as below)
    a = gamma * rstd;
    b = fmaf(-mean, a, beta);   // one rounding, written out (what the contraction gave where it happened)
}

// ---------------------------------------------------------------- per-channel constants of an stl_src
// ca/cb/cc: v = ca*x + cb (BN) or v = ca*dt + cb*y + cc (BNBWD).  mu/rs: mean and 1/std (for yhat).
__device__ __forceinline__ void bn_mean_rstd(const stl_src& s, int c, int C, float& mean, float& rstd) {
#pragma clang fp contract(off)   // per-channel constants: the same roundings in every kernel that derives them (fused multiply-adds are written out)
    if (s.stats) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < STL_NSHARD; ++k) {
            s0 += s.stats[(size_t)k * 2 * C + c];
            s1 += s.stats[(size_t)k * 2 * C + C + c];
        }
        double m = s0 * (double)s.inv_count;
        double var = fma(s1, (double)s.inv_count, -(m * m));
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        rstd = rsqrt_nr((float)(var + (double)s.eps));
    } else {
        mean = s.rmean[c];
        rstd = rsqrt_nr(s.rvar[c] + s.eps);
    }
}

__device__ __forceinline__ void src_consts(const stl_src& s, int c, int C, float& ca, float& cb, float& cc) {
#pragma clang fp contract(off)   // per-channel constants: the same roundings in every kernel that derives them (fused multiply-adds are written out)
    if (s.mode == STL_SRC_PLAIN) {
        ca = 1.f, cb = 0.f, cc = 0.f;
        return;
    }
    float mean, rstd;
    bn_mean_rstd(s, c, C, mean, rstd);
    const float g = s.gamma[c];
    if (s.mode == STL_SRC_BN || s.mode == STL_SRC_BNADD) {
        ca = g * rstd;
        cb = fmaf(-mean, ca, s.beta[c]);
        cc = 0.f;
    } else {  // BNBWD
        double r1 = 0.0, r2 = 0.0;
#pragma unroll
        for (int k = 0; k < STL_NSHARD; ++k) {
            r1 += s.rstats[(size_t)k * 2 * C + c];
            r2 += s.rstats[(size_t)k * 2 * C + C + c];
        }
        const float c1 = (float)(r1 * (double)s.inv_count);
        const float c2 = (float)(r2 * (double)s.inv_count);
        const float al = g * rstd;
        ca = al;
        cb = -al * rstd * c2;
        cc = al * fmaf(mean * rstd, c2, -c1);
    }
}

// floor(m / d) for 0 <= m, m * d < 2^21, with r = 1/d: exact, ~4 instructions instead of ~40
__device__ __forceinline__ int fdiv(int m, float r) { return (int)(((float)m + 0.5f) * r); }

// Two-phase variants: *_raw_load only ISSUES the loads of one channel's statistics (so that they
// queue ahead of a burst of tile loads), *_raw_finish does the arithmetic of bn_mean_rstd / src_consts.
struct SrcRaw {
    double st[2 * STL_NSHARD], rs[2 * STL_NSHARD];
    float g, b, rm, rv;
};
__device__ __forceinline__ void bn_raw_load(const stl_src& s, int c, int C, SrcRaw& r) {
    r.g = s.gamma[c];
    if (s.stats) {
#pragma unroll
        for (int k = 0; k < STL_NSHARD; ++k) {
            r.st[2 * k] = s.stats[(size_t)k * 2 * C + c];
            r.st[2 * k + 1] = s.stats[(size_t)k * 2 * C + C + c];
        }
    } else {
        r.rm = s.rmean[c], r.rv = s.rvar[c];
    }
}
__device__ __forceinline__ void bn_raw_finish(const stl_src& s, const SrcRaw& r, float& mean, float& rstd) {
#pragma clang fp contract(off)   // per-channel constants: the same roundings in every kernel that derives them (fused multiply-adds are written out)
    if (s.stats) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < STL_NSHARD; ++k) s0 += r.st[2 * k], s1 += r.st[2 * k + 1];
        double m = s0 * (double)s.inv_count;
        double var = fma(s1, (double)s.inv_count, -(m * m));
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        rstd = rsqrt_nr((float)(var + (double)s.eps));
    } else {
        mean = r.rm;
        rstd = rsqrt_nr(r.rv + s.eps);
    }
}
__device__ __forceinline__ void src_raw_load(const stl_src& s, int c, int C, SrcRaw& r) {
    if (s.mode == STL_SRC_PLAIN) return;
    bn_raw_load(s, c, C, r);
    if (s.mode == STL_SRC_BN || s.mode == STL_SRC_BNADD) {
        r.b = s.beta[c];
    } else {
#pragma unroll
        for (int k = 0; k < STL_NSHARD; ++k) {
            r.rs[2 * k] = s.rstats[(size_t)k * 2 * C + c];
            r.rs[2 * k + 1] = s.rstats[(size_t)k * 2 * C + C + c];
        }
    }
}
__device__ __forceinline__ void src_raw_finish(const stl_src& s, const SrcRaw& r, float& ca, float& cb, float& cc) {
#pragma clang fp contract(off)   // per-channel constants: the same roundings in every kernel that derives them (fused multiply-adds are written out)
    if (s.mode == STL_SRC_PLAIN) {
        ca = 1.f, cb = 0.f, cc = 0.f;
        return;
    }
    float mean, rstd;
    bn_raw_finish(s, r, mean, rstd);
    if (s.mode == STL_SRC_BN || s.mode == STL_SRC_BNADD) {
        ca = r.g * rstd;
        cb = fmaf(-mean, ca, r.b);
        cc = 0.f;
    } else {
        double r1 = 0.0, r2 = 0.0;
#pragma unroll
        for (int k = 0; k < STL_NSHARD; ++k) r1 += r.rs[2 * k], r2 += r.rs[2 * k + 1];
        const float c1 = (float)(r1 * (double)s.inv_count);
        const float c2 = (float)(r2 * (double)s.inv_count);
        const float al = r.g * rstd;
        ca = al;
        cb = -al * rstd * c2;
        cc = al * fmaf(mean * rstd, c2, -c1);
    }
}

// sum over the 16 lanes of a DPP row (every lane gets the total): 4 VALU ops, no LDS traffic
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm 1,0,3,2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm 2,3,0,1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
    return v;
}

// block-wide helpers ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- launches: issued, or recorded for an explicit HIP graph
// Every kernel of the library is launched through STL_LAUNCH.  Normally that is hipLaunchKernel on the caller's stream.  While a
// program is being turned into a hipGraph (program.hip: stl_program_graph_build) a thread-local recorder is set and the launch
// is RECORDED instead -- function, grid, block, LDS bytes and a private copy of the arguments -- to become a kernel node.
struct StlLaunchRec {
    const void* func;
    dim3 grid, block;
    size_t lds;
    void* args_owner;                 // heap copy of the argument tuple (kept alive by the graph's owner)
    void (*del)(void*);
    void** params;                    // one pointer per kernel argument, into args_owner
    int nparams;
};
struct StlRecorder { StlLaunchRec* recs; int n, cap; };
extern thread_local StlRecorder* g_stl_recorder;
void stl_recorder_push(const StlLaunchRec& r);

template <typename Tup, size_t... I>
static inline void stl_tuple_ptrs(Tup& t, void** out, std::index_sequence<I...>) { ((out[I] = (void*)&std::get<I>(t)), ...); }

template <typename... KArgs, typename... Args>
static inline void stl_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
    using Tup = std::tuple<std::decay_t<KArgs>...>;
    if (g_stl_recorder) {
        Tup* t = new Tup(static_cast<std::decay_t<KArgs>>(args)...);
        void** pp = new void*[sizeof...(KArgs) ? sizeof...(KArgs) : 1];
        stl_tuple_ptrs(*t, pp, std::index_sequence_for<KArgs...>{});
        stl_recorder_push(StlLaunchRec{(const void*)kernel, grid, block, lds, t, [](void* q) { delete static_cast<Tup*>(q); }, pp, (int)sizeof...(KArgs)});
        return;
    }
    Tup t(static_cast<std::decay_t<KArgs>>(args)...);
    void* pp[sizeof...(KArgs) ? sizeof...(KArgs) : 1];
    stl_tuple_ptrs(t, pp, std::index_sequence_for<KArgs...>{});
    (void)hipLaunchKernel((const void*)kernel, grid, block, pp, lds, st);
}
#define STL_LAUNCH(kernel, grid, block, lds, st, ...) stl_launch(kernel, grid, block, lds, st, ##__VA_ARGS__)

// "base<bf16,a0,a1,...>" for stl_note_kernel (built once per instantiation: function-local static in the launcher)
template <typename T>
static inline const char* stl_kname(char (&buf)[160], const char* base, std::initializer_list<int> args) {
    int n = snprintf(buf, sizeof(buf), "%s<%s", base, sizeof(T) == 4 ? "f32" : (std::is_same<T, __bf16>::value ? "bf16" : "f16"));
    for (int a : args) n += snprintf(buf + n, sizeof(buf) - n, ",%d", a);
    snprintf(buf + n, sizeof(buf) - n, ">");
    return buf;
}
