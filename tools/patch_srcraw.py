p='/root/repo/stlpose_amd/csrc/common.cuh'
s=open(p).read()
a='// block-wide helpers ------'
assert s.count(a)==1
new='''// Two-phase variant: src_raw_load only ISSUES the loads of one channel's statistics (so that they
// queue ahead of a burst of tile loads), src_raw_finish does the arithmetic of src_consts.
struct SrcRaw {
    double st[2 * STL_NSHARD], rs[2 * STL_NSHARD];
    float g, b, rm, rv;
};
__device__ __forceinline__ void src_raw_load(const stl_src& s, int c, int C, SrcRaw& r) {
    if (s.mode == STL_SRC_PLAIN) return;
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
    if (s.mode == STL_SRC_BN) {
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
    if (s.mode == STL_SRC_PLAIN) {
        ca = 1.f, cb = 0.f, cc = 0.f;
        return;
    }
    float mean, rstd;
    if (s.stats) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < STL_NSHARD; ++k) s0 += r.st[2 * k], s1 += r.st[2 * k + 1];
        double m = s0 * (double)s.inv_count;
        double var = s1 * (double)s.inv_count - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)s.eps));
    } else {
        mean = r.rm;
        rstd = (float)(1.0 / sqrt((double)r.rv + (double)s.eps));
    }
    if (s.mode == STL_SRC_BN) {
        ca = r.g * rstd;
        cb = r.b - mean * ca;
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
        cc = al * (mean * rstd * c2 - c1);
    }
}

'''
s=s.replace(a,new+a)
open(p,'w').write(s)

p='/root/repo/stlpose_amd/csrc/wgrad.hip'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, (s.count(a), a)
    s=s.replace(a,b)
rep('''    WSTAMP(1);
    if (have) setup(t);
    issue(have);
    WSTAMP(2);
    // BatchNorm constants: wave 3, after its loads are in flight
    if (tid >= 192 && tid < 224) {
        const int c = tid - 192;
        float a = 0.f, b = 0.f, cc = 0.f;
        if (co0 + c < p.Co) src_consts(p.g, co0 + c, p.Co, a, b, cc);
        cgc[c] = a, cgc[32 + c] = b, cgc[64 + c] = cc;
    } else if (tid >= 224) {
        const int c = tid - 224;
        float a = 0.f, b = 0.f, cc = 0.f;
        if (ci0 + c < p.Ci) src_consts(p.h, ci0 + c, p.Ci, a, b, cc);
        chc[c] = a, chc[32 + c] = b;
    }
    __syncthreads();  // constants visible''','''    WSTAMP(1);
    // BatchNorm constants (wave 3: lanes 0-31 those of g, 32-63 those of h): the statistics loads are
    // issued ahead of the first tile's loads, the arithmetic runs while those are in flight
    SrcRaw raw;
    const bool cw = wave == 3, cg = lane < 32;
    const int cch = lane & 31;
    const bool cok = cw && (cg ? co0 + cch < p.Co : ci0 + cch < p.Ci);
    if (cok) {
        if (cg) src_raw_load(p.g, co0 + cch, p.Co, raw);
        else src_raw_load(p.h, ci0 + cch, p.Ci, raw);
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
        if (cg) cgc[cch] = a, cgc[32 + cch] = b, cgc[64 + cch] = cc;
        else chc[cch] = a, chc[32 + cch] = b;
    }
    __syncthreads();  // constants visible''')
open(p,'w').write(s)
