p='/root/repo/stlpose_amd/csrc/conv_common.inc'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, (s.count(a), a)
    s=s.replace(a,b)
rep('''// Fused epilogue of ONE 16-pixel tile of a wave''','''// bf16, two neighbouring channel tiles at once: every lane loads the 16 contiguous bytes it also
// stores (8 channels of tile ni + (g & 1)), then v_permlane16_swap (its own inverse) hands each
// lane the 4 channels of BOTH tiles that the MFMA accumulator layout gives it.  One 16-byte load
// per tile pair instead of two 8-byte ones: 64 contiguous bytes per pixel and instruction.
__device__ __forceinline__ void load4p_pair(const void* base, size_t elem, f2v (&a)[2], f2v (&b)[2]) {
    const V16 v = ldg16((const char*)base + elem * 2);
    const auto r0 = __builtin_amdgcn_permlane16_swap(v.w[0], v.w[2], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(v.w[1], v.w[3], false, false);
    const uint32_t a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1];
    a[0][0] = __uint_as_float(a0 << 16), a[0][1] = __uint_as_float(a0 & 0xFFFF0000u);
    a[1][0] = __uint_as_float(a1 << 16), a[1][1] = __uint_as_float(a1 & 0xFFFF0000u);
    b[0][0] = __uint_as_float(b0 << 16), b[0][1] = __uint_as_float(b0 & 0xFFFF0000u);
    b[1][0] = __uint_as_float(b1 << 16), b[1][1] = __uint_as_float(b1 & 0xFFFF0000u);
}

// Fused epilogue of ONE 16-pixel tile of a wave''')
rep('''    f2v ad[NTW][2], my[NTW][2], mz[NTW][2];
    if (p.mask_z) {
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) load4p<T>(p.mask_z, eov[ni], mz[ni][0], mz[ni][1]);
    }
    if (p.addend) {
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) load4p<T>(p.addend, eov[ni], ad[ni][0], ad[ni][1]);
    }
    if (p.mask_y) {
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) load4p<T>(p.mask_y, eov[ni], my[ni][0], my[ni][1]);
    }''','''    f2v ad[NTW][2], my[NTW][2], mz[NTW][2];
    if constexpr (sizeof(T) == 2 && NTW % 2 == 0) {
        // wide path: the lane's own 8-channel group per tile pair (same address as its store)
        size_t e8[NTW / 2];
#pragma unroll
        for (int nj = 0; nj < NTW / 2; ++nj) {
            const int co = n0 + cbase + (2 * nj + (g & 1)) * 16 + 8 * (g >> 1);
            e8[nj] = (pok && co < p.Co) ? pix + co : 0;  // invalid lanes read element 0; their values are masked below
        }
        if (p.mask_z) {
#pragma unroll
            for (int nj = 0; nj < NTW / 2; ++nj) load4p_pair(p.mask_z, e8[nj], mz[2 * nj], mz[2 * nj + 1]);
        }
        if (p.addend) {
#pragma unroll
            for (int nj = 0; nj < NTW / 2; ++nj) load4p_pair(p.addend, e8[nj], ad[2 * nj], ad[2 * nj + 1]);
        }
        if (p.mask_y) {
#pragma unroll
            for (int nj = 0; nj < NTW / 2; ++nj) load4p_pair(p.mask_y, e8[nj], my[2 * nj], my[2 * nj + 1]);
        }
    } else {
        if (p.mask_z) {
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni) load4p<T>(p.mask_z, eov[ni], mz[ni][0], mz[ni][1]);
        }
        if (p.addend) {
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni) load4p<T>(p.addend, eov[ni], ad[ni][0], ad[ni][1]);
        }
        if (p.mask_y) {
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni) load4p<T>(p.mask_y, eov[ni], my[ni][0], my[ni][1]);
        }
    }''')
open(p,'w').write(s)
