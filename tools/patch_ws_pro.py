root='/root/repo/stlpose_amd/csrc/'
def rep(s,a,b,count=1):
    assert s.count(a)==count, (s.count(a), a)
    return s.replace(a,b)
s=open(root+'conv_core.hip').read()
s=rep(s,'(int64_t)vrows * 2 * k.PI < (1 << 21)','(int64_t)vrows * 2 < (1 << 21)')
open(root+'conv_core.hip','w').write(s)

s=open(root+'conv_ws.inc').read()
i0=s.index('    for (int c = tid; c < k.cipad; c += 512) {')
i1=s.index('    // uniform stage sequence (tile, chunk) of this block')
new='''    // ---- per-channel constants, phase 1: only ISSUE the statistics loads of item `tid` (items:
    // cipad source channels, then the BCO channels of the ReLU-mask BatchNorm)
    SrcRaw raw;
    const int nitem = k.cipad + (p.mask_y ? BCO : 0);
    const bool it_src = tid < k.cipad;
    const int it_c = it_src ? tid : n0 + (tid - k.cipad);
    const bool it_ok = it_src ? (tid < p.Ci) : (tid < nitem && it_c < p.Co);
    if (it_ok) {
        if (it_src) {
            src_raw_load(p.src, it_c, p.Ci, raw);
        } else {
            bn_raw_load(p.mask_bn, it_c, p.Co, raw);
            raw.b = p.mask_bn.beta[it_c];
        }
    }
    // phase 2 (called by every wave right before barrier (A); the loaders have their first tile's
    // loads in flight by then)
    auto consts_finish = [&]() {
        if (tid < nitem) {
            if (it_src) {
                float a = 0.f, b = 0.f, cc = 0.f;
                if (it_ok) src_raw_finish(p.src, raw, a, b, cc);
                cs[tid] = a, cs[k.cipad + tid] = b, cs[2 * k.cipad + tid] = cc;
            } else {
                float a = 0.f, b = 0.f, mu = 0.f, rs = 0.f;
                if (it_ok) {
                    bn_raw_finish(p.mask_bn, raw, mu, rs);
                    a = raw.g * rs;
                    b = raw.b - mu * a;
                }
                const int c = tid - k.cipad;
                cm[c] = a, cm[BCO + c] = b, cm[2 * BCO + c] = mu, cm[3 * BCO + c] = rs;
            }
        }
        for (int item = tid + 512; item < nitem; item += 512) {  // more items than threads: rare
            if (item < k.cipad) {
                float a = 0.f, b = 0.f, cc = 0.f;
                if (item < p.Ci) src_consts(p.src, item, p.Ci, a, b, cc);
                cs[item] = a, cs[k.cipad + item] = b, cs[2 * k.cipad + item] = cc;
            } else {
                const int c = item - k.cipad;
                float a = 0.f, b = 0.f, mu = 0.f, rs = 0.f;
                if (n0 + c < p.Co) {
                    bn_mean_rstd(p.mask_bn, n0 + c, p.Co, mu, rs);
                    a = p.mask_bn.gamma[n0 + c] * rs;
                    b = p.mask_bn.beta[n0 + c] - mu * a;
                }
                cm[c] = a, cm[BCO + c] = b, cm[2 * BCO + c] = mu, cm[3 * BCO + c] = rs;
            }
        }
        if (k.wres) {  // the block's whole filter slab (all K chunks) stays resident in LDS for all its tiles
            const int nv = k.nchunks * BCO * TAPS * 4;
            for (int v0 = tid; v0 < nv; v0 += 4 * 512) {  // four loads in flight per round trip
                V16 val[4];
                int dst[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = v0 + j * 512;
                    const int vc = v < nv ? v : 0;
                    const int c = vc / (BCO * TAPS * 4), vv = vc - c * (BCO * TAPS * 4);
                    const int n = vv / (TAPS * 4), r = vv - n * (TAPS * 4), tap = r >> 2, part = r & 3;
                    const int ch = c * CK + part * KV;
                    const bool ok = v < nv && n0 + n < p.Co && ch < p.Ci;
                    val[j] = ldg16((const char*)p.w + (ok ? ((size_t)((n0 + n) * TAPS + tap) * p.Ci + ch) : 0) * sizeof(T));
                    dst[j] = ok ? 1 : 0;
                    dst[j] |= (c * (BCO * ROWB) + n * ROWB + tap * 64 + part * 16) << 1;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mask16(val[j], dst[j] & 1);
                    if (v0 + j * 512 < nv) *reinterpret_cast<V16*>(sB + (dst[j] >> 1)) = val[j];
                }
            }
        }
    };

'''
s=s[:i0]+new+s[i1:]
s=rep(s,'''        // the cs[] constants are written by all threads above; loaders read them in write_lds
        __syncthreads();  // (A) constants + resident filters visible
        // stage 0 -> buffer 0, then stage 1 in flight
        int li = it, lt = t, lch = ch0;
        bool lhave = have;
        if (lhave) tile_setup(lt);
        issue(0, lhave);
        write_lds(0, 0);''','''        // stage 0's loads go out first, then the constants' arithmetic and the resident filters
        int li = it, lt = t, lch = ch0;
        bool lhave = have;
        if (lhave) tile_setup(lt);
        issue(0, lhave);
        consts_finish();
        __syncthreads();  // (A) constants + resident filters visible
        write_lds(0, 0);''')
s=rep(s,'''        __syncthreads();  // (A)
        __syncthreads();  // (B)''','''        consts_finish();
        __syncthreads();  // (A)
        __syncthreads();  // (B)''')
open(root+'conv_ws.inc','w').write(s)
print('ok')
