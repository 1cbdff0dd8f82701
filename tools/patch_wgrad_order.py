p='/root/repo/stlpose_amd/csrc/wgrad.hip'
s=open(p).read()
def rep(a,b,count=1):
    global s
    assert s.count(a)==count,(s.count(a),a)
    s=s.replace(a,b)
rep('''    float r_TW, r_HC, r_tc, r_vp, r_PI;  // reciprocals for fdiv
};''','''    float r_TW, r_HC, r_tc, r_vp, r_PI;  // reciprocals for fdiv
    int ny, nz;  // output- / input-channel blocks
};

// Block order.  The grid is 1-D: XCD = id & 7 (hardware round-robin); within an XCD consecutive
// blocks form a GROUP = the nz input-channel blocks of one (split, output-channel block).  A group
// reads the same gradient slice (dt, y) and, together, whole rows of the forward activations, so the
// gradient tensors cross the fabric once instead of nz times and every fetched line is fully used.
struct WgBlock { int split, by, bz; bool valid; };
__device__ __forceinline__ WgBlock wg_block(const WgK& k) {
    const int id = blockIdx.x, xcd = id & 7, l = id >> 3;
    const int bz = l % k.nz, gi = (l / k.nz) * 8 + xcd;   // group index over (split, by), round-robin over the XCDs
    WgBlock b;
    b.bz = bz, b.by = gi % k.ny, b.split = gi / k.ny;
    b.valid = b.split < k.p.nsplit;
    return b;
}''')
# 32x32 kernel
rep('''    const int co0 = blockIdx.y * 32, ci0 = blockIdx.z * 32;''','''    const WgBlock wb = wg_block(k);
    if (!wb.valid) return;  // padding blocks of the last group round (whole block, before any barrier)
    const int co0 = wb.by * 32, ci0 = wb.bz * 32;''')
rep('''    const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;''','''    const WgBlock wb = wg_block(k);
    if (!wb.valid) return;
    const int co0 = wb.by * 64, ci0 = wb.bz * 64;''')
rep('''    int t = blockIdx.x;''','''    int t = wb.split;''',2)
rep('''        const int tn = t + gridDim.x;''','''        const int tn = t + p.nsplit;''',2)
rep('''(size_t)blockIdx.x * p.Co * TAPS * p.Ci;''','''(size_t)wb.split * p.Co * TAPS * p.Ci;''',2)
rep('''if (k.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)''','''if (k.dbg && blockIdx.x == 0 && threadIdx.x == 0)''')
# host grids
rep('''        dim3 grid64(p.nsplit, ceil_div(p.Co, 64), ceil_div(p.Ci, 64));''','''        k.ny = ceil_div(p.Co, 64), k.nz = ceil_div(p.Ci, 64);
        dim3 grid64(ceil_div(p.nsplit * k.ny, 8) * 8 * k.nz, 1, 1);''')
rep('''    dim3 grid(p.nsplit, ceil_div(p.Co, 32), ceil_div(p.Ci, 32));''','''    k.ny = ceil_div(p.Co, 32), k.nz = ceil_div(p.Ci, 32);
    dim3 grid(ceil_div(p.nsplit * k.ny, 8) * 8 * k.nz, 1, 1);''')
open(p,'w').write(s)
