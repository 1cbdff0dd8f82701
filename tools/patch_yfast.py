root='/root/repo/stlpose_amd/csrc/'
def rep(path,a,b,count=1):
    s=open(root+path).read()
    assert s.count(a)==count,(path,s.count(a),a)
    open(root+path,'w').write(s.replace(a,b))
rep('conv_core.hip','''    float r_HC, r_TW, r_tc, r_PI, r_vp;  // reciprocals for fdiv
};''','''    float r_HC, r_TW, r_tc, r_PI, r_vp;  // reciprocals for fdiv
    int ny;  // output-channel blocks per pixel tile (they are the FAST block dimension, see kernel)
};''')
rep('conv_core.hip','''    const int n0 = blockIdx.y * BCO;
''','''    // Block order: XCD = blockIdx.x & 7; within an XCD the ny channel blocks of one pixel tile are
    // neighbours, so they run at the same time on the same L2: the input tile is fetched from HBM
    // once for all of them, and the ny pieces of every output row are written together.
    const int bl = blockIdx.x >> 3;
    const int by = bl % k.ny, lx = bl / k.ny;
    const int n0 = by * BCO;
''')
rep('conv_core.hip','''    const int xcd = blockIdx.x & 7, lx = blockIdx.x >> 3, nx = gridDim.x >> 3;''','''    const int xcd = blockIdx.x & 7, nx = (gridDim.x >> 3) / k.ny;''')
rep('conv_ws.inc','''    const int n0 = blockIdx.y * BCO;
''','''    const int bl = blockIdx.x >> 3;  // channel blocks of a pixel tile are neighbours on one XCD (see conv_core_kernel)
    const int by = bl % k.ny, lx = bl / k.ny;
    const int n0 = by * BCO;
''')
rep('conv_ws.inc','''    const int xcd = blockIdx.x & 7, lx = blockIdx.x >> 3, nx = gridDim.x >> 3;''','''    const int xcd = blockIdx.x & 7, nx = (gridDim.x >> 3) / k.ny;''')
rep('conv_core.hip','''    dim3 grid(gx, ceil_div(p.Co, sh.co));''','''    k.ny = ceil_div(p.Co, sh.co);
    dim3 grid(gx * k.ny, 1);''')
rep('conv_core.hip','''npt=%d grid=(%d,%d) lds=%zu''','''npt=%d grid=(%d x %d) lds=%zu''')
rep('conv_core.hip','''plan.shape, plan.TH, plan.TW, k.npt, gx, grid.y, lds, nva, k.nchunks);''','''plan.shape, plan.TH, plan.TW, k.npt, gx, k.ny, lds, nva, k.nchunks);''')
