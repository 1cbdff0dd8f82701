def rep(path,a,b,count=1):
    s=open(path).read()
    assert s.count(a)==count,(path,s.count(a),a)
    open(path,'w').write(s.replace(a,b))
E='/root/repo/stlpose_amd/csrc/elementwise.hip'
rep(E,'''extern "C" int stl_maxpool2x2(''','''// ---------------------------------------------------------------- gaussian heatmap targets
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
    hipLaunchKernelGGL(gaussian_targets_kernel, dim3(nblocks_for(total, 256, 4096)), dim3(256), 0, ST, joints_xy, vis, target, tweight,
                       B * J, Hh, Wh, (double)stride_x, (double)stride_y, sigma);
    STL_LAUNCH_CHECK("gaussian_targets");
    return 0;
}

extern "C" int stl_maxpool2x2(''')
H='/root/repo/include/stlpose_hip.h'
rep(H,'''/* PersonMSELoss forward+backward''','''/* Gaussian heatmap targets on device (reference data/JointsDataset.py:230-286 generate_target):
 * joints_xy [B][J][2] in image pixels, vis [B][J] in {0,1} -> target [B][J][Hh][Wh] (unnormalised
 * gaussian, centre value 1, support 3*sigma, clipped at the border), tweight [B][J] (vis, or 0 when
 * the gaussian lies completely outside the map).  stride = image_size / heatmap_size per axis. */
int stl_gaussian_targets(const float* joints_xy, const float* vis, float* target, float* tweight, int B, int J,
                         int Hh, int Wh, float stride_x, float stride_y, float sigma, void* stream);

/* PersonMSELoss forward+backward''')
C_='/root/repo/stlpose_amd/capi.py'
rep(C_,'''    "stl_heatmap_argmax":''','''    "stl_gaussian_targets": [vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, f32, vp],
    "stl_heatmap_argmax":''')
