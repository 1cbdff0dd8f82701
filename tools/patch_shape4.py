p='/root/repo/stlpose_amd/csrc/conv_core.hip'
s=open(p).read()
# occupancy template parameter: min waves per SIMD
s=s.replace("template <typename T, int KS, int WM, int WN, int MT, int NTW, int NVA, bool Q>\n__global__ __launch_bounds__(64 * WM * WN) void conv_core_kernel(const ConvK k) {",
"template <typename T, int KS, int WM, int WN, int MT, int NTW, int NVA, bool Q, int OCC>\n__global__ __launch_bounds__(64 * WM * WN, OCC) void conv_core_kernel(const ConvK k) {")
s=s.replace("template <typename T, int KS, int WM, int WN, int MT, int NTW, int NVA, bool Q>\nint launch(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {",
"template <typename T, int KS, int WM, int WN, int MT, int NTW, int NVA, bool Q, int OCC = 1>\nint launch(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {")
s=s.replace("conv_core_kernel<T, KS, WM, WN, MT, NTW, NVA, Q>)","conv_core_kernel<T, KS, WM, WN, MT, NTW, NVA, Q, OCC>)")
s=s.replace("// block shapes: 0 = 128 px x 64 co (4 waves), 1 = 512 px x 32 co, 2 = 256 px x 64 co, 3 = 256 px x 128 co (8 waves)",
"// block shapes: 0 = 128 px x 64 co (4 waves), 1 = 512 px x 32 co, 2 = 256 px x 64 co, 3 = 256 px x 128 co (8 waves),\n// 4 = 128 px x 32 co (4 waves, <=128 VGPRs, <=40 KB LDS: four blocks per CU hide each other's latency)")
s=s.replace("constexpr Shape SHAPES[4] = {{128, 64, 256}, {512, 32, 512}, {256, 64, 512}, {256, 128, 512}};","constexpr int NSHAPES = 5;\nconstexpr Shape SHAPES[NSHAPES] = {{128, 64, 256}, {512, 32, 512}, {256, 64, 512}, {256, 128, 512}, {128, 32, 256}};")
s=s.replace('''        case 3:
            if (nva <= 3) return launch<T, KS, 4, 2, 4, 4, 3, Q>(k, grid, lds, st);
            break;
    }''','''        case 3:
            if (nva <= 3) return launch<T, KS, 4, 2, 4, 4, 3, Q>(k, grid, lds, st);
            break;
        case 4:
            if (nva <= 3) return launch<T, KS, 4, 1, 2, 2, 3, Q, 4>(k, grid, lds, st);
            if (nva <= 6) return launch<T, KS, 4, 1, 2, 2, 6, Q, 3>(k, grid, lds, st);
            break;
    }''')
s=s.replace("    for (int shape = 0; shape < 4; ++shape) {","    for (int shape = 0; shape < NSHAPES; ++shape) {")
s=s.replace("        if (shape != 0 && p.stride == 2) continue;       // stride-2 halos only fit the small block","        if (shape != 0 && shape != 4 && p.stride == 2) continue;  // stride-2 halos only fit the small blocks")
s=s.replace("            if ((shape == 0 && nva > 9) || (shape == 1 && nva > 6) || (shape >= 2 && nva > 3)) continue;\n            const size_t lds = lds_bytes(p, shape, th, tw, ck, nullptr);",
"            if ((shape == 0 && nva > 9) || (shape == 1 && nva > 6) || ((shape == 2 || shape == 3) && nva > 3) || (shape == 4 && nva > 6)) continue;\n            const size_t lds = lds_bytes(p, shape, th, tw, ck, nullptr);\n            if (shape == 4 && lds > 40 * 1024 && nva <= 3) continue;  // keep four blocks per CU")
s=s.replace("    if (p.shape >= 0 && p.shape < 4 && p.TH > 0 && p.TW > 0) {","    if (p.shape >= 0 && p.shape < NSHAPES && p.TH > 0 && p.TW > 0) {")
s=s.replace("        STL_CHECK(nv <= (plan.shape == 0 ? 9 : plan.shape == 1 ? 6 : 3), ","        STL_CHECK(nv <= (plan.shape == 0 ? 9 : (plan.shape == 1 || plan.shape == 4) ? 6 : 3), ")
s=s.replace("        if (f >= 0 && f < 4 && !(f != 0 && p.stride == 2)) {","        if (f >= 0 && f < NSHAPES && !(f != 0 && f != 4 && p.stride == 2)) {")
s=s.replace("                if ((f == 0 && nva > 9) || (f == 1 && nva > 6) || (f >= 2 && nva > 3)) continue;","                if ((f == 0 && nva > 9) || ((f == 1 || f == 4) && nva > 6) || ((f == 2 || f == 3) && nva > 3)) continue;")
s=s.replace("    int cap = sh.thr == 512 ? 512 : 1024;","    int cap = sh.thr == 512 ? 512 : (plan.shape == 4 ? 2048 : 1024);")
open(p,'w').write(s)
print('ok')
