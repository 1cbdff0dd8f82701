p='/root/repo/stlpose_amd/csrc/conv_core.hip'
s=open(p).read()
s=s.replace("constexpr int PSA = 96;\n","constexpr int PSA = 96;\n\n// debug-only phase stamps (block 0, thread 0; enabled by STL_CONV_STAMPS=1): never read by the kernel\n__device__ long long g_stamps[32];\n#define STAMP(i)                                                          \\\n    do {                                                                  \\\n        if (k.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps[i] = wall_clock64(); \\\n    } while (0)\n",1)
s=s.replace("    int off_cs, off_cm, off_a, off_b, off_red;\n    int TH, TW;\n};","    int off_cs, off_cm, off_a, off_b, off_red;\n    int TH, TW;\n    int dbg;\n};")
s=s.replace("    float* cs = reinterpret_cast<float*>(smem + k.off_cs);  // [3][cipad] source transform","    STAMP(0);\n    float* cs = reinterpret_cast<float*>(smem + k.off_cs);  // [3][cipad] source transform")
s=s.replace("    // ---- loop-invariant per-thread descriptors\n","    STAMP(1);\n    // ---- loop-invariant per-thread descriptors\n",1)
s=s.replace("    V16 ra[NVA], rq[Q ? NVA : 1], rb[NVB];\n    int a_go[NVA];\n","    STAMP(2);\n    V16 ra[NVA], rq[Q ? NVA : 1], rb[NVB];\n    int a_go[NVA];\n")
s=s.replace("    if (have) tile_setup(t, a_go);\n    issue(a_go, 0, have);\n    __syncthreads();  // constants + resident filters visible\n","    STAMP(3);\n    if (have) tile_setup(t, a_go);\n    issue(a_go, 0, have);\n    __syncthreads();  // constants + resident filters visible\n    STAMP(4);\n")
s=s.replace("        write_lds(a_go, ch0 * CK);\n        __syncthreads();\n        const bool last_chunk","        write_lds(a_go, ch0 * CK);\n        __syncthreads();\n        if (ch0 == 0) STAMP(5);\n        const bool last_chunk")
s=s.replace("        issue(a_go, chn * CK, have_n);  // next stage's loads land during the MFMAs below\n","        issue(a_go, chn * CK, have_n);  // next stage's loads land during the MFMAs below\n        if (ch0 == 0) STAMP(6);\n")
s=s.replace("        __syncthreads();  // everyone is done with sA/sB of this stage\n        if (last_chunk) {","        __syncthreads();  // everyone is done with sA/sB of this stage\n        if (ch0 == 0) STAMP(7);\n        if (last_chunk) {\n            STAMP(8);")
s=s.replace("        it = itn, t = tn, ch0 = chn, have = have_n;\n    }\n","        if (last_chunk) STAMP(9);\n        it = itn, t = tn, ch0 = chn, have = have_n;\n    }\n    STAMP(10);\n")
s=s.replace("    // ---- flush statistics: lanes of one 16-lane group","    // ---- flush statistics: lanes of one 16-lane group",1)
# stamp at very end: append before the final closing of kernel: find the atomicAdd loop end
s=s.replace("                atomicAdd(dst + (size_t)(blockIdx.x & (STL_NSHARD - 1)) * 2 * p.Co + which * p.Co + n0 + cl, (double)s);\n        }\n    }\n}","                atomicAdd(dst + (size_t)(blockIdx.x & (STL_NSHARD - 1)) * 2 * p.Co + which * p.Co + n0 + cl, (double)s);\n        }\n    }\n    STAMP(11);\n}")
s=s.replace("    ConvK k;\n    k.p = p;","    ConvK k;\n    k.p = p;\n    k.dbg = getenv(\"STL_CONV_STAMPS\") ? 1 : 0;")
s=s.replace('extern "C" int stl_conv_plan(stl_conv* pp) {','''extern "C" int stl_debug_conv_stamps(long long* host12) {
    return hipMemcpyFromSymbol(host12, HIP_SYMBOL(g_stamps), 12 * sizeof(long long)) == hipSuccess ? 0 : stl_set_error("stamps: copy failed");
}

extern "C" int stl_conv_plan(stl_conv* pp) {''')
open(p,'w').write(s)
p='/root/repo/include/stlpose_hip.h'
s=open(p).read()
s=s.replace("int stl_conv_plan(stl_conv* p);","int stl_conv_plan(stl_conv* p);\n/* Debug: phase time stamps (100 MHz ticks) of block 0 of the last conv launched with STL_CONV_STAMPS=1. */\nint stl_debug_conv_stamps(long long* host12);")
open(p,'w').write(s)
p='/root/repo/stlpose_amd/capi.py'
s=open(p).read()
s=s.replace('    "stl_conv_plan": [C.POINTER(Conv)],','    "stl_conv_plan": [C.POINTER(Conv)],\n    "stl_debug_conv_stamps": [vp],')
open(p,'w').write(s)
print("ok")
