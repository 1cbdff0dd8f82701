p='/root/repo/stlpose_amd/csrc/conv_ws.inc'
s=open(p).read()
# loader stamps: per loop iteration i (first 6): [32+4i+0]=loop top, +1 = after write_lds, +2 = after issue, +3 = after barrier
s=s.replace('''        int buf = 0;
        while (have) {
            // registers hold stage s+1 (described by a_go / lch); write it, then issue stage s+2
            if (lhave) write_lds(buf ^ 1, lch * CK);
            step(li, lt, lch, lhave);
            if (lhave && lch == 0) tile_setup(lt);
            issue(lch * CK, lhave);
            __syncthreads();  // (C) one barrier per stage''','''        int buf = 0;
        int dbi = 0;
#define WSTAMP(slot)                                                                                   \\
    do {                                                                                               \\
        if (k.dbg && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 255) == 0 && dbi < 6) g_stamps2[slot] = wall_clock64(); \\
    } while (0)
        while (have) {
            WSTAMP(dbi * 4 + 0);
            // registers hold stage s+1 (described by a_go / lch); write it, then issue stage s+2
            if (lhave) write_lds(buf ^ 1, lch * CK);
            WSTAMP(dbi * 4 + 1);
            step(li, lt, lch, lhave);
            if (lhave && lch == 0) tile_setup(lt);
            issue(lch * CK, lhave);
            WSTAMP(dbi * 4 + 2);
            __syncthreads();  // (C) one barrier per stage
            WSTAMP(dbi * 4 + 3);
            ++dbi;''')
s=s.replace('''        int buf = 0;
        while (have) {
            const char* cA = sA + buf * k.sz_a;''','''        int buf = 0;
        int dbi = 0;
        while (have) {
            WSTAMP(32 + dbi * 4 + 0);
            const char* cA = sA + buf * k.sz_a;''')
s=s.replace('''            if (ch0 + 1 == k.nchunks) {
                // ---- epilogue straight from the accumulators: lane = pixel r16, 4 channels per tile''','''            WSTAMP(32 + dbi * 4 + 1);
            if (ch0 + 1 == k.nchunks) {
                // ---- epilogue straight from the accumulators: lane = pixel r16, 4 channels per tile''')
s=s.replace('''            __syncthreads();  // (C)
            step(it, t, ch0, have);
            buf ^= 1;
        }
        // ---- statistics: xor-reduce''','''            WSTAMP(32 + dbi * 4 + 2);
            __syncthreads();  // (C)
            WSTAMP(32 + dbi * 4 + 3);
            ++dbi;
            step(it, t, ch0, have);
            buf ^= 1;
        }
        // ---- statistics: xor-reduce''')
open(p,'w').write(s)
p='/root/repo/stlpose_amd/csrc/conv_core.hip'
s=open(p).read()
s=s.replace("__device__ long long g_stamps[32];","__device__ long long g_stamps[32];\n__device__ long long g_stamps2[64];  // wave-specialised kernel: [0..23] loader, [32..55] compute (6 stages x 4)")
s=s.replace('''extern "C" int stl_debug_conv_stamps(long long* host12) {
    return hipMemcpyFromSymbol(host12, HIP_SYMBOL(g_stamps), 12 * sizeof(long long)) == hipSuccess ? 0 : stl_set_error("stamps: copy failed");
}''','''extern "C" int stl_debug_conv_stamps(long long* host12) {
    return hipMemcpyFromSymbol(host12, HIP_SYMBOL(g_stamps), 12 * sizeof(long long)) == hipSuccess ? 0 : stl_set_error("stamps: copy failed");
}
extern "C" int stl_debug_conv_stamps2(long long* host64) {
    return hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_stamps2), 64 * sizeof(long long)) == hipSuccess ? 0 : stl_set_error("stamps: copy failed");
}''')
open(p,'w').write(s)
p='/root/repo/include/stlpose_hip.h'
s=open(p).read()
s=s.replace("int stl_debug_conv_stamps(long long* host12);","int stl_debug_conv_stamps(long long* host12);\nint stl_debug_conv_stamps2(long long* host64);")
open(p,'w').write(s)
p='/root/repo/stlpose_amd/capi.py'
s=open(p).read()
s=s.replace('    "stl_debug_conv_stamps": [vp],','    "stl_debug_conv_stamps": [vp],\n    "stl_debug_conv_stamps2": [vp],')
open(p,'w').write(s)
