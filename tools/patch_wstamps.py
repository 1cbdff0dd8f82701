p='/root/repo/stlpose_amd/csrc/wgrad.hip'
s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, (s.count(a), a)
    s=s.replace(a,b)
rep('''struct WgK {
    stl_wgrad p;''','''// debug-only phase stamps (block 0, thread 0; STL_CONV_STAMPS=1): never read by the kernel
__device__ long long g_wstamps[16];
#define WSTAMP(i)                                                                                   \\
    do {                                                                                            \\
        if (k.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_wstamps[i] = wall_clock64(); \\
    } while (0)

struct WgK {
    stl_wgrad p;
    int dbg;''')
rep('''    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.z * 32;''','''    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.z * 32;
    WSTAMP(0);''')
rep('''    int t = blockIdx.x;
    bool have = t < k.npt;
    if (have) setup(t);
    issue(have);
    __syncthreads();  // constants visible
''','''    int t = blockIdx.x;
    bool have = t < k.npt;
    WSTAMP(1);
    if (have) setup(t);
    issue(have);
    WSTAMP(2);
    __syncthreads();  // constants visible
    WSTAMP(3);
    bool first = true;
''')
rep('''        write_lds();
        __syncthreads();
        const int tn = t + gridDim.x;''','''        write_lds();
        __syncthreads();
        if (first) WSTAMP(4);
        const int tn = t + gridDim.x;''')
rep('''        __syncthreads();
        t = tn, have = have_n;
    }''','''        __syncthreads();
        if (first) WSTAMP(5);
        first = false;
        t = tn, have = have_n;
    }
    WSTAMP(6);''')
rep('''    float* slab = p.partial + (size_t)blockIdx.x * p.Co * TAPS * p.Ci;''','''    WSTAMP(7);
    float* slab = p.partial + (size_t)blockIdx.x * p.Co * TAPS * p.Ci;''')
rep('''        slab[((size_t)(co0 + col) * TAPS + tap) * p.Ci + ci0 + cil] = red[((mt * 2 + nt) * TAPS + tap) * 256 + ln * 4 + r];
    }
}''','''        slab[((size_t)(co0 + col) * TAPS + tap) * p.Ci + ci0 + cil] = red[((mt * 2 + nt) * TAPS + tap) * 256 + ln * 4 + r];
    }
    WSTAMP(8);
}''')
rep('''    k.p = p;
    k.taps = p.ks * p.ks;''','''    k.p = p;
    k.dbg = getenv("STL_CONV_STAMPS") ? 1 : 0;
    k.taps = p.ks * p.ks;''')
rep('''extern "C" int stl_conv_wgrad(''','''extern "C" int stl_debug_wgrad_stamps(long long* host16) {
    return hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_wstamps), 16 * sizeof(long long)) == hipSuccess ? 0 : stl_set_error("stamps: copy failed");
}

extern "C" int stl_conv_wgrad(''')
open(p,'w').write(s)

p='/root/repo/include/stlpose_hip.h'
s=open(p).read()
rep('int stl_debug_conv_stamps2(long long* host64);','int stl_debug_conv_stamps2(long long* host64);\nint stl_debug_wgrad_stamps(long long* host16);')
open(p,'w').write(s)
