// Error reporting + version of the C ABI.
#include <stdarg.h>
#include "common.cuh"

static thread_local char g_err[512] = "";

int stl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

extern "C" const char* stl_last_error(void) { return g_err; }
extern "C" int stl_version(void) { return 1; }

// ---- CU-masked streams: a HIP stream whose kernels may only occupy the compute units named by `mask`
// (bit i of word i / 32 = CU i in the driver's enumeration).  The planner uses them to keep the off-chain
// weight-gradient launches of backward on a fixed share of the chip, away from the data-gradient chain.
extern "C" int stl_stream_create_masked(const uint32_t* mask, int nwords, void** out) {
    STL_CHECK(mask && out && nwords > 0 && nwords <= 32, "stream_create_masked: bad arguments");
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask);
    STL_CHECK(e == hipSuccess, "stream_create_masked: %s", hipGetErrorString(e));
    *out = s;
    return 0;
}
extern "C" int stl_stream_destroy(void* stream) {
    STL_CHECK(hipStreamDestroy((hipStream_t)stream) == hipSuccess, "stream_destroy: failed");
    return 0;
}

// Probe: every block reports where it ran -- out[2b] = XCC_ID, out[2b+1] = HW_ID (wave / SIMD / CU / SH / SE fields)
// -- after spinning `spin` clock reads so that a grid spreads over every CU the stream may use.
__global__ void probe_place_kernel(uint32_t* out, int spin) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20);       // HW_REG_XCC_ID, 4 bits
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID, 32 bits
    }
}
extern "C" int stl_probe_placement(uint32_t* out, int nblocks, int spin_ticks, void* stream) {
    STL_CHECK(out && nblocks > 0, "probe_placement: bad arguments");
    hipLaunchKernelGGL(probe_place_kernel, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, out, spin_ticks);
    STL_LAUNCH_CHECK("probe_placement");
    return 0;
}
