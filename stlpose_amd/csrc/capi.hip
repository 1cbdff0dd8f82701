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

// Hash of csrc/* and include/* at compile time (stlpose_amd/build.py passes it): tests compare it with the hash of the
// tree, so a library that was not rebuilt after a source edit is caught before it is measured.
#ifndef STL_BUILD_ID
#define STL_BUILD_ID "unknown"
#endif
extern "C" const char* stl_build_id(void) { return STL_BUILD_ID; }

static thread_local const char* g_kname = "";
static thread_local bool g_kname_specific = false;
void stl_note_kernel(const char* name, bool specific) {
    if (specific) {
        g_kname = name, g_kname_specific = true;
    } else {                       // the generic site name only when the launcher did not name its instantiation
        if (!g_kname_specific) g_kname = name;
        g_kname_specific = false;
    }
}
extern "C" const char* stl_last_kernel(void) { return g_kname; }

thread_local StlRecorder* g_stl_recorder = nullptr;
void stl_recorder_push(const StlLaunchRec& r) {
    StlRecorder* rec = g_stl_recorder;
    if (rec->n == rec->cap) {
        const int cap = rec->cap ? 2 * rec->cap : 8;
        StlLaunchRec* nr = new StlLaunchRec[cap];
        for (int i = 0; i < rec->n; ++i) nr[i] = rec->recs[i];
        delete[] rec->recs;
        rec->recs = nr, rec->cap = cap;
    }
    rec->recs[rec->n++] = r;
}
