/*
 * stlpose_hip_debug.h -- entry points of the STAMPED diagnostic build only (python -m stlpose_amd.build --stamps ->
 * libstlpose_hip_stamps.so, compiled with -DSTL_STAMPS).  The product library libstlpose_hip.so neither carries the in-kernel
 * phase stamps nor exports these symbols (a disabled stamp still costs the stage loop a drained load queue).
 * Used by tools/conv_stamps.py, tools/conv_db_stamps.py, tools/wgrad_probe.py, tools/wgrad_tile_stamps.py.
 */
#ifndef STLPOSE_HIP_DEBUG_H
#define STLPOSE_HIP_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif
/* phase time stamps (100 MHz ticks) of block 0 of the last conv launched with STL_CONV_STAMPS=1 */
int stl_debug_conv_stamps(long long* host12);
int stl_debug_conv_stamps2(long long* host64);
int stl_debug_wgrad_stamps(long long* host16);
int stl_debug_wgrad_stamps2(long long* host64); /* per-tile stamps of block 0 (weight gradient) */
#ifdef __cplusplus
}
#endif
#endif
