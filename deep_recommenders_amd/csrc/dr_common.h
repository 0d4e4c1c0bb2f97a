// Shared helpers for the gfx950 kernels behind include/dr_hotpath.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dr_hotpath.h"

#define DR_WAVE 64

#define DR_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return DR_ELAUNCH;           \
    } while (0)

static inline hipStream_t dr_s(dr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// memory-bound kernels: cap the grid at 256 CUs x 8 blocks and grid-stride the rest
static inline int dr_grid_for(int64_t work_items, int items_per_block, int max_blocks = 2048) {
    int64_t g = (work_items + items_per_block - 1) / items_per_block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

__device__ __forceinline__ float dr_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
