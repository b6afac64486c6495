// TEST INFRASTRUCTURE ONLY — host emulation of the one hip_ext.h entry the library uses.
#pragma once
#include <hip/hip_runtime.h>

// kernel launch with start/stop events attached to the dispatch: here simply timed around the call
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, startEvent, stopEvent, flags, ...) \
    do {                                                                                             \
        (void)hipEventRecord(startEvent, stream);                                                    \
        hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), stream, ##__VA_ARGS__);     \
        (void)hipEventRecord(stopEvent, stream);                                                     \
    } while (0)
