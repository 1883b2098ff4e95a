// Host emulation of the one hip_ext.h entry the engine uses: a launch without the AQL barrier bit is an
// ordinary (serial) launch here.
#pragma once
#include "hip_runtime.h"
#define hipExtAnyOrderLaunch 0x01
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, start_event, stop_event, flags, ...) \
    hipemu_launch(kernel, grid, block, __VA_ARGS__)
