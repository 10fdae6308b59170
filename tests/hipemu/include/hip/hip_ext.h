// emulator shim: hipExtLaunchKernelGGL lives in hip_runtime.h
#pragma once
#include <hip/hip_runtime.h>
