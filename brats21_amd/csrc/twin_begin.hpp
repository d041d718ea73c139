// First project include of every translation unit that handles 16-bit activations.  In the -DBRATS_FP16 twin build (see
// common.hpp) it pulls in every system header the project headers use -- their include guards make the later includes no-ops --
// and opens namespace brats_f16, so that the fp16 kernels, their host stubs and the file-scope state are distinct symbols
// from the bf16 build's.  twin_end.hpp closes it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <type_traits>
#include "../../include/brats_hip.h"
#ifdef BRATS_FP16
namespace brats_f16 {
#endif
