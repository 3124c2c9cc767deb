// TEST INFRASTRUCTURE — stand-in for <hip/hip_fp16.h> (tests/emu).
#pragma once
#include "hip_runtime.h"
struct __half2 { _Float16 x, y; };
static inline __half2 __floats2half2_rn(float a, float b) { return {(_Float16)a, (_Float16)b}; }  // (x86 float -> _Float16 conversion rounds to nearest even)
static inline float2 __half22float2(__half2 h) { return {(float)h.x, (float)h.y}; }
