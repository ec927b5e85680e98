// amwg_stdint.h -- fixed-width integer names for both hipcc and hiprtc builds.
#pragma once
#if defined(__HIPCC_RTC__)
// hiprtc has no <stdint.h>; it keeps its own fixed-width names in a private namespace
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long long uintptr_t;
#else
#include <stddef.h>
#include <stdint.h>
#endif
