/* x265_oracle_sadsurf.c — TEST INFRASTRUCTURE ONLY (see x265_oracle.h): the SAD surfaces behind the integer-pel lookups of
 * MotionEstimate::motionEstimate (include/x265hip.h, x265hip_sadsurf_*), restated on the CPU. */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include "../include/x265hip.h"
#include "x265_oracle.h"

#define PIX uint8_t
#define FN(x) x##_8
#include "x265_oracle_sadsurf.inc"
#undef PIX
#undef FN
#define PIX uint16_t
#define FN(x) x##_16
#include "x265_oracle_sadsurf.inc"
#undef PIX
#undef FN
