/* x265_oracle_loop.c — TEST INFRASTRUCTURE ONLY (see x265_oracle.h): deblocking edge filters, SAO offset application and SAO statistics
 * (SURVEY.md §8f rank 4), restated on the CPU. */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

#define PIX uint8_t
#define FN(x) x##_8
#include "x265_oracle_loop.inc"
#undef PIX
#undef FN
#define PIX uint16_t
#define FN(x) x##_16
#include "x265_oracle_loop.inc"
#undef PIX
#undef FN
