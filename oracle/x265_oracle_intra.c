/* x265_oracle_intra.c — TEST INFRASTRUCTURE ONLY (see x265_oracle.h): intra prediction primitives, lowres init and the
 * lookahead intra estimate, restated on the CPU. */
#include "x265_oracle.h"
#include <stdlib.h>
#include <string.h>

#define PIX uint8_t
#define FN(x) x##_8
#include "x265_oracle_intra.inc"
#undef PIX
#undef FN
#define PIX uint16_t
#define FN(x) x##_16
#include "x265_oracle_intra.inc"
#undef PIX
#undef FN
