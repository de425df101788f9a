// tld_layout.cpp — TEST INFRASTRUCTURE: prints the two numbers tests/test_reference_races.py needs to poison Analysis::m_refineLevel inside the
// reference's `new ThreadLocalData[numTLD]` (frameencoder.cpp:298): the element size and the member's offset, from the reference's own headers.
#include <cstddef>
#include <cstdio>
#define protected public
#define private public
#include "common.h"
#include "analysis.h"
using namespace X265_NS;
int main()
{
    printf("%zu %zu\n", sizeof(ThreadLocalData), offsetof(Analysis, m_refineLevel));
    return 0;
}
