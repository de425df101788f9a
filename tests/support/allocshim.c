/* allocshim.c — TEST INFRASTRUCTURE (LD_PRELOAD): writes a 32-bit word into chosen bytes of chosen heap blocks before the program sees them.
 *
 *   ALLOCSHIM_SIZE=<n>      only blocks of exactly n bytes (an array new[] of ELEM-byte objects with an 8-byte count in front)
 *   ALLOCSHIM_ELEM=<e>      element size
 *   ALLOCSHIM_OFFSET=<o>    byte offset inside every element
 *   ALLOCSHIM_WORD=<w>      the value
 *
 * tests/test_reference_races.py uses it to put the reference's Analysis::m_refineLevel — a member the reference reads without ever having written
 * it (analysis.cpp:1314, :2019) — into the state recycled heap memory can leave it in.  malloc / posix_memalign / memalign / aligned_alloc only;
 * calloc stays zeroed. */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
extern void* __libc_malloc(size_t);
extern void* __libc_memalign(size_t, size_t);
static size_t g_size, g_elem, g_off;
static uint32_t g_word;
static int g_ready;
static void poison(void* p, size_t n)
{
    if (!g_ready)
    {
        const char* s = getenv("ALLOCSHIM_SIZE"), *e = getenv("ALLOCSHIM_ELEM"), *o = getenv("ALLOCSHIM_OFFSET"), *w = getenv("ALLOCSHIM_WORD");
        g_size = s ? strtoull(s, 0, 10) : 0; g_elem = e ? strtoull(e, 0, 10) : 0; g_off = o ? strtoull(o, 0, 10) : 0; g_word = w ? (uint32_t)strtoul(w, 0, 10) : 0;
        g_ready = 1;
    }
    if (!p || !g_size || n != g_size || !g_elem)
        return;
    for (size_t at = 8; at + g_elem <= n; at += g_elem)
        memcpy((char*)p + at + g_off, &g_word, 4);
}
void* malloc(size_t n) { void* p = __libc_malloc(n); poison(p, n); return p; }
int posix_memalign(void** out, size_t a, size_t n) { void* p = __libc_memalign(a, n); if (!p) return 12; poison(p, n); *out = p; return 0; }
void* memalign(size_t a, size_t n) { void* p = __libc_memalign(a, n); poison(p, n); return p; }
void* aligned_alloc(size_t a, size_t n) { void* p = __libc_memalign(a, n); poison(p, n); return p; }
