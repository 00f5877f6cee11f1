/*
 * ref_rng_shim.c -- TEST INSTRUMENTATION for the reference build in oracle/_ref.
 *
 * The reference draws every random number through the Fortran intrinsic
 * `random_number` (src/polychord/random_utils.F90:128), which amdflang lowers to
 * the flang runtime entry points below.  Linking this object in front of
 * libflang_rt.runtime.a replaces ONLY that generator: the reference sources are
 * compiled untouched, where they lie.  The stream handed out is the oracle's
 * sequential Philox stream (pc_oracle.c, pc_rng_u in sequential mode), so that
 * `pc_oracle_run(sequential_rng=1, batch=1)` and the reference consume identical
 * numbers in identical order -- the strongest pin available: every comparison,
 * every death and every likelihood call of the two runs must coincide.
 *
 * Test infrastructure only; never linked into the shipped engine.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
/* flang's CFI descriptor header (flang/ISO_Fortran_binding.h:123-167), restated so that
 * gcc does not need clang's resource directory on its include path */
typedef struct { ptrdiff_t lower_bound, extent, sm; } CFI_dim_t;
typedef struct {
    void *base_addr; size_t elem_len; int version;
    unsigned char rank; signed char type; unsigned char attribute; unsigned char extra;
    CFI_dim_t dim[];
} CFI_cdesc_t;
#include "pc_oracle.h"

static uint32_t g_key[2] = { 0u, 0x504F4C59u };
static uint64_t g_seq = 0;

/* called by the driver before each run */
void pc_shim_reset(uint32_t seed) { g_key[0] = seed; g_key[1] = 0x504F4C59u; g_seq = 0; }
uint64_t pc_shim_consumed(void) { return g_seq; }
/* the farm (oracle/Makefile ref_mpi): every rank draws from a stream of its own -- the administrator's is the run's, worker w's
 * (rank w) carries w in the second key word, as pc_oracle.c's farm mode expects */
void pc_shim_set_rank(uint32_t rank) { g_key[1] = 0x504F4C59u ^ (rank * 0x9E3779B9u); g_seq = 0; }

static double next_u(void)
{
    uint64_t n = g_seq++;
    return pc_uniform_keyed(g_key, PC_DOM_SEQ, (uint32_t)(n >> 32), 0u, (uint32_t)n);
}

void _FortranARandomInit(_Bool repeatable, _Bool image_distinct) { (void)repeatable; (void)image_distinct; }

void _FortranARandomNumber(const CFI_cdesc_t *h, const char *src, int line)
{
    (void)src; (void)line;
    size_t n = 1;
    for (int r = 0; r < h->rank; ++r) n *= (size_t)h->dim[r].extent;
    /* the reference only harvests contiguous real(8) scalars / rank-1 arrays */
    if (h->elem_len == 8) {
        if (h->rank == 0) { *(double *)h->base_addr = next_u(); return; }
        char *p = (char *)h->base_addr;
        ptrdiff_t sm = h->dim[0].sm;
        for (size_t i = 0; i < n; ++i) *(double *)(p + (ptrdiff_t)i * sm) = next_u();
    } else if (h->elem_len == 4) {
        char *p = (char *)h->base_addr;
        ptrdiff_t sm = h->rank ? h->dim[0].sm : 4;
        for (size_t i = 0; i < n; ++i) *(float *)(p + (ptrdiff_t)i * sm) = (float)next_u();
    }
}

static void put_int(const CFI_cdesc_t *d, long v)
{
    if (!d || !d->base_addr) return;
    if (d->elem_len == 8) *(int64_t *)d->base_addr = v; else *(int32_t *)d->base_addr = (int32_t)v;
}

void _FortranARandomSeedSize(const CFI_cdesc_t *size, const char *s, int l) { (void)s; (void)l; put_int(size, 1); }
void _FortranARandomSeedPut(const CFI_cdesc_t *put, const char *s, int l) { (void)put; (void)s; (void)l; g_seq = 0; }
void _FortranARandomSeedGet(const CFI_cdesc_t *get, const char *s, int l) { (void)s; (void)l; put_int(get, 0); }
void _FortranARandomSeedDefaultPut(void) { g_seq = 0; }
void _FortranARandomSeed(const CFI_cdesc_t *size, const CFI_cdesc_t *put, const CFI_cdesc_t *get, const char *s, int l)
{
    if (size && size->base_addr) _FortranARandomSeedSize(size, s, l);
    else if (put && put->base_addr) _FortranARandomSeedPut(put, s, l);
    else if (get && get->base_addr) _FortranARandomSeedGet(get, s, l);
    else _FortranARandomSeedDefaultPut();
}
