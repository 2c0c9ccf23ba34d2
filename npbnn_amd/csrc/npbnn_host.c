/*
 * npbnn_host.c — host-side pre-draw of Metropolis-Hastings proposals with numpy-identical streams.
 *
 * The reference draws every proposal from a numpy Generator inside MCMC.mh_step
 * (np_bnn/BNN_env.py:383-384, 446-447, 452-453, 493; UpdateNormal, np_bnn/BNN_mcmc.py:57-69).  With the
 * default unbounded random-walk proposal the *draws* do not depend on the chain state, so K iterations
 * can be drawn ahead of time and shipped to the GPU in one buffer while the chain itself runs on the
 * device.  To keep seed parity the draws below come from numpy's own C distribution routines
 * (libnpyrandom.a, shipped in the numpy wheel: random_standard_uniform_fill, random_bounded_uint64_fill,
 * random_normal) driven either by the Generator's live bitgen_t (plain chains) or by a PCG64 seeded
 * exactly like np.random.default_rng(seed) (MC3 chains re-seed every iteration with
 * iteration + mcmc_id, BNN_env.py:384).
 *
 * Draw order of one iteration (must not change):
 *   [reseed]  random(n_layers)  { integers(rows) integers(cols) normal(scale[ix,iy]) } per updated layer
 *   random()  (accept test)
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "numpy/random/bitgen.h"
#include "numpy/random/distributions.h"

#define NPBNN_HOST_MAX_LAYERS 8

/* ---------------------------------------------------------------------------------------------
 * SeedSequence(entropy).generate_state(4, uint64) for a non-negative integer seed
 * (numpy/random/bit_generator.pyx; the algorithm of M.E. O'Neill's seed_seq_fe).
 * ------------------------------------------------------------------------------------------- */
#define SS_INIT_A 0x43b0d7e5u
#define SS_MULT_A 0x931e8875u
#define SS_INIT_B 0x8b51f9ddu
#define SS_MULT_B 0x58f38dedu
#define SS_MIX_L 0xca01f9ddu
#define SS_MIX_R 0x4973f715u
#define SS_XSHIFT 16
#define SS_POOL 4

static uint32_t ss_hashmix(uint32_t value, uint32_t* hash_const) {
    value ^= *hash_const;
    *hash_const *= SS_MULT_A;
    value *= *hash_const;
    value ^= value >> SS_XSHIFT;
    return value;
}

static uint32_t ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = SS_MIX_L * x - SS_MIX_R * y;
    r ^= r >> SS_XSHIFT;
    return r;
}

static void seed_sequence_state(uint64_t seed, uint64_t out[4]) {
    uint32_t entropy[2];
    int n_ent = 1;
    entropy[0] = (uint32_t)(seed & 0xffffffffu);
    entropy[1] = (uint32_t)(seed >> 32);
    if (entropy[1] != 0) n_ent = 2;
    uint32_t pool[SS_POOL];
    uint32_t hash_const = SS_INIT_A;
    for (int i = 0; i < SS_POOL; ++i) pool[i] = ss_hashmix(i < n_ent ? entropy[i] : 0u, &hash_const);
    for (int i_src = 0; i_src < SS_POOL; ++i_src)
        for (int i_dst = 0; i_dst < SS_POOL; ++i_dst)
            if (i_src != i_dst) pool[i_dst] = ss_mix(pool[i_dst], ss_hashmix(pool[i_src], &hash_const));
    /* generate_state(8 x uint32) */
    uint32_t words[8];
    uint32_t hc = SS_INIT_B;
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pool[i % SS_POOL];
        v ^= hc;
        hc *= SS_MULT_B;
        v *= hc;
        v ^= v >> SS_XSHIFT;
        words[i] = v;
    }
    for (int i = 0; i < 4; ++i) out[i] = (uint64_t)words[2 * i] | ((uint64_t)words[2 * i + 1] << 32);
}

/* ---------------------------------------------------------------------------------------------
 * PCG64 (XSL-RR 128/64), the default bit generator of np.random.default_rng
 * ------------------------------------------------------------------------------------------- */
typedef unsigned __int128 u128;

typedef struct {
    u128 state, inc;
    int has_uint32;
    uint32_t uinteger;
} pcg64_t;

#define PCG_MULT ((((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF645ULL)

static inline void pcg_step(pcg64_t* r) { r->state = r->state * PCG_MULT + r->inc; }

static void pcg64_seed(pcg64_t* r, uint64_t seed) {
    uint64_t v[4];
    seed_sequence_state(seed, v);
    const u128 initstate = (((u128)v[0]) << 64) | v[1];
    const u128 initseq = (((u128)v[2]) << 64) | v[3];
    r->state = 0;
    r->inc = (initseq << 1) | 1;
    pcg_step(r);
    r->state += initstate;
    pcg_step(r);
    r->has_uint32 = 0;
    r->uinteger = 0;
}

static uint64_t pcg64_next64(void* st) {
    pcg64_t* r = (pcg64_t*)st;
    pcg_step(r);
    const uint64_t hi = (uint64_t)(r->state >> 64), lo = (uint64_t)r->state;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(r->state >> 122);
    return (x >> rot) | (x << ((-rot) & 63));
}

static uint32_t pcg64_next32(void* st) {
    pcg64_t* r = (pcg64_t*)st;
    if (r->has_uint32) {
        r->has_uint32 = 0;
        return r->uinteger;
    }
    const uint64_t next = pcg64_next64(st);
    r->has_uint32 = 1;
    r->uinteger = (uint32_t)(next >> 32);
    return (uint32_t)(next & 0xffffffffu);
}

static double pcg64_next_double(void* st) { return (pcg64_next64(st) >> 11) * (1.0 / 9007199254740992.0); }

static void pcg64_bitgen(pcg64_t* r, bitgen_t* bg) {
    bg->state = r;
    bg->next_uint64 = pcg64_next64;
    bg->next_uint32 = pcg64_next32;
    bg->next_double = pcg64_next_double;
    bg->next_raw = pcg64_next64;
}

/* ---------------------------------------------------------------------------------------------
 * The same draws without a call through a function pointer per random number ("fast path").
 *
 * numpy's routines reach the generator through bitgen_t's function pointers and are calls themselves: ~4.5 ns per number, 5.8 us
 * per iteration of BASELINE config 2 (428 normals, 856 bounded integers), 10 us of config 5 - as long as the GPU takes to evaluate
 * the iteration.  Below the generator (the PCG64 above, on a COPY of the Generator's state that the caller hands in and takes back as
 * plain integers - no assumption about numpy's structs) and the two distributions are inlined:
 *   - bounded integers: Lemire's multiply-shift with rejection on 32-bit halves of the 64-bit outputs (low half first), which is
 *     what random_bounded_uint64_fill does for ranges below 2^32;
 *   - normal deviates: the fast branch of the 256-layer ziggurat (one 64-bit output: 8 bits layer, 1 bit sign, 52 bits magnitude;
 *     98.8 % of the draws).  Its two tables are LEARNT from numpy's own random_standard_normal by feeding it crafted outputs
 *     (wi[layer] = the value returned for magnitude 1; ki[layer] = the first magnitude that leaves the fast branch, by bisection),
 *     so they are the installed numpy's whatever its version; any other draw rewinds the generator by the one output and lets
 *     numpy's routine make it.
 * Before first use the inlined forms are run against numpy's routines on identical generator states (2^18 normals, bounded
 * integers over a spread of ranges, uniforms); any difference disables the fast path for the life of the process.
 * ------------------------------------------------------------------------------------------- */
static double zig_wi[256];
static uint64_t zig_ki[256];
static int fast_state = 0;          /* 0 not initialised, 1 verified, -1 disabled */

typedef struct { uint64_t r0, filler; int n64, ndbl; } probe_t;
static uint64_t probe_next64(void* st) { probe_t* p = (probe_t*)st; return p->n64++ == 0 ? p->r0 : p->filler; }
static uint32_t probe_next32(void* st) { return (uint32_t)probe_next64(st); }
static double probe_double(void* st) { ((probe_t*)st)->ndbl++; return 0.25; }

/* does numpy's routine take the fast branch for (layer, magnitude)?  *x = what it returns */
static int probe_fast(int layer, uint64_t rabs, double* x) {
    probe_t p;
    bitgen_t bg;
    p.r0 = (uint64_t)layer | (rabs << 9);                 /* sign bit (bit 8) clear */
    p.filler = 0;                                          /* layer 0, magnitude 0: returns 0 at once (ends any retry loop) */
    p.n64 = p.ndbl = 0;
    bg.state = &p; bg.next_uint64 = probe_next64; bg.next_uint32 = probe_next32; bg.next_double = probe_double; bg.next_raw = probe_next64;
    const double v = random_standard_normal(&bg);
    if (x) *x = v;
    return p.ndbl == 0 && p.n64 == 1;
}

static inline uint32_t fast_next32(pcg64_t* r) {
    if (r->has_uint32) { r->has_uint32 = 0; return r->uinteger; }
    pcg_step(r);
    const uint64_t hi = (uint64_t)(r->state >> 64), lo = (uint64_t)r->state;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(r->state >> 122);
    const uint64_t next = (x >> rot) | (x << ((-rot) & 63));
    r->has_uint32 = 1;
    r->uinteger = (uint32_t)(next >> 32);
    return (uint32_t)(next & 0xffffffffu);
}

static inline uint64_t fast_next64(pcg64_t* r) {
    pcg_step(r);
    const uint64_t hi = (uint64_t)(r->state >> 64), lo = (uint64_t)r->state;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(r->state >> 122);
    return (x >> rot) | (x << ((-rot) & 63));
}

static inline double fast_double(pcg64_t* r) { return (fast_next64(r) >> 11) * (1.0 / 9007199254740992.0); }

/* integers in [0, rng] for rng < 2^32 - 1 (numpy: buffered_bounded_lemire_uint32) */
static inline uint32_t fast_bounded32(pcg64_t* r, uint32_t rng) {
    const uint32_t rng_excl = rng + 1u;
    uint64_t m = (uint64_t)fast_next32(r) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        const uint32_t threshold = (0xffffffffu - rng) % rng_excl;
        while (leftover < threshold) {
            m = (uint64_t)fast_next32(r) * rng_excl;
            leftover = (uint32_t)m;
        }
    }
    return (uint32_t)(m >> 32);
}

static inline double fast_standard_normal(pcg64_t* r, bitgen_t* own_bg) {
    const u128 before = r->state;                       /* (the 32-bit buffer is not touched by a 64-bit draw) */
    const uint64_t u = fast_next64(r);
    const int layer = (int)(u & 0xff);
    const uint64_t v = u >> 8;
    const uint64_t rabs = (v >> 1) & 0x000fffffffffffffULL;
    if (rabs < zig_ki[layer]) {
        const double x = (double)rabs * zig_wi[layer];
        return (v & 1) ? -x : x;
    }
    r->state = before;                                   /* the rare branches: numpy's own routine, from the same output on */
    return random_standard_normal(own_bg);
}

/* ---- raw outputs made a block ahead of their use ----
 * One PCG64 output hangs on the one before it through a 128-bit multiply-add and, with the state behind a pointer, a store and a load:
 * ~10 cycles per output, and every draw of a wide proposal (53 k perturbed weights per iteration of a million-weight network: two
 * bounded integers and a normal each) waits on that chain - 10.6 ns per entry on the GPU box's EPYC 9575F, 6.6 of them the normal.
 * state[i + 4] = A^4 state[i] + c (A^3 + A^2 + A + 1): four chains side by side fill a block of outputs with no dependence between
 * neighbours; the draws then read the block.  The SEQUENCE of outputs is the generator's own, and so is the order the draws take
 * them in (rejections of Lemire's method and the ziggurat's rare branches simply take the next ones); blk_sync puts the generator's
 * state where the last output USED leaves it (before numpy's own routines take over, and at the end of every iteration). */
#define PCG_BLK 32
typedef struct {
    uint64_t out[PCG_BLK];
    int pos, n;            /* next output to hand out; outputs in the block (0: none made) */
    u128 start;            /* the state the block was made from */
} pcg_block_t;

static inline uint64_t pcg_output(u128 st) {
    const uint64_t hi = (uint64_t)(st >> 64), lo = (uint64_t)st;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(st >> 122);
    return (x >> rot) | (x << ((-rot) & 63));
}

static void blk_refill(pcg64_t* r, pcg_block_t* b) {
    const u128 A = PCG_MULT, A2 = A * A, A4 = A2 * A2, c = r->inc, c4 = c * (A2 * A + A2 + A + 1);
    b->start = r->state;
    u128 s0 = r->state * A + c, s1 = s0 * A + c, s2 = s1 * A + c, s3 = s2 * A + c;
    for (int k = 0; k < PCG_BLK; k += 4) {
        b->out[k] = pcg_output(s0); b->out[k + 1] = pcg_output(s1); b->out[k + 2] = pcg_output(s2); b->out[k + 3] = pcg_output(s3);
        if (k + 4 == PCG_BLK) r->state = s3;           /* the state after the block's last output */
        s0 = s0 * A4 + c4; s1 = s1 * A4 + c4; s2 = s2 * A4 + c4; s3 = s3 * A4 + c4;
    }
    b->pos = 0;
    b->n = PCG_BLK;
}

static void blk_sync(pcg64_t* r, pcg_block_t* b) {
    if (b->n == 0) return;
    u128 st = b->start;
    for (int i = 0; i < b->pos; ++i) st = st * PCG_MULT + r->inc;
    r->state = st;
    b->pos = b->n = 0;
}

static inline uint64_t blk_next64(pcg64_t* r, pcg_block_t* b) {
    if (b->pos == b->n) blk_refill(r, b);
    return b->out[b->pos++];
}

static inline uint32_t blk_next32(pcg64_t* r, pcg_block_t* b) {
    if (r->has_uint32) { r->has_uint32 = 0; return r->uinteger; }
    const uint64_t next = blk_next64(r, b);
    r->has_uint32 = 1;
    r->uinteger = (uint32_t)(next >> 32);
    return (uint32_t)(next & 0xffffffffu);
}

static inline double blk_double(pcg64_t* r, pcg_block_t* b) { return (blk_next64(r, b) >> 11) * (1.0 / 9007199254740992.0); }

static inline uint32_t blk_bounded32(pcg64_t* r, pcg_block_t* b, uint32_t rng) {
    const uint32_t rng_excl = rng + 1u;
    uint64_t m = (uint64_t)blk_next32(r, b) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        const uint32_t threshold = (0xffffffffu - rng) % rng_excl;
        while (leftover < threshold) {
            m = (uint64_t)blk_next32(r, b) * rng_excl;
            leftover = (uint32_t)m;
        }
    }
    return (uint32_t)(m >> 32);
}

/* n bounded integers (numpy: random_bounded_uint64_fill -> buffered_bounded_lemire_uint32 for ranges below 2^32): two per 64-bit output,
 * low half first - what the buffered 32-bit draws amount to - with the rare rejection (a product's low word below the range) redone by
 * the one-at-a-time routine from the same output on */
static inline void blk_bounded32_fill(pcg64_t* r, pcg_block_t* b, uint32_t rng, int n, uint64_t* out) {
    const uint32_t rng_excl = rng + 1u;
    int j = 0;
    if (n > 0 && r->has_uint32) out[j++] = blk_bounded32(r, b, rng);       /* (the half a previous draw left behind goes first) */
    for (; j + 1 < n; j += 2) {
        const uint64_t raw = blk_next64(r, b);
        const uint64_t m0 = (uint64_t)(uint32_t)raw * rng_excl, m1 = (raw >> 32) * rng_excl;
        if ((uint32_t)m0 < rng_excl || (uint32_t)m1 < rng_excl) {          /* possibly a rejection: one at a time, from this output on */
            b->pos -= 1;
            out[j] = blk_bounded32(r, b, rng);
            out[j + 1] = blk_bounded32(r, b, rng);
            if (r->has_uint32) {                                            /* (a rejection shifted the halves: realign on the next draw) */
                if (j + 2 < n) { out[j + 2] = blk_bounded32(r, b, rng); j += 1; }
            }
            continue;
        }
        out[j] = m0 >> 32;
        out[j + 1] = m1 >> 32;
        r->uinteger = (uint32_t)(raw >> 32);                                /* (what the buffered draws leave in the generator's state: the half last held) */
    }
    for (; j < n; ++j) out[j] = blk_bounded32(r, b, rng);
}

static inline double blk_standard_normal(pcg64_t* r, pcg_block_t* b, bitgen_t* own_bg) {
    const uint64_t u = blk_next64(r, b);
    const int layer = (int)(u & 0xff);
    const uint64_t v = u >> 8;
    const uint64_t rabs = (v >> 1) & 0x000fffffffffffffULL;
    if (rabs < zig_ki[layer]) {
        /* (the sign as a bit, not as a choice: a random branch is mispredicted every other time - half of what a normal cost;
         *  x >= 0, so setting the sign bit is the negation, -0.0 included) */
        const double x = (double)(int64_t)rabs * zig_wi[layer];
        uint64_t bits;
        memcpy(&bits, &x, sizeof bits);
        bits |= (v & 1) << 63;
        double y;
        memcpy(&y, &bits, sizeof y);
        return y;
    }
    b->pos -= 1;                                         /* the rare branches: numpy's own routine, from the same output on */
    blk_sync(r, b);
    return random_standard_normal(own_bg);
}

static int fast_build_and_verify(void);
/* Runs ONCE per process (pthread_once: ctypes releases the interpreter lock, so the helper thread's pre-draw and a pre-draw on the
 * main thread - or two samplers - can arrive together); fast_state says "verified" only after the last comparison has passed, so
 * nobody draws from tables that verification is about to reject. */
static void fast_init_once(void) {
    if (getenv("NPBNN_NO_FAST_PREDRAW")) { __atomic_store_n(&fast_state, -1, __ATOMIC_RELEASE); return; }
    __atomic_store_n(&fast_state, fast_build_and_verify() ? 1 : -1, __ATOMIC_RELEASE);
}
static void fast_init(void) {
    static pthread_once_t once = PTHREAD_ONCE_INIT;
    pthread_once(&once, fast_init_once);
}
static int fast_build_and_verify(void) {
    for (int layer = 0; layer < 256; ++layer) {
        double w = 0.0;
        if (!probe_fast(layer, 1, &w) || !(w > 0.0)) {       /* magnitude 1 leaves the fast branch: not the table we think it is */
            if (!probe_fast(layer, 0, NULL)) { zig_ki[layer] = 0; zig_wi[layer] = 0.0; continue; }
            return 0;
        }
        zig_wi[layer] = w;
        uint64_t lo = 1, hi = (uint64_t)1 << 52;             /* fast at lo; ki in (lo, hi] */
        if (probe_fast(layer, hi - 1, NULL)) { zig_ki[layer] = hi; continue; }
        hi -= 1;                                             /* slow at hi */
        while (hi - lo > 1) {
            const uint64_t mid = lo + ((hi - lo) >> 1);
            if (probe_fast(layer, mid, NULL)) lo = mid; else hi = mid;
        }
        zig_ki[layer] = hi;                                  /* the first magnitude that is not fast */
    }
    /* verification against numpy's routines on identical states */
    pcg64_t a, b;
    bitgen_t bga, bgb;
    pcg64_seed(&a, 0x9e3779b97f4a7c15ULL);
    b = a;
    pcg64_bitgen(&a, &bga);
    pcg64_bitgen(&b, &bgb);
    for (int i = 0; i < (1 << 18); ++i) {
        const double x = fast_standard_normal(&a, &bga), y = random_standard_normal(&bgb);
        if (memcmp(&x, &y, sizeof x) != 0 || a.state != b.state) return 0;
    }
    static const uint32_t ranges[] = {1, 2, 4, 7, 9, 31, 32, 63, 255, 256, 511, 1000, 8191, 65535, 65536, 1000003, 0x7fffffffu, 0xfffffffeu};
    for (size_t q = 0; q < sizeof ranges / sizeof ranges[0]; ++q) {
        uint64_t want[257];
        random_bounded_uint64_fill(&bgb, 0, ranges[q], 257, 0, want);
        for (int i = 0; i < 257; ++i)
            if ((uint64_t)fast_bounded32(&a, ranges[q]) != want[i]) return 0;
        /* an odd number of 32-bit draws leaves a buffered half behind: the two generators must agree on it too */
        if (a.state != b.state || a.has_uint32 != b.has_uint32 || a.uinteger != b.uinteger) return 0;
        const double x = fast_double(&a), y = random_standard_uniform(&bgb);
        if (x != y) return 0;
    }
    /* ... and the same draws read from blocks of outputs made ahead (blk_*), interleaved with numpy's own routines on the shared state */
    {
        pcg_block_t blk;
        blk.pos = blk.n = 0;
        for (int round = 0; round < 64; ++round) {
            for (int i = 0; i < 1000 + 37 * round; ++i) {
                const double x = blk_standard_normal(&a, &blk, &bga), y = random_standard_normal(&bgb);
                if (memcmp(&x, &y, sizeof x) != 0) return 0;
            }
            const uint32_t rng = ranges[round % (sizeof ranges / sizeof ranges[0])];
            uint64_t want[129];
            random_bounded_uint64_fill(&bgb, 0, rng, 129, 0, want);
            for (int i = 0; i < 129; ++i)
                if ((uint64_t)blk_bounded32(&a, &blk, rng) != want[i]) return 0;
            if (blk_double(&a, &blk) != random_standard_uniform(&bgb)) return 0;
            {
                uint64_t got[131], want2[131];
                const int nf = 128 + (round % 3);                       /* (even and odd counts, with and without a half left behind) */
                random_bounded_uint64_fill(&bgb, 0, rng, nf, 0, want2);
                blk_bounded32_fill(&a, &blk, rng, nf, got);
                for (int i = 0; i < nf; ++i)
                    if (got[i] != want2[i]) return 0;
                if (round & 1) { if ((uint64_t)blk_bounded32(&a, &blk, 6) != (random_bounded_uint64_fill(&bgb, 0, 6, 1, 0, want2), want2[0])) return 0; }
            }
            blk_sync(&a, &blk);
            if (a.state != b.state || a.has_uint32 != b.has_uint32 || a.uinteger != b.uinteger) return 0;
            const double x = random_standard_normal(&bga), y = random_standard_normal(&bgb);     /* (numpy's routine on the synchronised state) */
            if (memcmp(&x, &y, sizeof x) != 0 || a.state != b.state) return 0;
        }
    }
    return 1;
}

/* 1 when the inlined draws are in use (verified against numpy's routines in this process), else 0 */
int npbnn_host_fast_predraw(void) {
    fast_init();
    return fast_state == 1;
}

/* ---------------------------------------------------------------------------------------------
 * pre-draw
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_layers;
    int32_t rows[NPBNN_HOST_MAX_LAYERS];
    int32_t cols[NPBNN_HOST_MAX_LAYERS];      /* including the bias column */
    int32_t w_off[NPBNN_HOST_MAX_LAYERS];     /* offset of the layer in the packed weight vector */
    int32_t update_n[NPBNN_HOST_MAX_LAYERS];  /* MCMC._update_n */
    const double* update_ws[NPBNN_HOST_MAX_LAYERS]; /* MCMC._update_ws[i], rows*cols doubles */
    double freq_layer_update[NPBNN_HOST_MAX_LAYERS];
    int32_t ws_uniform[NPBNN_HOST_MAX_LAYERS]; /* 1: every entry of update_ws[i] is the same number (the sampler's default: a scalar per layer,
                                                  np_bnn/BNN_env.py:291) - the step size is then read once, not per drawn entry: on a layer of a
                                                  million weights that read is a cache miss per entry */
} npbnn_proposal_spec;

int npbnn_host_abi_version(void) { return 1; }

/* Seeds a PCG64 like np.random.default_rng(seed) and returns its first `n` doubles (self-test hook). */
int npbnn_host_selftest_doubles(uint64_t seed, int n, double* out) {
    pcg64_t r;
    bitgen_t bg;
    pcg64_seed(&r, seed);
    pcg64_bitgen(&r, &bg);
    random_standard_uniform_fill(&bg, n, out);
    return 0;
}

/*
 * Draw the proposals of K consecutive iterations.
 *   bitgen            the Generator's bitgen_t* (Generator.bit_generator.ctypes.bit_generator); used when
 *                     randomize_seed == 0 and advanced in place
 *   randomize_seed    1: iteration t uses default_rng(first_iteration + t + mcmc_id)   (MC3 chains)
 *   max_per_iter      M = capacity per iteration of idx/delta (>= sum of update_n)
 * Outputs (row t = iteration t):
 *   idx[t*M + j]      flat index into the packed weights, or -1 for an entry superseded by a later draw of
 *                     the same position (numpy fancy-index assignment: the last write wins)
 *   delta[t*M + j]    the normal deviate to add
 *   cnt[t]            entries used in row t
 *   log_u[t]          the uniform draw of the accept test (the caller takes np.log, as the reference does)
 *   layer_mask[t]     bit i set when layer i was updated
 * Returns 0, or -1 on bad arguments / capacity.
 */
int npbnn_host_predraw2(void* bitgen, int randomize_seed, int64_t first_iteration, int64_t mcmc_id, int K,
                        const npbnn_proposal_spec* spec, int max_per_iter, int32_t* idx, double* delta,
                        int32_t* cnt, double* log_u, int32_t* layer_mask, int n_weights, int sigma_k, double sigma_f,
                        double* sigma_chosen, double* sigma_u);

int npbnn_host_predraw3(void* bitgen, int randomize_seed, int64_t first_iteration, int64_t mcmc_id, int K,
                        const npbnn_proposal_spec* spec, int max_per_iter, int32_t* idx, double* delta,
                        int32_t* cnt, double* log_u, int32_t* layer_mask, int n_weights, int sigma_k, double sigma_f,
                        double* sigma_chosen, double* sigma_u, int n_slopes, double slope_d, int32_t* slope_idx,
                        double* slope_delta);

int npbnn_host_predraw(void* bitgen, int randomize_seed, int64_t first_iteration, int64_t mcmc_id, int K,
                       const npbnn_proposal_spec* spec, int max_per_iter, int32_t* idx, double* delta,
                       int32_t* cnt, double* log_u, int32_t* layer_mask, int n_weights) {
    return npbnn_host_predraw2(bitgen, randomize_seed, first_iteration, mcmc_id, K, spec, max_per_iter, idx, delta, cnt, log_u,
                               layer_mask, n_weights, 0, 0.0, NULL, NULL);
}

/* The same with the draws of the regression error-parameter proposal (multiplier_proposal_vector, np_bnn/BNN_mcmc.py:101-113,
 * called before the weight proposals of an iteration, BNN_env.py:435-442) when sigma_k > 0:
 *   sigma_chosen[t*sigma_k + q]   rs.binomial(1, sigma_f, sigma_k)   (1.0 / 0.0)
 *   sigma_u[t*sigma_k + q]        rs.random(sigma_k)
 * The caller forms the multipliers and the Hastings term with numpy, as the reference does. */
int npbnn_host_predraw2(void* bitgen, int randomize_seed, int64_t first_iteration, int64_t mcmc_id, int K,
                        const npbnn_proposal_spec* spec, int max_per_iter, int32_t* idx, double* delta,
                        int32_t* cnt, double* log_u, int32_t* layer_mask, int n_weights, int sigma_k, double sigma_f,
                        double* sigma_chosen, double* sigma_u) {
    return npbnn_host_predraw3(bitgen, randomize_seed, first_iteration, mcmc_id, K, spec, max_per_iter, idx, delta, cnt, log_u,
                               layer_mask, n_weights, sigma_k, sigma_f, sigma_chosen, sigma_u, 0, 0.0, NULL, NULL);
}

/* The same with the draws of the trainable activation slopes (UpdateNormal1D(acc_prm, d, n=1), np_bnn/BNN_mcmc.py:44-55, the FIRST
 * draws of an iteration, BNN_env.py:416-421) when n_slopes > 0:
 *   slope_idx[t]     rs.integers(0, n_slopes, 1)   (no draw when n_slopes == 1: numpy returns the only value)
 *   slope_delta[t]   rs.normal(0, slope_d, 1) */
/* own_state: NULL (the draws come from `bitgen`, numpy's live generator, or - randomize_seed - from a generator seeded per iteration), or
 * the generator to draw from and leave advanced (npbnn_host_predraw_state).  Wherever the generator is one of this file's own
 * (own_state, or the per-iteration ones) and the inlined draws have been verified (fast_init), they are used. */
static int predraw_core(void* bitgen, pcg64_t* own_state, int randomize_seed, int64_t first_iteration, int64_t mcmc_id, int K,
                        const npbnn_proposal_spec* spec, int max_per_iter, int32_t* idx, double* delta,
                        int32_t* cnt, double* log_u, int32_t* layer_mask, int n_weights, int sigma_k, double sigma_f,
                        double* sigma_chosen, double* sigma_u, int n_slopes, double slope_d, int32_t* slope_idx,
                        double* slope_delta) {
    if (!spec || K < 0 || spec->n_layers < 1 || spec->n_layers > NPBNN_HOST_MAX_LAYERS) return -1;
    if (sigma_k < 0 || (sigma_k > 0 && (!sigma_chosen || !sigma_u))) return -1;
    if (n_slopes < 0 || (n_slopes > 0 && (!slope_idx || !slope_delta))) return -1;
    if (!randomize_seed && !bitgen && !own_state) return -1;
    fast_init();
    const int fast_ok = fast_state == 1;
    int total = 0, max_n = 0;
    for (int i = 0; i < spec->n_layers; ++i) {
        total += spec->update_n[i];
        if (spec->update_n[i] > max_n) max_n = spec->update_n[i];
    }
    if (total > max_per_iter) return -1;
    int32_t* last = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_weights);
    uint64_t* ix = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(max_n > 0 ? max_n : 1) * 2);
    if (!last || !ix) {
        free(last);
        free(ix);
        return -2;
    }
    uint64_t* iy = ix + (max_n > 0 ? max_n : 1);
    for (int i = 0; i < n_weights; ++i) last[i] = -1;
    int64_t stamp = 0;          /* entries get running numbers in the last-writer table: a slot at or above a layer's first number was
                                   written by that layer in this iteration - nothing to wipe between layers (wrapped: the table starts over) */
    pcg64_t local;
    bitgen_t local_bg;
    pcg_block_t blk;            /* outputs made ahead for the inlined draws (empty between two iterations) */
    blk.pos = blk.n = 0;
    if (own_state) pcg64_bitgen(own_state, &local_bg);
    for (int t = 0; t < K; ++t) {
        bitgen_t* bg = own_state ? &local_bg : (bitgen_t*)bitgen;
        pcg64_t* fr = (own_state && fast_ok) ? own_state : NULL;      /* the generator of the inlined draws, when they apply */
        if (randomize_seed) {
            pcg64_seed(&local, (uint64_t)(first_iteration + t + mcmc_id));
            pcg64_bitgen(&local, &local_bg);
            bg = &local_bg;
            fr = fast_ok ? &local : NULL;
        }
        if (n_slopes > 0) {
            uint64_t pick = 0;
            random_bounded_uint64_fill(bg, 0, (uint64_t)(n_slopes - 1), 1, 0, &pick);
            slope_idx[t] = (int32_t)pick;
            slope_delta[t] = random_normal(bg, 0.0, slope_d);
        }
        if (sigma_k > 0) {
            binomial_t binom;
            memset(&binom, 0, sizeof binom);
            for (int q = 0; q < sigma_k; ++q) sigma_chosen[(size_t)t * sigma_k + q] = (double)random_binomial(bg, sigma_f, 1, &binom);
            random_standard_uniform_fill(bg, sigma_k, sigma_u + (size_t)t * sigma_k);
        }
        double rr[NPBNN_HOST_MAX_LAYERS];
        if (fr) { for (int i = 0; i < spec->n_layers; ++i) rr[i] = blk_double(fr, &blk); }
        else random_standard_uniform_fill(bg, spec->n_layers, rr);
        int amin = 0;
        for (int i = 1; i < spec->n_layers; ++i)
            if (rr[i] < rr[amin]) amin = i;
        rr[amin] = 0.0;
        int32_t* row_idx = idx + (size_t)t * max_per_iter;
        double* row_delta = delta + (size_t)t * max_per_iter;
        int used = 0, mask = 0;
        for (int i = 0; i < spec->n_layers; ++i) {
            if (!(rr[i] < spec->freq_layer_update[i])) continue;
            mask |= 1 << i;
            const int n = spec->update_n[i];
            if (fr) {      /* (a range of one value draws nothing, as in numpy's fill) */
                const uint32_t rmax = (uint32_t)(spec->rows[i] - 1), cmax = (uint32_t)(spec->cols[i] - 1);
                if (rmax == 0) { for (int j = 0; j < n; ++j) ix[j] = 0; } else blk_bounded32_fill(fr, &blk, rmax, n, ix);
                if (cmax == 0) { for (int j = 0; j < n; ++j) iy[j] = 0; } else blk_bounded32_fill(fr, &blk, cmax, n, iy);
            } else {
                random_bounded_uint64_fill(bg, 0, (uint64_t)(spec->rows[i] - 1), n, 0, ix);
                random_bounded_uint64_fill(bg, 0, (uint64_t)(spec->cols[i] - 1), n, 0, iy);
            }
            const int base = used;
            const int uniform = spec->ws_uniform[i];
            const double scale0 = spec->update_ws[i][0];
            const int cols_i = spec->cols[i], off_i = spec->w_off[i];
            if (stamp + n >= 0x7fffffff) {                      /* (never in practice: 2^31 entries in one call) */
                for (int q = 0; q < n_weights; ++q) last[q] = -1;
                stamp = 0;
            }
            const int s0 = (int)stamp;
            stamp += n;
            if (fr && uniform) {
                /* the sampler's default (one step size per layer) with the inlined draws: the deviates in a loop of their own - nothing in it
                 * waits for the last-writer table - then the table's loop with its slots requested well ahead */
                for (int j = 0; j < n; ++j) {
                    row_idx[base + j] = off_i + (int)ix[j] * cols_i + (int)iy[j];
                    row_delta[base + j] = 0.0 + scale0 * blk_standard_normal(fr, &blk, bg);
                }
                for (int j = 0; j < n; ++j) {
                    if (j + 16 < n) __builtin_prefetch(&last[row_idx[base + j + 16]], 1, 1);
                    const int flat = row_idx[base + j];
                    if (last[flat] >= s0) row_idx[base + (last[flat] - s0)] = -1;   /* superseded within this layer */
                    last[flat] = s0 + j;
                }
            } else
            for (int j = 0; j < n; ++j) {
                if (j + 8 < n) {      /* (the entry's slot of the last-writer table, requested ahead: 4 bytes per weight, out of cache on wide layers) */
                    const int ahead = off_i + (int)ix[j + 8] * cols_i + (int)iy[j + 8];
                    __builtin_prefetch(&last[ahead], 1, 1);
                    if (!uniform) __builtin_prefetch(&spec->update_ws[i][ahead - off_i], 0, 1);
                }
                const int local_pos = (int)ix[j] * spec->cols[i] + (int)iy[j];
                const double scale = uniform ? scale0 : spec->update_ws[i][local_pos];
                const int flat = spec->w_off[i] + local_pos;
                /* random_normal(loc, scale) = loc + scale * standard normal */
                row_delta[base + j] = fr ? 0.0 + scale * blk_standard_normal(fr, &blk, bg) : random_normal(bg, 0.0, scale);
                if (last[flat] >= s0) row_idx[base + (last[flat] - s0)] = -1;   /* superseded within this layer */
                row_idx[base + j] = flat;
                last[flat] = s0 + j;
            }
            used += n;
        }
        for (int j = used; j < max_per_iter; ++j) { row_idx[j] = -1; row_delta[j] = 0.0; }      /* (the caller's arrays come uninitialised) */
        cnt[t] = used;
        layer_mask[t] = mask;
        log_u[t] = fr ? blk_double(fr, &blk) : random_standard_uniform(bg);
        if (fr) blk_sync(fr, &blk);      /* (the next iteration may start with numpy's own routines, or with a generator of its own) */
    }
    free(last);
    free(ix);
    return 0;
}

int npbnn_host_predraw3(void* bitgen, int randomize_seed, int64_t first_iteration, int64_t mcmc_id, int K,
                        const npbnn_proposal_spec* spec, int max_per_iter, int32_t* idx, double* delta,
                        int32_t* cnt, double* log_u, int32_t* layer_mask, int n_weights, int sigma_k, double sigma_f,
                        double* sigma_chosen, double* sigma_u, int n_slopes, double slope_d, int32_t* slope_idx,
                        double* slope_delta) {
    return predraw_core(bitgen, NULL, randomize_seed, first_iteration, mcmc_id, K, spec, max_per_iter, idx, delta, cnt, log_u, layer_mask,
                        n_weights, sigma_k, sigma_f, sigma_chosen, sigma_u, n_slopes, slope_d, slope_idx, slope_delta);
}

/* The same for a chain that draws from ONE generator (randomize_seed == 0), given by value: gen = {state high, state low, increment
 * high, increment low, has_uint32, uinteger} - the integers of numpy's PCG64 state dictionary (Generator.bit_generator.state) - read
 * at entry, written back advanced.  The draws are numpy's to the bit (the inlined forms where verified, numpy's routines on this
 * file's PCG64 otherwise); nothing of numpy's own structs is touched. */
int npbnn_host_predraw_state(uint64_t* gen, int64_t first_iteration, int64_t mcmc_id, int K, const npbnn_proposal_spec* spec,
                             int max_per_iter, int32_t* idx, double* delta, int32_t* cnt, double* log_u, int32_t* layer_mask,
                             int n_weights, int sigma_k, double sigma_f, double* sigma_chosen, double* sigma_u, int n_slopes,
                             double slope_d, int32_t* slope_idx, double* slope_delta) {
    if (!gen) return -1;
    pcg64_t g;
    g.state = (((u128)gen[0]) << 64) | (u128)gen[1];
    g.inc = (((u128)gen[2]) << 64) | (u128)gen[3];
    g.has_uint32 = (int)gen[4];
    g.uinteger = (uint32_t)gen[5];
    const int rc = predraw_core(NULL, &g, 0, first_iteration, mcmc_id, K, spec, max_per_iter, idx, delta, cnt, log_u, layer_mask, n_weights,
                                sigma_k, sigma_f, sigma_chosen, sigma_u, n_slopes, slope_d, slope_idx, slope_delta);
    gen[0] = (uint64_t)(g.state >> 64); gen[1] = (uint64_t)g.state;
    gen[2] = (uint64_t)(g.inc >> 64); gen[3] = (uint64_t)g.inc;
    gen[4] = (uint64_t)g.has_uint32; gen[5] = (uint64_t)g.uinteger;
    return rc;
}
