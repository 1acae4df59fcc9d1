/*
 * mtg_bubble.h -- MonumentTraversal::explore_branching served by a GROUP of lanes with its work areas in LDS.
 *
 * The reference resolves a branching node with a bounded breadth-first search (Frontline / FrontlineBranching), a path enumeration and a
 * validation of the consensuses (gatb-core, absent; call site /root/reference/src/Filler.cpp:866-867,884).  The walk
 * kernel (one gap per lane) answers simple paths and the strict SNP pattern itself; a gap whose walk meets any other branching node is
 * PARKED and handed to the kernel that hosts this file (k_finish): G lanes (a whole 64-lane wave, or an aligned part of one) serve ONE gap:
 *   - frontier expansion: the nodes of the frontline are read by different lanes at once, their 4-way successor enumeration is one lane
 *     per (node, nucleotide), duplicates and survivors are settled with ballots and the next frontline is compacted by prefix popcounts;
 *   - the visited sets, both frontlines, the involved list, the path enumeration's frames and the consensuses live in LDS (BubbleLds,
 *     ~5 KB per gap) instead of the per-gap scratch in HBM that the one-lane form (mtg_traverse.h) works from;
 *   - the branching tests that decide which involved nodes get marked are made by all lanes at once.
 * Semantics are those of the one-lane routines, which stay the reference form (and the fallback for anything that does not fit the LDS
 * areas: the solver then answers "too big" and the group runs the general code from HBM scratch).  The order in which a frontline is
 * processed is only observable where the one-lane form says so (first-come orientation of a node reached twice in one level; the nodes
 * left when a backward frontline meets a marked node), and there the group reproduces the sequential order exactly.  The TEST-ONLY
 * emulation build (one lane per group) runs the general code next to every answer of this file and compares consensus and marks (0xBADC).
 *
 * Included by mtg_traverse.h (between the general bubble code and stage_a_gap).
 */
#ifndef MTG_BUBBLE_H
#define MTG_BUBBLE_H
#ifndef MTG_TRAVERSE_H
#error "include mtg_traverse.h"
#endif

namespace mtg {

/* capacities of the LDS form (anything larger is answered "too big").  The area of a gap is what limits how many gaps a compute unit serves
 * at a time (160 KB of LDS): the bubble kernel of the rounds, which may have every gap of a launch before it, takes the small ones (the bubbles
 * of heterozygous data: a frontline of two or three nodes, a few dozen nodes seen), the finishing kernel, with a few gaps left, the large ones. */
struct CapsSmall { enum { SEEN = 64, FL = 12, INV = 24, ISEEN = 32, IFL = 6, FR = 16, NT = 96, CONS = 4, CLEN = 96 }; };
struct CapsLarge { enum { SEEN = 256, FL = 24, INV = 96, ISEEN = 64, IFL = 12, FR = 32, NT = 128, CONS = 8, CLEN = 128 }; };

struct FlNode { /* a frontline node: oriented k-mer, its place in the unitig store (rp_pack, 0 = unknown), nodes ahead in its unitig, node_aux */
    uint64_t f, rp;
    uint32_t ra, aux;
};
struct FlExp { /* what a frontline node leads to (one index read, or nothing to read at all) */
    uint64_t krp;
    uint32_t kra, kid, out, chk;
};
struct DfsFrame {
    uint64_t f, c, rp;
    uint32_t ra, dep, xsn, kid, mask, pad_;
};
template <class C> struct BubbleLdsT {
    typedef C Caps;
    uint64_t inv[C::INV];     /* involved nodes that are not known to be simple (canonical k-mers): candidates for marking; once the bubble is
                                 answered, its first n_marks entries are the nodes to mark (the plan; applied by coop_apply_marks) */
    uint8_t invbr[C::INV];
    uint32_t n_marks, pad_;
    union {
        struct { /* find_end_of_branching */
            uint64_t seen[C::SEEN];
            FlNode fl[2][C::FL];
            FlExp ex[C::FL];
            uint64_t iseen[C::ISEEN];
            FlNode ifl[2][C::IFL];
            FlExp iex[C::IFL];
        } a;
        struct { /* all_consensuses_between + validate_consensuses */
            uint64_t pset[C::ISEEN];
            DfsFrame fr[C::FR];
            uint8_t nt[C::NT + 16];
            uint8_t cons[C::CONS][C::CLEN];
            uint16_t len[C::CONS];
            int32_t sum[C::CONS];
        } b;
    };
};
typedef BubbleLdsT<CapsSmall> BubbleLds;    /* k_bubble */
typedef BubbleLdsT<CapsLarge> BubbleLdsBig; /* k_finish */

/* ---- a group of lanes: G consecutive lanes of a wave (G a power of two, 64 = the whole wave).  Control flow inside the routines below is
 * uniform over the group; the emulation build has one lane per group. */
#if defined(MTG_EMU) && defined(MTG_EMU_LANES)
} // namespace mtg
#include <pthread.h>
#include <dlfcn.h>
#include <time.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
namespace mtg {
/* TEST-ONLY (tests/emu, -DMTG_EMU_LANES=N): the group form executed by N lanes IN LOCK STEP.  Every lane is a host thread with its own copy
 * of the per-lane state (the Worker, the locals of the routines); the group's collectives -- ballot, shuffle, minimum, sync -- are
 * rendezvous of all N, so the lanes advance from collective to collective together, as the lanes of a wave advance from instruction to
 * instruction.  Between two collectives the lanes touch shared memory (the LDS areas) only at places the code assigns to them by lane
 * number, or through the atomic compare-and-swap of the sets: what ballot compaction, chunk-wise de-duplication and the prefix popcounts
 * compute with several lanes is then really computed by several lanes, under AddressSanitizer / UBSan / ThreadSanitizer.  A rendezvous
 * that not all lanes reach (divergent control flow around a collective -- a hang on the device) is a deadlock here: the watchdog of the
 * test reports it.  emu_group_run(f) runs f(lane) on the N lanes and returns lane 0's value; the lanes must return the same value. */
struct EmuLanes {
    enum { N = MTG_EMU_LANES };
    pthread_mutex_t m = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t c = PTHREAD_COND_INITIALIZER;
    int arrived = 0;
    unsigned long gen = 0;
    uint64_t slot[N];
    int ret[N];
    void* site[N];   /* where every lane last entered a collective: printed when a rendezvous is not reached by all */
    int done[N];     /* the lane has left the group's routine */
    EmuLanes() { for (int j = 0; j < N; j++) { site[j] = nullptr; done[j] = 0; ret[j] = 0; slot[j] = 0; } }
    /* a rendezvous of the N lanes.  A lane that has RETURNED while others still wait, or a wait of 20 s, is a divergence: reported and fatal
     * (on the device: a hang, or lanes computing with another lane's stale values) */
    void wait(void* from)
    {
        pthread_mutex_lock(&m);
        site[emu_lane_of()] = from;
        const unsigned long my = gen;
        if (++arrived == N) { arrived = 0; gen++; pthread_cond_broadcast(&c); }
        else {
            struct timespec ts;
            clock_gettime(CLOCK_REALTIME, &ts);
            ts.tv_sec += 20;
            while (gen == my) {
                int finished = 0;
                for (int j = 0; j < N; j++) finished += done[j];
                if (finished || pthread_cond_timedwait(&c, &m, &ts) != 0) {
                    if (gen != my) break;
                    fprintf(stderr, "[emu lanes] a rendezvous of the group was not reached by all %d lanes (%d returned):", (int)N, finished);
                    for (int j = 0; j < N; j++) {
                        Dl_info di;
                        const bool ok = site[j] && dladdr(site[j], &di) && di.dli_fbase;
                        fprintf(stderr, " lane %d %s at +0x%lx;", j, done[j] ? "returned, last" : "waits/last", ok ? (unsigned long)((char*)site[j] - (char*)di.dli_fbase) : 0ul);
                    }
                    fprintf(stderr, "\n");
                    abort();
                }
            }
        }
        pthread_mutex_unlock(&m);
    }
    void leave(uint32_t lane) { pthread_mutex_lock(&m); done[lane] = 1; pthread_cond_broadcast(&c); pthread_mutex_unlock(&m); }
    static uint32_t emu_lane_of();
};
inline thread_local EmuLanes* emu_group = nullptr; /* null: outside a group run, the code is one lane */
inline thread_local uint32_t emu_lane = 0;
inline uint32_t EmuLanes::emu_lane_of() { return emu_lane; }
template <int G> struct Grp {
    enum { N = MTG_EMU_LANES };
    static uint32_t gl() { return emu_lane; }
    __attribute__((noinline)) static uint64_t ballot(bool p)
    {
        EmuLanes* g = emu_group;
        if (!g) return p ? 1ull : 0ull;
        g->slot[emu_lane] = p ? 1ull : 0ull;
        g->wait(__builtin_return_address(0));
        uint64_t b = 0;
        for (int j = 0; j < N; j++) b |= g->slot[j] << j;
        g->wait(__builtin_return_address(0));
        return b;
    }
    __attribute__((always_inline)) static bool any(bool p) { return ballot(p) != 0ull; }
    __attribute__((always_inline)) static bool all(bool p) { return ballot(!p) == 0ull; }
    __attribute__((noinline)) static uint32_t min32(uint32_t v)
    {
        EmuLanes* g = emu_group;
        if (!g) return v;
        g->slot[emu_lane] = v;
        g->wait(__builtin_return_address(0));
        uint32_t m = v;
        for (int j = 0; j < N; j++) if ((uint32_t)g->slot[j] < m) m = (uint32_t)g->slot[j];
        g->wait(__builtin_return_address(0));
        return m;
    }
    __attribute__((noinline)) static uint64_t from_lane64(uint64_t v, int j)
    {
        EmuLanes* g = emu_group;
        if (!g) return v;
        g->slot[emu_lane] = v;
        g->wait(__builtin_return_address(0));
        const uint64_t r = g->slot[j];
        g->wait(__builtin_return_address(0));
        return r;
    }
    __attribute__((noinline)) static void sync() { if (emu_group) emu_group->wait(__builtin_return_address(0)); }
    /* an operation every lane of a wave performs identically on the SAME shared words (read, decide, write: consistent because the lanes of a
     * wave execute each instruction together): threads that are not in lock step between collectives cannot do that, so lane 0 performs
     * it between two rendezvous and the others take its result */
    template <typename F> __attribute__((noinline)) static int uniform(F f)
    {
        EmuLanes* g = emu_group;
        if (!g) return f();
        g->wait(__builtin_return_address(0));
        if (emu_lane == 0) g->slot[0] = (uint64_t)(int64_t)f();
        g->wait(__builtin_return_address(0));
        const int r = (int)(int64_t)g->slot[0];
        g->wait(__builtin_return_address(0));
        return r;
    }
};
template <typename F> int emu_group_run(F f)
{
    EmuLanes g;
    std::vector<std::thread> ts;
    for (uint32_t lane = 1; lane < (uint32_t)EmuLanes::N; lane++)
        ts.emplace_back([&g, &f, lane] { emu_group = &g; emu_lane = lane; g.ret[lane] = f(lane); g.leave(lane); emu_group = nullptr; emu_lane = 0; });
    emu_group = &g; emu_lane = 0;
    g.ret[0] = f(0u);
    g.leave(0);
    emu_group = nullptr;
    for (auto& t : ts) t.join();
    for (int j = 1; j < EmuLanes::N; j++) if (g.ret[j] != g.ret[0]) return -0x7BADD; /* the lanes disagree about a uniform value */
    return g.ret[0];
}
MTG_DEV int popc64(uint64_t x) { return __builtin_popcountll(x); }
MTG_DEV int ctz64(uint64_t x) { return __builtin_ctzll(x); }
MTG_DEV uint64_t lds_cas64(uint64_t* p, uint64_t cmp, uint64_t val) { return __sync_val_compare_and_swap(p, cmp, val); }
#elif defined(MTG_EMU)
template <int G> struct Grp {
    enum { N = 1 };
    static uint32_t gl() { return 0; }
    static uint64_t ballot(bool p) { return p ? 1ull : 0ull; }
    static bool any(bool p) { return p; }
    static bool all(bool p) { return p; }
    static uint32_t min32(uint32_t v) { return v; }
    static uint64_t from_lane64(uint64_t v, int) { return v; }
    static void sync() {}
    template <typename F> static int uniform(F f) { return f(); }
};
MTG_DEV int popc64(uint64_t x) { return __builtin_popcountll(x); }
MTG_DEV int ctz64(uint64_t x) { return __builtin_ctzll(x); }
MTG_DEV uint64_t lds_cas64(uint64_t* p, uint64_t cmp, uint64_t val) { const uint64_t o = *p; if (o == cmp) *p = val; return o; }
#else
template <int G> struct Grp {
    enum { N = G };
    MTG_DEV static uint32_t lane() { return threadIdx.x & 63u; }
    MTG_DEV static uint32_t gl() { return lane() & (uint32_t)(G - 1); }
    MTG_DEV static uint64_t ballot(bool p)
    {
        const uint64_t b = __ballot(p);
        if (G == 64) return b;
        return (b >> (lane() & ~(uint32_t)(G - 1))) & ((1ull << (G & 63)) - 1ull);
    }
    MTG_DEV static bool any(bool p) { return ballot(p) != 0ull; }
    MTG_DEV static bool all(bool p) { return ballot(!p) == 0ull; }
    MTG_DEV static uint32_t min32(uint32_t v)
    {
        for (int m = G / 2; m >= 1; m >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)v, m, 64); v = y < v ? y : v; }
        return v;
    }
    /* the value lane j of the group holds */
    MTG_DEV static uint64_t from_lane64(uint64_t v, int j)
    {
        const int src = (int)(lane() & ~(uint32_t)(G - 1)) + j;
        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
        return ((uint64_t)hi << 32) | lo;
    }
    /* what the lanes of the group wrote to their LDS area is visible to its other lanes: they are lanes of one wave, whose LDS operations
     * execute in order; the fences keep the compiler from moving the accesses */
    MTG_DEV static void sync()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    /* an operation all lanes perform identically on the same LDS words (the lanes of a wave execute every instruction together: reads before
     * writes, for all of them).  Named so that the test build with one host thread per lane can let one lane perform it (see EmuLanes). */
    template <typename F> MTG_DEV static int uniform(F f) { return f(); }
};
MTG_DEV int popc64(uint64_t x) { return __popcll(x); }
MTG_DEV int ctz64(uint64_t x) { return __ffsll((long long)x) - 1; }
MTG_DEV uint64_t lds_cas64(uint64_t* p, uint64_t cmp, uint64_t val) { return (uint64_t)atomicCAS(reinterpret_cast<unsigned long long*>(p), (unsigned long long)cmp, (unsigned long long)val); }
#endif

/* open-addressing sets in LDS (canonical k-mer + 1, 0 = empty); the capacity tests are the callers' (fill at most 3/4) */
MTG_DEV bool lset_has(const uint64_t* tab, uint32_t cap, uint64_t c)
{
    uint32_t h = set_hash(c, cap);
    for (uint32_t probe = 0; probe < cap; probe++) {
        const uint64_t v = tab[h];
        if (v == 0) return false;
        if (v == c + 1) return true;
        h = (h + 1) & (cap - 1);
    }
    return false;
}
/* several lanes of a group may insert (different keys) at once */
MTG_DEV void lset_insert(uint64_t* tab, uint32_t cap, uint64_t c)
{
    uint32_t h = set_hash(c, cap);
    for (uint32_t probe = 0; probe < cap; probe++) {
        const uint64_t v = lds_cas64(&tab[h], 0ull, c + 1);
        if (v == 0 || v == c + 1) return;
        h = (h + 1) & (cap - 1);
    }
}

/* answers of the group form besides a consensus length: rejected (the one-lane form would return 0), or too big for the LDS areas -- any
 * negative value, the value says which area (diagnostics) */
enum { COOP_FAIL = 0, COOP_TOOBIG = -1, COOP_BIG_CHECK = -2, COOP_BIG_DEPTH = -3, COOP_BIG_NCONS = -4, COOP_BIG_FRAMES = -5, COOP_BIG_PATH = -6, COOP_BIG_ALIGN = -7, COOP_BIG_RULE = -8 };

/* [MEM] gatb FrontlineBranching::check for the frontline node mf, by the group (one-lane form: fl_check).  1 = passes, 0 = large
 * in-branching, COOP_TOOBIG.  n_inv: the involved list's length (uniform). */
template <int G, class LT> MTG_DEV_NOINLINE int coop_fl_check(Worker& W, LT& L, uint64_t mf, uint32_t& n_inv)
{
    typedef Grp<G> GP;
    const int k = W.k;
    const UStore& us = W.ix.us;
    const uint32_t gl = GP::gl();
    const Kmer m = make_kmer(mf, k);
    const Adj l = adj_left(W.ix, m, W.mk1, W.lines);
    if (popc4(l.in) == 1) return 1;
    for (uint32_t em = l.in & 15u; em; em &= em - 1u) {
        const uint32_t nt = low_nt(em);
        const Kmer b = kmer_prev(m, nt, k, W.mk);
        if (lset_has(L.a.seen, LT::Caps::SEEN, canon(b))) continue;
        for (uint32_t i = gl; i < (uint32_t)LT::Caps::ISEEN; i += GP::N) L.a.iseen[i] = 0;
        GP::sync();
        if (gl == 0) {
            lset_insert(L.a.iseen, LT::Caps::ISEEN, canon(b));
            lset_insert(L.a.iseen, LT::Caps::ISEEN, canon(m));
            FlNode s;
            s.f = b.f; s.rp = 0; s.ra = 0; s.aux = 0;
            L.a.ifl[0][0] = s;
        }
        uint32_t n_iseen = 2;
        GP::sync();
        int cur = 0, ncur = 1, depth = 0, remaining = 0;
        MTG_GUARD_DECL(g1);
        for (;;) {
            MTG_GUARD(g1, 100000u, 10, return COOP_TOOBIG);
            FlNode* cf = L.a.ifl[cur];
            FlNode* nf = L.a.ifl[cur ^ 1];
            /* the skip: every node has two or more nodes of its unitig behind it, in pairwise different unitigs */
            if (us.nwords && ncur >= 1 && depth > 0) {
                bool ok_l = true;
                uint32_t d_l = 0xFFFFFFFFu;
                for (int i = (int)gl; i < ncur; i += GP::N) {
                    const uint32_t ra = cf[i].ra;
                    if (!(cf[i].rp & RP_VALID) || ra < 2u) ok_l = false;
                    else if (ra - 1u < d_l) d_l = ra - 1u;
                    for (int j = 0; j < i && ok_l; j++) if (rp_unitig(cf[j].rp) == rp_unitig(cf[i].rp)) ok_l = false;
                }
                if (GP::all(ok_l)) {
                    const uint32_t D = GP::min32(d_l);
                    if ((uint32_t)depth + D > 3u * (uint32_t)k) { remaining = ncur; break; }
                    for (int i = (int)gl; i < ncur; i += GP::N) {
                        const uint64_t rp = cf[i].rp;
                        cf[i].f = run_node(us, rp & RP_KPOS, (rp & RP_BWD) != 0, D, k).r; /* the store walk is that of the reverse complement */
                        cf[i].rp = rp_step(rp, D);
                        cf[i].ra -= D;
                    }
                    W.lines += (uint32_t)ncur;
                    depth += (int)D;
                    remaining = ncur;
                    GP::sync();
                    continue;
                }
            }
            /* what every node leads back to: the reads of a level are in flight together */
            for (int i = (int)gl; i < ncur; i += GP::N) {
                const Kmer x = make_kmer(cf[i].f, k);
                FlExp e;
                e.krp = 0; e.kra = 0; e.kid = 0; e.chk = 0;
                const uint64_t rp = cf[i].rp;
                if ((rp & RP_VALID) && cf[i].ra >= 1u) {
                    e.out = 1u << (run_next_nt(us, rp & RP_KPOS, (rp & RP_BWD) != 0, k) ^ 2u); /* the reverse complement's next nucleotide, complemented */
                    e.krp = rp_step(rp, 1);
                    e.kra = cf[i].ra - 1u;
                } else {
                    Kmer xr;
                    xr.f = x.r; xr.r = x.f;
                    const Adj ar = adj_right_t(W.ix.adj, xr, W.mk1, W.lines);
                    e.out = comp_mask(ar.out);
                    RunAt r;
                    if (us.nwords && run_at(us, ar, k, r, W.lines)) { e.krp = rp_step(rp_pack(r), 1); e.kra = r.ahead - 1u; }
                }
                L.a.iex[i] = e;
            }
            GP::sync();
            /* the predecessors, node by node in frontline order (the order decides what is left when a marked node ends the search); the four
             * candidates of a node are looked at together */
            bool cont = true;
            int nnext = 0;
            for (int i = 0; i < ncur && cont; i++) {
                const Kmer x = make_kmer(cf[i].f, k);
                const FlExp e = L.a.iex[i];
                for (uint32_t base = 0; base < 4u && cont; base += GP::N) {
                    const uint32_t n2 = base + gl;
                    bool cand = false;
                    Kmer y;
                    y.f = y.r = 0;
                    uint64_t cy = 0;
                    if (n2 < 4u && (e.out & (1u << n2))) {
                        y = kmer_prev(x, n2, k, W.mk);
                        cy = canon(y);
                        cand = !lset_has(L.a.iseen, LT::Caps::ISEEN, cy);
                        /* a sibling with a smaller nucleotide and the same canonical k-mer comes first */
                        for (uint32_t n1 = 0; n1 < n2 && cand; n1++)
                            if ((e.out & (1u << n1)) && canon(kmer_prev(x, n1, k, W.mk)) == cy) cand = false;
                    }
                    const bool hit = cand && W.is_marked(cy);
                    const uint64_t hb = GP::ballot(hit), cb = GP::ballot(cand);
                    uint64_t take = cb;
                    if (hb) { take &= (1ull << ctz64(hb)) - 1ull; cont = false; remaining = ncur - i - 1; }
                    const uint64_t below = take & ((1ull << gl) - 1ull);
                    if ((take >> gl) & 1ull) {
                        const int pos = nnext + popc64(below);
                        if (pos < LT::Caps::IFL) { FlNode s; s.f = y.f; s.rp = e.krp; s.ra = e.kra; s.aux = 0; nf[pos] = s; }
                        lset_insert(L.a.iseen, LT::Caps::ISEEN, cy);
                    }
                    /* involved: only the nodes not known to be simple are remembered */
                    const bool simple = (e.krp & RP_VALID) && e.kra >= 1u;
                    const uint64_t ib = simple ? 0ull : take;
                    if ((ib >> gl) & 1ull) {
                        const uint32_t ipos = n_inv + (uint32_t)popc64(ib & ((1ull << gl) - 1ull));
                        if (ipos < (uint32_t)LT::Caps::INV) L.inv[ipos] = cy;
                    }
                    nnext += popc64(take);
                    n_iseen += (uint32_t)popc64(take);
                    n_inv += (uint32_t)popc64(ib);
                    GP::sync();
                }
                if (n_iseen > (uint32_t)LT::Caps::ISEEN * 3u / 4u || n_inv > (uint32_t)LT::Caps::INV || (nnext > LT::Caps::IFL && nnext <= 10)) return COOP_BIG_CHECK;
            }
            if (!cont) break;
            cur ^= 1; ncur = nnext; remaining = ncur; depth++;
            if (depth > 3 * k) break;
            if (ncur > 10) break;
            if (ncur == 0) break;
        }
        if (remaining > 0) return 0;
    }
    return 1;
}

/* [MEM] MonumentTraversal::find_end_of_branching by the group (one-lane form: find_end_of_branching).  Returns the depth (> 0), COOP_FAIL or
 * COOP_TOOBIG; leaves the involved list in L.inv / n_inv. */
template <int G, class LT> MTG_DEV_NOINLINE int coop_find_end(Worker& W, LT& L, const Kmer& start, uint64_t prev_c, uint64_t& end_f, uint64_t& end_rp, uint32_t& n_inv)
{
    typedef Grp<G> GP;
    const int k = W.k;
    const UStore& us = W.ix.us;
    const uint32_t gl = GP::gl();
    for (uint32_t i = gl; i < (uint32_t)LT::Caps::SEEN; i += GP::N) L.a.seen[i] = 0;
    GP::sync();
    if (gl == 0) {
        lset_insert(L.a.seen, LT::Caps::SEEN, canon(start));
        lset_insert(L.a.seen, LT::Caps::SEEN, prev_c);
        FlNode s;
        s.f = start.f; s.rp = 0; s.ra = 0; s.aux = 0;
        L.a.fl[0][0] = s;
    }
    uint32_t n_seen = 2;
    GP::sync();
    int cur = 0, ncur = 1, depth = 0;
    const bool may_skip = us.nwords != 0 && prev_c != 0;
    uint32_t prev_unitig = 0xFFFFFFFFu;
    bool prev_known = false;
    MTG_GUARD_DECL(g2);
    for (;;) {
        MTG_GUARD(g2, 100000u, 11, return COOP_TOOBIG);
        FlNode* cf = L.a.fl[cur];
        FlNode* nf = L.a.fl[cur ^ 1];
        /* ---- the skip (see the one-lane form for why it changes nothing) */
        if (may_skip && ncur >= 2 && depth > 0) {
            bool ok_l = true;
            uint32_t d_l = 0xFFFFFFFFu;
            for (int i = (int)gl; i < ncur; i += GP::N) {
                const uint32_t ra = cf[i].ra;
                if (!(cf[i].rp & RP_VALID) || ra < 2u) ok_l = false;
                else if (ra - 1u < d_l) d_l = ra - 1u;
            }
            bool ok = GP::all(ok_l);
            if (ok) {
                if (!prev_known) { /* the junction between the previous node and the start = the start's left junction */
                    prev_known = true;
                    prev_unitig = left_junction_unitig(W.ix, start, W.mk1, W.lines);
                }
                ok_l = true;
                for (int i = (int)gl; i < ncur; i += GP::N) {
                    const uint32_t u = rp_unitig(cf[i].rp);
                    if (u == prev_unitig) ok_l = false;
                    for (int j = 0; j < i && ok_l; j++) if (rp_unitig(cf[j].rp) == u) ok_l = false;
                }
                ok = GP::all(ok_l);
            }
            if (ok) {
                const uint32_t D = GP::min32(d_l);
                if ((uint32_t)depth + D > (uint32_t)W.cfg.mono_max_depth) return COOP_FAIL;
                for (int i = (int)gl; i < ncur; i += GP::N) {
                    const uint64_t rp = cf[i].rp;
                    cf[i].f = run_node(us, rp & RP_KPOS, (rp & RP_BWD) != 0, D, k).f;
                    cf[i].rp = rp_step(rp, D);
                    cf[i].ra -= D;
                    cf[i].aux = AUX_IN1;
                }
                W.lines += (uint32_t)ncur;
                depth += (int)D;
                GP::sync();
                continue;
            }
        }
        /* ---- one level.  What every node leads to: one lane per node, the index reads of a level in flight together */
        bool chk_l = false;
        for (int i = (int)gl; i < ncur; i += GP::N) {
            const FlNode nd = cf[i];
            const Kmer x = make_kmer(nd.f, k);
            FlExp e;
            e.krp = 0; e.kra = 0;
            e.chk = (depth > 0 && !(nd.aux & AUX_IN1)) ? 1u : 0u; /* a node of in-degree 1 passes the check at once */
            if ((nd.rp & RP_VALID) && nd.ra >= 1u) { /* inside a unitig: the way ahead is known */
                e.out = 1u << run_next_nt(us, nd.rp & RP_KPOS, (nd.rp & RP_BWD) != 0, k);
                e.kid = AUX_IN1;
                e.krp = rp_step(nd.rp, 1);
                e.kra = nd.ra - 1u;
            } else if (nd.aux & 15u) { e.out = 1u << ((nd.aux >> 4) & 3u); e.kid = aux_step(nd.aux); }
            else {
                Adj a = adj_right_t(W.ix.adj, x, W.mk1, W.lines);
                RunAt r;
                if (run_at(us, a, k, r, W.lines)) { e.krp = rp_step(rp_pack(r), 1); e.kra = r.ahead - 1u; }
                else adj_resolve_la(W.ix, a, W.lines);
                e.out = a.out;
                e.kid = aux_of_children(a);
            }
            chk_l = chk_l || e.chk != 0;
            L.a.ex[i] = e;
        }
        const bool any_chk = GP::any(chk_l);
        GP::sync();
        /* ---- the successors: one lane per (node, nucleotide).  A node that must pass the in-branching check first sees the visited set as
         * the nodes before it left it, so a level with such a node is taken node by node; otherwise all of it at once. */
        int nnext = 0;
        for (int seg = 0; seg < (any_chk ? ncur : 1); seg++) {
            uint32_t lo = 0, hi = 4u * (uint32_t)ncur;
            if (any_chk) {
                lo = 4u * (uint32_t)seg; hi = lo + 4u;
                if (L.a.ex[seg].chk) {
                    const int c = coop_fl_check<G>(W, L, cf[seg].f, n_inv);
                    if (c != 1) return c;
                }
            }
            for (uint32_t base = lo; base < hi; base += GP::N) {
                const uint32_t j = base + gl, node = j >> 2, nt = j & 3u;
                bool cand = false;
                Kmer y;
                y.f = y.r = 0;
                uint64_t cy = 0;
                FlExp e;
                e.krp = 0; e.kra = 0; e.kid = 0; e.out = 0; e.chk = 0;
                if (j < hi) {
                    e = L.a.ex[node];
                    if (e.out & (1u << nt)) {
                        y = kmer_next(make_kmer(cf[node].f, k), nt, k, W.mk);
                        cy = canon(y);
                        cand = !lset_has(L.a.seen, LT::Caps::SEEN, cy);
                    }
                }
                /* a node reached twice in this chunk: the first one in (node, nucleotide) order stands, as in the one-lane form (every
                 * candidate's k-mer is passed round the group) */
                uint64_t cb = GP::ballot(cand);
                {
                    bool dup = false;
                    for (uint64_t m = cb; m; m &= m - 1ull) {
                        const int b2 = ctz64(m);
                        const uint64_t other = GP::from_lane64(cy, b2);
                        if ((uint32_t)b2 < gl && other == cy) dup = true;
                    }
                    if (dup) cand = false;
                }
                cb = GP::ballot(cand);
                if (GP::any(cand && W.is_marked(cy))) return COOP_FAIL; /* the bubble touches an assembled region */
                const bool simple = (e.kid & 15u) || ((e.krp & RP_VALID) && e.kra >= 1u);
                const uint64_t ib = GP::ballot(cand && !simple);
                if (cand) {
                    const int pos = nnext + popc64(cb & ((1ull << gl) - 1ull));
                    if (pos < LT::Caps::FL) { FlNode s; s.f = y.f; s.rp = e.krp; s.ra = e.kra; s.aux = e.kid; nf[pos] = s; }
                    lset_insert(L.a.seen, LT::Caps::SEEN, cy);
                    if (!simple) {
                        const uint32_t ipos = n_inv + (uint32_t)popc64(ib & ((1ull << gl) - 1ull));
                        if (ipos < (uint32_t)LT::Caps::INV) L.inv[ipos] = cy;
                    }
                }
                nnext += popc64(cb);
                n_seen += (uint32_t)popc64(cb);
                n_inv += (uint32_t)popc64(ib);
                GP::sync();
                if (nnext > W.cfg.mono_max_breadth) return COOP_FAIL; /* the one-lane form finishes the level first; nothing it does there changes the answer */
                if (nnext > LT::Caps::FL) return COOP_TOOBIG;
                if (n_seen > (uint32_t)LT::Caps::SEEN * 3u / 4u || n_inv > (uint32_t)LT::Caps::INV) return COOP_TOOBIG;
            }
        }
        cur ^= 1; ncur = nnext; depth++;
        if (depth > W.cfg.mono_max_depth) return COOP_FAIL;
        if (ncur > W.cfg.mono_max_breadth) return COOP_FAIL;
        if (ncur == 0) return COOP_FAIL;
        if (ncur == 1) break; /* end_rule_nonbranching is served by the one-lane form */
    }
    end_f = L.a.fl[cur][0].f;
    end_rp = L.a.fl[cur][0].rp;
    return depth;
}

/* [MEM] MonumentTraversal::all_consensuses_between with its frames, path set and consensuses in LDS (one-lane form: all_consensuses_between,
 * which explains the unitig-wise frames).  The enumeration is a depth-first search whose steps depend on each other: every lane of the
 * group runs it with the same values (the loads are one request per group), the bulk copies of nucleotides are dealt to the lanes.
 * 1 = ok (ncons consensuses in L.b), COOP_FAIL, COOP_TOOBIG. */
template <int G, class LT> MTG_DEV_NOINLINE int coop_consensuses(Worker& W, LT& L, const Kmer& start, uint64_t end_c, uint64_t end_rp, int traversal_depth, int& ncons)
{
    typedef Grp<G> GP;
    const int k = W.k;
    const UStore& us = W.ix.us;
    const uint32_t gl = GP::gl();
    if (traversal_depth + 2 > LT::Caps::NT || traversal_depth + 2 > LT::Caps::CLEN) return COOP_BIG_DEPTH;
    DfsFrame* fr = L.b.fr;
    uint8_t* dfs_nt = L.b.nt;
    uint64_t* pset = L.b.pset;
    const uint64_t TOMB = ~0ULL - 1;
    for (uint32_t i = gl; i < (uint32_t)LT::Caps::ISEEN; i += GP::N) pset[i] = 0;
    GP::sync();
    ncons = 0;
    uint32_t n_path = 0; /* slots of the path set in use (tombstones included) */
    /* 0: added, 1: already on the path, 2: full.  Every lane performs the same operations on the same words. */
    auto path_add = [&](uint64_t c) -> int {
        uint32_t h = set_hash(c, LT::Caps::ISEEN);
        int64_t tomb = -1;
        MTG_GUARD_DECL(g4);
        for (;;) {
            MTG_GUARD(g4, 1000u, 13, return 2);
            const uint64_t v = pset[h];
            if (v == c + 1) return 1;
            if (v == TOMB && tomb < 0) tomb = (int64_t)h;
            if (v == 0) {
                if (tomb >= 0) { pset[(uint32_t)tomb] = c + 1; return 0; }
                if (n_path + 2 >= (uint32_t)LT::Caps::ISEEN * 3u / 4u) return 2;
                n_path++;
                pset[h] = c + 1;
                return 0;
            }
            h = (h + 1) & (LT::Caps::ISEEN - 1);
        }
    };
    auto path_del = [&](uint64_t c) {
        uint32_t h = set_hash(c, LT::Caps::ISEEN);
        MTG_GUARD_DECL(g5);
        for (;;) {
            MTG_GUARD(g5, 1000u, 14, return);
            const uint64_t v = pset[h];
            if (v == 0) return;
            if (v == c + 1) { pset[h] = TOMB; return; }
            h = (h + 1) & (LT::Caps::ISEEN - 1);
        }
    };
    int f = 0;
    {
        DfsFrame z;
        z.f = start.f; z.c = canon(start); z.rp = 0; z.ra = 0; z.dep = 0; z.xsn = 0; z.kid = 0; z.mask = 0; z.pad_ = 0;
        fr[0] = z;
    }
    GP::sync();
    GP::uniform([&] { return path_add(fr[0].c); });
    GP::sync();
    const uint32_t end_u = (end_rp & RP_VALID) ? rp_unitig(end_rp) : 0xFFFFFFFFu;
    bool entering = true;
    MTG_GUARD_DECL(g3);
    for (;;) {
        MTG_GUARD(g3, 1000000u, 12, return COOP_TOOBIG);
        if (entering) {
            entering = false;
            const int d = (int)fr[f].dep;
            if (traversal_depth - d < -1) return COOP_FAIL;
            if (fr[f].c == end_c) {
                if (ncons >= LT::Caps::CONS) return COOP_BIG_NCONS;
                if (d > LT::Caps::CLEN) return COOP_BIG_DEPTH;
                for (int i = (int)gl; i < d; i += GP::N) L.b.cons[ncons][i] = dfs_nt[i];
                if (gl == 0) { L.b.len[ncons] = (uint16_t)d; L.b.sum[ncons] = f ? (int32_t)fr[f - 1].xsn : 0; }
                ncons++;
                GP::sync();
                fr[f].mask = 0; /* return */
            } else {
                const Kmer x = make_kmer(fr[f].f, k);
                const uint32_t xs = f ? fr[f - 1].xsn : 0u;
                const uint32_t aux = f ? fr[f - 1].kid : 0u;
                uint64_t rp = fr[f].rp;
                uint32_t ra = fr[f].ra, mask, kid, abx;
                if ((rp & RP_VALID) && ra >= 1u) {
                    mask = 1u << run_next_nt(us, rp & RP_KPOS, (rp & RP_BWD) != 0, k);
                    kid = AUX_IN1;
                    abx = us.ab[rp & RP_KPOS];
                    W.lines++;
                } else if (aux & 15u) {
                    mask = 1u << ((aux >> 4) & 3u);
                    kid = aux_step(aux);
                    abx = abundance(W.ix, x, W.lines);
                    rp = 0; ra = 0;
                } else {
                    Adj a = adj_right_t(W.ix.adj, x, W.mk1, W.lines);
                    RunAt r;
                    if (us.nwords && run_at(us, a, k, r, W.lines)) { rp = rp_pack(r); ra = r.ahead; abx = us.ab[r.kpos]; }
                    else { adj_resolve_la(W.ix, a, W.lines); rp = 0; ra = 0; abx = abundance(W.ix, x, W.lines); }
                    mask = a.out;
                    kid = aux_of_children(a);
                }
                GP::sync();
                fr[f].mask = mask; fr[f].kid = kid; fr[f].rp = rp; fr[f].ra = ra; fr[f].xsn = xs + abx;
            }
            GP::sync();
        }
        const uint32_t mask = fr[f].mask;
        if (mask == 0) {
            if (f == 0) return 1;
            const uint64_t cdel = fr[f].c;
            GP::sync();
            GP::uniform([&] { path_del(cdel); return 0; });
            GP::sync();
            f--;
            if (ncons > W.cfg.mono_max_breadth) return COOP_FAIL;
            continue;
        }
        const uint32_t nt = (uint32_t)ctz4(mask);
        const int d = (int)fr[f].dep;
        const uint64_t rp = fr[f].rp;
        const uint32_t ra = fr[f].ra;
        const uint64_t ff = fr[f].f;
        const uint32_t xsn0 = fr[f].xsn;
        GP::sync();
        fr[f].mask = mask & (mask - 1);
        Kmer y;
        uint32_t t = 1u, kra = 0;
        uint64_t krp = 0;
        if ((rp & RP_VALID) && ra >= 2u && rp_unitig(rp) != end_u) {
            /* the whole stretch: the child is the unitig's end node */
            t = ra;
            if (d + (int)t > traversal_depth + 1) return COOP_FAIL;
            const bool bwd = (rp & RP_BWD) != 0;
            const uint64_t kpos = rp & RP_KPOS;
            for (uint32_t i = 16u * gl; i < t; i += 16u * GP::N) {
                const uint32_t n = t - i < 16u ? t - i : 16u;
                uint32_t seq = us_peek(us.words, bwd ? kpos - 1u - i : kpos + (uint32_t)k + i, n, bwd);
                for (uint32_t j = 0; j < n; j++) { dfs_nt[(size_t)d + i + j] = (uint8_t)(seq & 3u); seq >>= 2; }
            }
            uint32_t sum = 0;
            for (uint32_t i = 1; i < t; i += 64u) {
                const uint32_t n = t - i < 64u ? t - i : 64u;
                sum += us_ab_sum(us.ab, bwd ? kpos - (i + n - 1u) : kpos + i, n);
            }
            W.lines += (t >> 5) + 2u;
            fr[f].xsn = xsn0 + sum;
            fr[f].kid = AUX_IN1;
            y = run_node(us, kpos, bwd, t, k);
            krp = rp_step(rp, t);
            kra = 0;
        } else {
            y = kmer_next(make_kmer(ff, k), nt, k, W.mk);
            dfs_nt[d] = (uint8_t)nt;
            if ((rp & RP_VALID) && ra >= 1u) { krp = rp_step(rp, 1u); kra = ra - 1u; }
        }
        const uint64_t cy = canon(y);
        if (f + 1 >= LT::Caps::FR || d + (int)t >= LT::Caps::NT) return COOP_BIG_FRAMES;
        GP::sync();
        const int pa = GP::uniform([&] { return path_add(cy); });
        if (pa == 1) return COOP_FAIL; /* loop inside the bubble */
        if (pa == 2) return COOP_BIG_PATH;
        f++;
        {
            DfsFrame z;
            z.f = y.f; z.c = cy; z.rp = krp; z.ra = kra; z.dep = (uint32_t)d + t; z.xsn = 0; z.kid = 0; z.mask = 0; z.pad_ = 0;
            fr[f] = z;
        }
        GP::sync();
        entering = true;
    }
}

/* [MEM] MonumentTraversal::validate_consensuses + most_abundant_consensus on the consensuses in L.b (one-lane form: validate_consensuses).
 * Index of the chosen consensus, -1 = rejected, -2 = the alignment itself is needed (left to the one-lane form). */
template <int G, class LT> MTG_DEV_NOINLINE int coop_validate(Worker& W, LT& L, int ncons)
{
    if (ncons <= 0) return -1;
    const int k = W.k;
    int mean = 0;
    for (int c = 0; c < ncons; c++) mean += L.b.len[c];
    mean /= ncons;
    long long ss = 0;
    for (int c = 0; c < ncons; c++) { const long long dl = (long long)L.b.len[c] - mean; ss += dl * dl; }
    if (mean > W.cfg.mono_max_depth) return -1;
    if (ncons == 1 && mean > k + 1) return -1;
    const long long t = mean / 5;
    if (ss > t * t * ncons) return -1;
    for (int a = 0; a < ncons; a++)
        for (int b = a + 1; b < ncons; b++) {
            const int na = L.b.len[a], nb = L.b.len[b];
            const uint8_t* pa = L.b.cons[a];
            const uint8_t* pb = L.b.cons[b];
            const int mn = na < nb ? na : nb, mx = na < nb ? nb : na;
            int s0 = -5 * (mx - mn);
            for (int i = 0; i < mn; i++) s0 += (pa[i] == pb[i]) ? 10 : -5;
            if (mx != mn && identity_below_90((s0 + 5 * mx <= 0) ? 0 : (s0 + 5 * mx + 14) / 15, na, nb)) {
                const uint8_t* lg = na >= nb ? pa : pb;
                const uint8_t* sh = na >= nb ? pb : pa;
                const int dl = mx - mn;
                int sc = -5 * dl;
                for (int i = 0; i < mn; i++) sc += (lg[i + dl] == sh[i]) ? 10 : -5;
                int best = sc;
                for (int p = 0; p < mn; p++) {
                    sc += ((lg[p] == sh[p]) ? 10 : -5) - ((lg[p + dl] == sh[p]) ? 10 : -5);
                    best = sc > best ? sc : best;
                }
                s0 = best > s0 ? best : s0;
            }
            const int num = s0 + 5 * mx;
            const int mlb = num <= 0 ? 0 : (num + 14) / 15;
            if (identity_below_90(mlb, na, nb)) return -2; /* the bound does not decide */
        }
    unsigned long best = 0;
    int chosen = -1;
    for (int c = 0; c < ncons; c++) {
        const int len = L.b.len[c];
        if (len == 0) continue;
        unsigned long sum = (unsigned long)L.b.sum[c];
        sum /= (unsigned long)len;
        if (sum > best) { best = sum; chosen = c; }
    }
    if (chosen < 0) return -1;
    if ((int)L.b.len[chosen] > W.cfg.mono_max_depth) return -1;
    return chosen;
}

/* [MEM] MonumentTraversal::explore_branching by the group: the plan.  Returns the length of the chosen consensus (its nucleotides in
 * L.b.cons[chosen], the nodes to mark in L.inv[0, L.n_marks)), COOP_FAIL (the one-lane form would return 0) or COOP_TOOBIG.  Nothing
 * outside L is written: apply_marks makes the plan effective. */
template <int G, class LT> MTG_DEV_NOINLINE int coop_explore(Worker& W, LT& L, const Kmer& cur, uint64_t prev_c, int& chosen)
{
    typedef Grp<G> GP;
    if (W.cfg.end_rule_nonbranching) return COOP_BIG_RULE;
    const uint32_t gl = GP::gl();
    uint32_t n_inv = 0;
    uint64_t end_f = 0, end_rp = 0;
    const int d = coop_find_end<G>(W, L, cur, prev_c, end_f, end_rp, n_inv);
    GP::sync();
    if (d <= 0) return d;
    const Kmer e = make_kmer(end_f, W.k);
    int ncons = 0;
    const int okc = coop_consensuses<G>(W, L, cur, canon(e), end_rp, d + 1, ncons);
    GP::sync();
    if (okc != 1) return okc;
    chosen = coop_validate<G>(W, L, ncons);
    if (chosen == -2) return COOP_BIG_ALIGN;
    if (chosen < 0) return COOP_FAIL;
    /* which of the involved nodes are branching: all lanes at once */
    for (uint32_t i = gl; i < n_inv; i += GP::N) {
        Kmer x;
        x.f = L.inv[i];
        x.r = revcomp(x.f, W.k);
        L.invbr[i] = W.is_branching(x) ? 1 : 0;
    }
    GP::sync();
    uint32_t nm = 0;
    for (uint32_t i = 0; i < n_inv; i++) if (L.invbr[i]) { if (gl == 0) L.inv[nm] = L.inv[i]; nm++; } /* compacted in place: nm <= i */
    if (gl == 0) L.n_marks = nm;
    GP::sync();
    return (int)L.b.len[chosen];
}
template <int G, class LT> MTG_DEV void coop_apply_marks(Worker& W, LT& L)
{
    const uint32_t nm = L.n_marks;
    for (uint32_t i = 0; i < nm; i++) W.mark_canon(L.inv[i]);
}

} // namespace mtg
#endif
