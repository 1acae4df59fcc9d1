/*
 * mtg_tuning.h -- every switch of the library in ONE table: its name, its default, what it does.
 *
 * Rounds 1-3 grew some forty getenv() calls next to the code they steer (capacities, A/B hooks of measured alternatives, hooks the tests use
 * to force rare paths).  They are entries of MTG_TUNABLES now: the C-ABI lists, reads and sets them (include/mtg_fill.h: mtg_tuning_*), the
 * environment still works -- entry X starts from the variable MTG_X, and MTG_TUNING="X=1,Y=0.5" sets several -- and the code asks
 * tune::i(T_X) / tune::f(T_X) / tune::on(T_X) where it used to parse a string.  Value of an entry: what mtg_tuning_set (or MTG_TUNING) put
 * there, else MTG_X of the environment as it is at the moment of the question, else the default; a change takes effect for the calls that
 * begin after it (an index keeps the capacities it was built with).
 *
 * Kinds: "cap" a capacity or size a deployment may want to change; "ab" an A/B hook that keeps a measured alternative runnable (DESIGN.md
 * says which measurement decided); "test" a hook the tests use to force a rare path; "diag" diagnostics on stderr.
 * An empty default means "not set": flags are off, numbers take the value the code computes.
 */
#ifndef MTG_TUNING_H
#define MTG_TUNING_H
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>

/* X(name, default, kind, what) */
#define MTG_TUNABLES(X) \
    X(POOL_THREADS, "", "cap", "size of the host worker pool (default: the CPUs the process may use -- hardware threads, affinity mask, cgroup quota -- at most 64); read when the pool starts") \
    X(COPY_SLOTS, "3", "cap", "batches of a device that copy their results to the host at the same time (0: no limit); more copies at once share the link worse") \
    X(UPLOAD_OWN_STREAM, "", "ab", "every batch uploads its input on its own stream instead of the device's one upload stream (three uploads at a time take the link from the downloads: profiles/r04_text_entry.txt)") \
    X(SPARSE_ADJ_LOAD, "0.49", "cap", "load factor of the sparse junction table, as a share of the dense one's; 0.7 overflows the displacement range at human scale") \
    X(JT_LOAD, "0.7", "cap", "load factor of the lean build's junction table") \
    X(BLOOM_BITS, "12", "cap", "Bloom filter bits per k-mer (0: no filter, no sequence scan)") \
    X(LOAD_THREADS, "", "cap", "threads that stream a container's abundance bytes to the device (default: up to 8)") \
    X(LOAD_PIECE, "", "test", "k-mers per piece when an index container is loaded (default 2^26): small pieces exercise the piece loop") \
    X(COUNT_PASSES, "", "test", "force this many passes of the k-mer counting (default: as many as HBM needs)") \
    X(ROUNDS, "", "ab", "bubble rounds between launches of the walk kernel (default: 6 when most gaps of the previous launch parked, else 0)") \
    X(LIGHT_WALK, "", "ab", "1 / 0: the first walk of a launch by the light kernel (simple paths only, every branching node parks) / by the full one; default: light when fewer than 1 in 128 gaps of the previous launch met a branching node") \
    X(BUILD_PARTITIONED, "", "ab", "the junction table of packed sequences built partition by partition in LDS (1) or by scattered insertion (0); default: partitioned from 2^27 junction positions on") \
    X(NO_SPLIT_LONG, "", "test", "index construction from packed sequences: a sequence of more than 131 072 nucleotides stays one work item (no pieces of 65 536 that share k - 1 nucleotides)") \
    X(BUILD_POSITIONAL, "", "ab", "packed sequences: the chains that lie whole in one sequence are found by position (1, default) or every chain is walked on the junction table (0)") \
    X(FINISH_G, "", "ab", "lanes per parked gap in the finishing kernel: 1, 8, 16 or 64 (default: 64 while few gaps park, 16 otherwise)") \
    X(FINISH_WAVE_BELOW, "2048", "ab", "a whole wave per parked gap while the previous launch parked fewer gaps than this") \
    X(NO_POST_INDEX, "", "test", "the terminal search counts every contig position against every target of its gap (no piece index of the batch's dictionaries)") \
    X(NO_LEAN, "", "test", "every contig is materialised (no lean gaps)") \
    X(NO_DEFER, "", "test", "the lanes of the traversal copy their long runs themselves (no copy commands, no k_copy work)") \
    X(MAX_CHUNK, "", "test", "gaps per traversal launch (default: what the scratch holds): several launches per batch") \
    X(HOST_GENERAL, "", "test", "the multi-contig gaps by the host's path (candidate sequences, de-duplication, coverage: as until round 4) instead of k_general") \
    X(HOST_PATHS, "", "test", "leave the path enumeration of multi-contig gaps to the host") \
    X(DEBUG_SKIP_FINISH, "", "diag", "parked gaps stay parked (and fail as overflowing gaps)") \
    X(KERNEL_TIMERS, "", "diag", "every batch records an event between its kernels and mtg_last_batch_stats carries each kernel's own time (off: three events per batch instead of nine, only device_span_ms)") \
    X(DEBUG_TIMERS, "", "diag", "per-phase wall times of every batch and every index construction on stderr") \
    X(NO_VEC, "", "test", "scalar instead of pext host code in the input pass") \
    X(NO_PREFETCH, "", "test", "no software prefetch in the input pass") \
    X(NB_GPUS, "", "cap", "the tool: devices to use (default: all visible; -nb-gpus overrides)") \
    X(CLI_BATCH, "", "cap", "the tool: sites per batch (default 100 000)") \
    X(CLI_IN_FLIGHT, "", "cap", "the tool: worker threads per device (default 3, at most 6)") \
    X(CLI_WRITERS, "1", "cap", "the tool: writer threads per output file (a file's lock serialises write(2): more than one per file was measured the same or slower)") \
    X(CLI_NO_MMAP, "", "test", "the tool reads its breakpoint file instead of mapping it") \
    X(HOST_FORMAT, "", "ab", "the tool: every site's text by the host's writers (round 3) instead of the device's formatter") \
    X(TOOL_TIMERS, "", "diag", "the tool: where the time of a run went (stderr)") \
    X(TOOL_QUIET, "", "diag", "the tool: no summary on stdout")

namespace mtgi {
namespace tune {
enum Id {
#define MTG_TUNE_ENUM(name, dflt, kind, what) T_##name,
    MTG_TUNABLES(MTG_TUNE_ENUM)
#undef MTG_TUNE_ENUM
    T_COUNT
};
struct Entry { const char* name; const char* env; const char* dflt; const char* kind; const char* what; };
inline const Entry g_entries[T_COUNT] = {
#define MTG_TUNE_ROW(name, dflt, kind, what) {#name, "MTG_" #name, dflt, kind, what},
    MTG_TUNABLES(MTG_TUNE_ROW)
#undef MTG_TUNE_ROW
};
/* The value of an entry: what mtg_tuning_set (or MTG_TUNING, read once) has put there, else the environment variable MTG_<NAME> AS IT IS NOW
 * (a caller that changes its environment between two calls is heard, as it was when the code asked getenv itself), else the default. */
struct Values {
    std::mutex m;
    /* fixed storage: a reader must never see a string move.  A value longer than this is refused by mtg_tuning_set. */
    char v[T_COUNT][64];
    bool forced[T_COUNT];
    Values()
    {
        memset(v, 0, sizeof v);
        memset(forced, 0, sizeof forced);
        if (const char* e = getenv("MTG_TUNING")) { /* NAME=value,NAME=value */
            std::string all(e);
            size_t b = 0;
            while (b < all.size()) {
                size_t c = all.find(',', b);
                if (c == std::string::npos) c = all.size();
                const std::string item = all.substr(b, c - b);
                const size_t eq = item.find('=');
                const std::string nm = item.substr(0, eq), val = eq == std::string::npos ? "1" : item.substr(eq + 1);
                const int t = find(nm.c_str());
                if (t >= 0) put(t, val.c_str());
                else fprintf(stderr, "MTG_TUNING: no entry named %s (mtg_tuning_describe lists them)\n", nm.c_str());
                b = c + 1;
            }
        }
    }
    static int find(const char* name)
    {
        if (!name) return -1;
        if (!strncmp(name, "MTG_", 4)) name += 4;
        for (int t = 0; t < T_COUNT; t++) if (!strcmp(g_entries[t].name, name)) return t;
        return -1;
    }
    bool put(int t, const char* val)
    {
        const size_t n = val ? strlen(val) : 0;
        if (n >= sizeof v[t]) return false;
        memset(v[t], 0, sizeof v[t]);
        if (n) memcpy(v[t], val, n);
        forced[t] = true;
        return true;
    }
    /* the value of entry t into out (at most 63 characters); returns whether the entry has one.  Taken under the table's lock: mtg_tuning_set may
     * run while batches read their switches (round 4 read the strings lock-free while put() rewrote them: a data race the advisor named).
     * flag: an environment variable set to the empty string counts as "1" for a flag (MTG_X= switched it on when the code asked getenv() != NULL)
     * and as "not set" for a number. */
    bool read(int t, char out[64], bool flag)
    {
        std::lock_guard<std::mutex> lk(m);
        const char* src = nullptr;
        if (forced[t]) src = v[t];
        else if (const char* e = getenv(g_entries[t].env)) src = *e ? e : (flag ? "1" : "");
        else src = g_entries[t].dflt;
        size_t n = strlen(src);
        if (n > 63) n = 63;
        memcpy(out, src, n);
        out[n] = 0;
        return n != 0;
    }
};
inline Values& values() { static Values V; return V; }
inline bool is_set(Id id) { char b[64]; return values().read(id, b, true); } /* has a value (mtg_tuning_set's, the environment's or its default) that is not empty */
/* flags: any value but "0" switches them on (NO_LEAN=0 and MTG_TUNING=NO_LEAN=0 leave the flag off; round 4 took every non-empty value for on) */
inline bool on(Id id) { char b[64]; return values().read(id, b, true) && strcmp(b, "0") != 0; }
inline long i(Id id, long unset = 0) { char b[64]; return values().read(id, b, false) ? atol(b) : unset; } /* the value as an integer (not set: the argument) */
inline double f(Id id, double unset = 0.0) { char b[64]; return values().read(id, b, false) ? atof(b) : unset; }
} // namespace tune
} // namespace mtgi
#endif
