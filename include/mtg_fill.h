/*
 * mtg_fill.h -- C ABI of libmtgfill.so: the MI355X-native drop-in for the hot path of `MindTheGap fill`.
 *
 * The reference has no plugin / FFI seam; its de-facto boundary for this path is
 *   (i)  the tool entry      Filler::run(argc, argv) -> Filler::execute   (src/main.cpp:105-120, src/Filler.cpp:136)
 *   (ii) the per-gap call    Filler::gapFillFromSource<span>               (src/Filler.hpp:188-189, src/Filler.cpp:854-1026)
 *   (iii) the gatb Graph it queries: Graph::create / Graph::load / successors / predecessors / contains /
 *        queryAbundance (call sites src/Filler.cpp:210,222,415-428,978)
 * Every entry point below names the reference interface it replaces (paths relative to /root/reference).
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: plain C types only; every function returns an int status (MTG_OK = 0) unless noted; handles are
 * created by the library and released by the matching *_free; result arenas are library-owned until freed; all
 * functions require a HIP device (gfx950) and return MTG_ERR_NO_DEVICE without one -- there is no CPU fallback.
 * k-mer encoding everywhere: 2 bits per nucleotide, A=0 C=1 T=2 G=3 ((ascii>>1)&3, as gatb), first nucleotide in
 * the most significant bits of a uint64_t; k <= 31.
 */
#ifndef MTG_FILL_H
#define MTG_FILL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
    MTG_OK = 0,
    MTG_ERR_NO_DEVICE = 1,   /* no HIP device / HIP runtime failure (message via mtg_last_error) */
    MTG_ERR_ARG = 2,         /* invalid argument (k out of range, null pointer, non-ACGT k-mer, ...) */
    MTG_ERR_IO = 3,          /* cannot read / write a file */
    MTG_ERR_NOMEM = 4,       /* device or host allocation failed */
    MTG_ERR_OVERFLOW = 5,    /* a gap exceeded the largest scratch tier (pathological traversal) */
    MTG_ERR_FORMAT = 6       /* unsupported index container (e.g. a GATB HDF5 graph) */
};

const char* mtg_last_error(void);           /* thread-local message of the last failing call */
int mtg_device_count(void);                 /* number of visible HIP devices (0 if none); not a status */
int mtg_set_device(int device);             /* device used by subsequent calls of this thread */

/* ------------------------------------------------------------------------------------------------------------
 * Index = gatb Graph (solid canonical k-mers + abundance), device resident.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct mtg_index mtg_index;

typedef struct mtg_index_info {
    int k;
    int abundance_min;          /* threshold used ("abundance_min (used)", src/Filler.cpp:419-424) */
    int abundance_auto;         /* auto-inferred cut-off or -1 (src/Filler.cpp:415) */
    uint64_t nb_solid_kmers;    /* "kmers_nb_solid"  (src/Filler.cpp:427) */
    uint64_t nb_branching;      /* "nb_branching"    (src/Filler.cpp:428) */
    uint64_t device_bytes;      /* HBM held by the index */
    uint64_t adj_buckets, abnd_buckets;
    uint32_t adj_bucket_bytes, abnd_bucket_bytes; /* one bucket = one read of the walk / of an abundance query */
    uint64_t bloom_blocks;      /* 64-byte blocks of the Bloom filter (0 = not built) */
    uint32_t bloom_minimizer;   /* minimizer length selecting the block */
    uint64_t nb_unitigs;        /* maximal simple paths (>= 2 k-mers) held in the unitig store */
    uint64_t unitig_bytes;      /* HBM held by the unitig store (2-bit sequences + one abundance byte per k-mer) */
    uint64_t nb_saturated;      /* solid k-mers whose abundance exceeds 255 and is stored as 255 (gatb reports discretised values above ~70:
                                   coverage statistics of very deep regions may differ from the reference's) */
    uint32_t sparse;            /* 1: the sparse form -- the ADJ table holds every second junction of a stored unitig (and all others), the ABND
                                   table only the k-mers of no stored unitig; the rest is read off the unitig store */
    uint32_t pad_;
    uint64_t nb_kmers_outside_unitigs;
} mtg_index_info;

/* Graph::create(props) from read files (src/Filler.cpp:172-213): paths_csv = comma separated FASTA/FASTQ(.gz);
 * abundance_min < 0 = "auto" (src/Filler.cpp:106); abundance_max <= 0 = unlimited. */
int mtg_index_create_from_reads(const char* paths_csv, int k, int abundance_min, int abundance_max, mtg_index** out);
/* Graph from an already counted solid set (host arrays; canonical k-mers, abundance >= 1). */
int mtg_index_create_from_kmers(const uint64_t* canon_kmers, const uint32_t* abundance, size_t n, int k, mtg_index** out);
/* Graph whose nodes are all k-mers of the given sequences (2-bit packed, 32 nt per uint64_t word, nt i at bits
 * 2*(i%32); seq s occupies words [word_off[s], ...) and has len[s] nts; arrays in DEVICE memory).  Abundance of a
 * k-mer (deterministic): abund_span > 0: abund_lo + hash(k-mer) % abund_span; abund_span == 0: a Poisson(24) variate drawn from the hash
 * (SURVEY.md 8d), at least abund_lo.  Used for the synthetic benchmark sets.  Sequences of any length (shorter than k: ignored; a donor of a few
 * chromosome-sized sequences is taken in pieces internally); total_kmers_upper_bound >= the sum of max(len - k + 1, 0). */
int mtg_index_create_from_packed_device(const uint64_t* d_words, const uint64_t* d_word_off, const uint32_t* d_len, size_t nseq,
                                        uint64_t total_kmers_upper_bound, int k, uint32_t abund_lo, uint32_t abund_span, mtg_index** out);
/* Graph::load / save (src/Filler.cpp:222): this library's own container; a GATB .h5 gives MTG_ERR_FORMAT. */
int mtg_index_load(const char* path, mtg_index** out);
int mtg_index_save(const mtg_index* idx, const char* path);
/* A copy of the index on another device of the node (tables and unitig store copied device to device, no rebuild): what the tool uses to
 * run on every GPU, the reference's Dispatcher threads (src/Filler.cpp:824,844) becoming one host thread per device.  The copy is an
 * independent index: free it with mtg_index_free. */
int mtg_index_replicate(const mtg_index* idx, int device, mtg_index** out);
int mtg_index_get_info(const mtg_index* idx, mtg_index_info* info);
/* How Graph::create / Graph::load (src/Filler.cpp:172-226) went on the device: one record per phase of the construction (kernel or group of
 * kernels, in order) with its device time and the bytes the implemented layout has to move for it (0: not accounted), the most device
 * memory the construction held at any moment, and its whole duration.  out may be NULL (n receives the number of phases). */
typedef struct mtg_build_phase {
    char name[32];
    double ms;
    uint64_t bytes;   /* bytes of the implemented layout the phase must read + write (random accesses counted as the lines they touch) */
    uint64_t units;   /* what the phase processed: k-mers, junctions, table slots ... (named in DESIGN.md section 3) */
} mtg_build_phase;
int mtg_index_build_profile(const mtg_index* idx, mtg_build_phase* out, size_t cap, size_t* n, uint64_t* peak_device_bytes, double* total_ms);
void mtg_index_free(mtg_index* idx);

/* Batched graph queries (host arrays in, host arrays out).  kmers[] in any orientation.
 * contains   : Graph::contains            -> 1/0
 * abundance  : Graph::queryAbundance      (src/Filler.cpp:978), 0 if absent, saturates at 255
 * neighbors  : Graph::successors / predecessors: bit nt of succ[i] set iff kmer[1:]+nt is solid (order A,C,T,G) */
int mtg_index_contains(const mtg_index* idx, const uint64_t* kmers, size_t n, uint8_t* out);
int mtg_index_abundance(const mtg_index* idx, const uint64_t* kmers, size_t n, uint32_t* out);
int mtg_index_neighbors(const mtg_index* idx, const uint64_t* kmers, size_t n, uint8_t* succ, uint8_t* pred);

/* Membership of every k-mer along sequences = Graph::contains per position (the access pattern of the reference's `find` scan,
 * src/FindBreakpoints.hpp:851-853,1012-1046).  A rolling 2-bit k-mer per position, probed against the minimizer-blocked Bloom filter
 * of the index (blocks staged in LDS, one coalesced 64-byte read per run of k-mers sharing a minimizer).
 * mode 0: Bloom answer only (no false negatives); mode 1: exact (Bloom pre-filter, positives confirmed in the abundance table).
 * out[s] receives len(seqs[s]) - k + 1 bytes (0/1); k-mers with a non-ACGT character give 0. */
typedef struct mtg_scan_stats {
    uint64_t n_kmers, bloom_positive, confirmed, blocks_staged;
    double kernel_ms;
} mtg_scan_stats;
int mtg_index_scan_sequences(const mtg_index* idx, const char* const* seqs, size_t nseq, int mode, uint8_t* const* out, mtg_scan_stats* st);
/* same on 2-bit packed sequences already in DEVICE memory (layout of mtg_index_create_from_packed_device);
 * bit (p % 64) of d_out_bits[word_off[s] + p / 64] = membership of the k-mer starting at nt p of sequence s */
int mtg_index_scan_packed_device(const mtg_index* idx, const uint64_t* d_words, const uint64_t* d_word_off, const uint32_t* d_len, size_t nseq, int mode,
                                 uint64_t* d_out_bits, mtg_scan_stats* st);

/* ------------------------------------------------------------------------------------------------------------
 * Gap filling = Filler::gapFillFromSource over a batch of gaps.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct mtg_params {
    int max_nodes;              /* -max-nodes  (100),   src/Filler.cpp:101 */
    int max_depth;              /* -max-length (10000), src/Filler.cpp:100 */
    int nb_mis_allowed;         /* 2, src/Filler.cpp:56 */
    int end_rule_nonbranching;  /* 0; see SURVEY.md A.5(i) */
    int nb_host_threads;        /* host workers of a batch (0 = the library's pool: the CPUs the process may use, at most 64) */
} mtg_params;
void mtg_default_params(mtg_params* p);

/* one call of gapFillFromSource (arguments as src/Filler.hpp:188-189) */
typedef struct mtg_gap {
    const char* source;               /* sourceSequence (first k nts seed the traversal) */
    const char* target;               /* targetSequence: the early-stop pattern R */
    int n_targets;                    /* targetDictionary, in the caller's iteration order */
    const char* const* target_seqs;   /*   key: k-mer string */
    const char* const* target_names;  /*   value.first */
    const uint8_t* target_is_rc;      /*   value.second */
    int is_anchor_repeated;
    int reverse;
} mtg_gap;

typedef struct mtg_filled {           /* filled_insertion_t, src/Utils.hpp:46-104 */
    const char* seq;                  /* NUL terminated, owned by the result arena */
    int nb_errors_in_anchor;
    int target_index;                 /* index into the gap's targetDictionary */
    float avg_coverage, median_coverage;
    int qual, solution_count, solution_rank;
} mtg_filled;

typedef struct mtg_gap_result {
    int nb_nodes;                     /* infostring fields, src/IGraphOutput.cpp:82-83, src/Filler.cpp:905,1012-1016 */
    int total_nt;
    int nb_terminal;
    int has_solution_counts;          /* whether the two fields below are appended to the infostring */
    int nb_total_filled, nb_reported;
    int n_filled;
    const mtg_filled* filled;         /* appended to filledSequences */
    const char* extension;            /* extensionSequence ("" when solutions exist) */
} mtg_gap_result;

typedef struct mtg_results mtg_results;
int mtg_fill_batch(const mtg_index* idx, const mtg_params* p, const mtg_gap* gaps, size_t n, mtg_results** out);
/* The same, and the filled sequences of the batch also laid out in seq_out: NUL-terminated, in gap order (solution order within a gap),
 * *seq_bytes in total -- the form in which results travel between ranks.  In the common case they are decoded there directly (the
 * results then point into seq_out, which must stay untouched until mtg_results_free); MTG_ERR_ARG when cap is too small (32 bytes per
 * packed word of contig plus one per gap always suffice: a little over the sum of the insert lengths). */
int mtg_fill_batch_serial(const mtg_index* idx, const mtg_params* p, const mtg_gap* gaps, size_t n, char* seq_out, uint64_t cap, uint64_t* seq_bytes,
                          mtg_results** out);
/* A batch marshalled ahead of time and kept in device memory: what a caller that fills the same gaps more than once, or reads its
 * breakpoints while an earlier batch is still on the device, would use instead of mtg_fill_batch (which does exactly this and then
 * forgets the batch).  `gaps` and every string it points to must stay alive and unchanged for as long as the batch is used.  A batch can
 * be filled by several threads at once.  Its device copies live on the device of `idx`: filling it with an index of another device (a
 * replica made by mtg_index_replicate), or with another nb_mis_allowed than it was prepared with, is refused with MTG_ERR_ARG. */
typedef struct mtg_batch mtg_batch;
int mtg_batch_prepare(const mtg_index* idx, const mtg_params* p, const mtg_gap* gaps, size_t n, mtg_batch** out);
void mtg_batch_free(mtg_batch* b);
int mtg_fill_prepared(const mtg_index* idx, const mtg_params* p, const mtg_batch* b, mtg_results** out);
int mtg_fill_prepared_serial(const mtg_index* idx, const mtg_params* p, const mtg_batch* b, char* seq_out, uint64_t cap, uint64_t* seq_bytes, mtg_results** out);
/* The same with the serialised sequences produced in DEVICE memory: d_seq_out is a buffer of `cap` bytes on the index's device (a consumer
 * on the device, or the send buffer of a gather over RCCL / xGMI).  host_copy == NULL: they stay there only -- the ASCII never crosses PCIe, the
 * records come to the host as always and the `seq` pointers of their filled sequences are device addresses into d_seq_out.  host_copy != NULL
 * (cap bytes, page-locked for speed): they are copied there as well, by the batch's own stream, and the records point into host_copy as
 * with mtg_fill_prepared_serial.  Both buffers are complete when the call returns. */
int mtg_fill_prepared_serial_device(const mtg_index* idx, const mtg_params* p, const mtg_batch* b, char* d_seq_out, uint64_t cap, char* host_copy, uint64_t* seq_bytes,
                                    mtg_results** out);
/* The same, and the whole batch -- records AND sequences -- also in relocatable form (mtg_wire_* below), tagged, in DEVICE memory: d_wire is a
 * buffer of `cap` bytes on the index's device, typically the send buffer of a gather over RCCL / xGMI.  The result kernel writes the
 * payload there next to the records (no host pass); the results come to the host as with mtg_fill_prepared.  The payload is complete when
 * the call returns. */
int mtg_fill_prepared_wire_device(const mtg_index* idx, const mtg_params* p, const mtg_batch* b, uint64_t tag, void* d_wire, uint64_t cap, uint64_t* wire_bytes, mtg_results** out);
/* A batch of gapFillFromSource calls whose strings lie in ONE block of text, the way a reader of a breakpoint file has them: strings are
 * (offset, length) pairs into the block, no pointer per string, no NUL terminators.  The library copies the block and the offset arrays into
 * page-locked memory, sends them up and ENCODES ON THE DEVICE (source k-mers, early-stop patterns, dictionary keys: k_marshal_text,
 * k_marshal_targets) -- the host does no per-string work, so a caller with one or two threads feeds the device as fast as a prepared batch
 * does.  Results exactly as mtg_fill_batch for the same strings.  Every offset + length must lie inside the block (MTG_ERR_ARG otherwise);
 * a source shorter than k is MTG_ERR_ARG as in mtg_fill_batch.  The block and the arrays must stay unchanged until the call returns (the
 * rare multi-contig gap reads its strings again). */
typedef struct mtg_text_gaps {
    const char* text;                 /* the block */
    uint64_t text_bytes;
    uint64_t n;                       /* gaps */
    const uint64_t* source_off;       /* sourceSequence of gap i = text[source_off[i], + source_len[i]) */
    const uint32_t* source_len;
    const uint64_t* pattern_off;      /* targetSequence: the early-stop pattern R */
    const uint32_t* pattern_len;
    const uint32_t* dict_first;       /* n + 1 entries: targetDictionary of gap i = entries [dict_first[i], dict_first[i + 1]) of the arrays below, in iteration order */
    const uint64_t* dict_seq_off;     /*   key: k-mer string */
    const uint32_t* dict_seq_len;
    const uint64_t* dict_name_off;    /*   value.first */
    const uint32_t* dict_name_len;
    const uint8_t* dict_is_rc;        /*   value.second (NULL: all 0) */
    const uint8_t* gap_flags;         /* per gap, bit 0: is_anchor_repeated, bit 1: reverse (NULL: all 0) */
} mtg_text_gaps;
int mtg_fill_text(const mtg_index* idx, const mtg_params* p, const mtg_text_gaps* g, mtg_results** out);
/* A caller that keeps its text where it is for many batches (a mapped or fully read breakpoint file) can page-lock it once: the block of a
 * mtg_text_gaps that lies inside a registered range goes to the device straight from the caller's memory, without the copy into the
 * library's page-locked block (13 MB per 100 000 sites).  The offset arrays are still copied (they are small).  The range must stay
 * registered and unchanged while calls that use it run.  MTG_ERR_ARG: the range cannot be page-locked (or overlaps a registered one). */
int mtg_host_register(void* p, size_t bytes);
int mtg_host_unregister(void* p);
/* mtg_fill_text for the gaps of BREAKPOINT sites (one-entry dictionaries), and the text the tool's writers add to its output files for them
 * -- writeFilledBreakpoint (src/Filler.cpp:1029-1093), the info line, writeVcf (:1095-1214) -- FORMATTED ON THE DEVICE from the records and
 * sequences the result kernel has just written there: what crosses PCIe is the files' next bytes, and the host only writes them.
 * name_off / name_len: breakpointName of site i = text[name_off[i], + name_len[i]) (the left record's header up to its first space).
 * The device writes the SIMPLE sites -- one solution, found on the common path of this (forward) attempt; every other site (no solution: the
 * caller's reverse attempt decides; several solutions; a record the host wrote) is listed in complex_sites with the offsets where the
 * caller's own text for it belongs, and its record is in *out as always (the ASCII arena of the simple sites is NOT brought to the host:
 * their records' seq pointers must not be followed).  *text: NULL for a new object, or one to use again (its arenas are page-locked). */
typedef struct mtg_formatted mtg_formatted;
typedef struct mtg_formatted_view {
    const char* text[3];            /* FASTA, info, VCF bytes of the simple sites, in site order */
    uint64_t bytes[3];
    uint64_t n_sites, n_simple;
    const uint32_t* complex_sites;  /* ascending */
    const uint64_t* complex_off[3]; /* complex_off[s][j]: bytes of text[s] that precede the text of site complex_sites[j] */
    uint64_t n_complex;
    double kernel_ms;               /* HIP-event time of the three formatting kernels */
} mtg_formatted_view;
int mtg_fill_text_formatted(const mtg_index* idx, const mtg_params* p, const mtg_text_gaps* g, const uint64_t* name_off, const uint32_t* name_len, mtg_results** out, mtg_formatted** text);
int mtg_formatted_get(const mtg_formatted* t, mtg_formatted_view* v);
void mtg_formatted_free(mtg_formatted* t);
/* and the filled sequences laid out in seq_out as with mtg_fill_batch_serial */
int mtg_fill_text_serial(const mtg_index* idx, const mtg_params* p, const mtg_text_gaps* g, char* seq_out, uint64_t cap, uint64_t* seq_bytes, mtg_results** out);
const mtg_gap_result* mtg_results_get(const mtg_results* r, size_t i);
/* Every pointer obtained from r dies here.  The library keeps the storage of up to six freed result sets (a few hundred bytes per
 * gap plus the sequences) and hands it to the next batches, which then pay no allocation, page fault or memset. */
void mtg_results_free(mtg_results* r);
/* bulk view of a result set (for gathers / checksums): n_filled[i] = number of sequences of gap i (may be NULL);
 * *seq_bytes = size of the concatenation "seq\n" of all filled sequences in gap order, which mtg_results_copy_seqs writes to dst */
int mtg_results_summary(const mtg_results* r, uint32_t* n_filled, uint64_t* seq_bytes, uint64_t* n_gaps_filled);
int mtg_results_copy_seqs(const mtg_results* r, char* dst, uint64_t cap);

/* ------------------------------------------------------------------------------------------------------------
 * A result set in relocatable form (offsets, no pointers): what travels between the ranks of a multi-GPU job -- one process per GPU, every
 * rank fills its shard of the sites, the rank that owns the output files gathers the shards' records AND sequences and writes them in
 * input order (the reference's writers under flockfile, src/Filler.cpp:682-683,1029-1214).  Layout:
 *   mtg_wire_header | n_gaps x mtg_wire_gap | n_filled x mtg_wire_filled | seq_bytes (NUL-terminated fills) | ext_bytes (NUL-terminated extensions)
 * every section 8-byte aligned.  tag is the caller's (the global index of the batch: gathers of several batches in flight arrive in any order).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct mtg_wire_header {
    uint64_t magic;                   /* "MTGWIRE1" */
    uint64_t tag;
    uint64_t n_gaps, n_filled, seq_bytes, ext_bytes;
    uint64_t total_bytes;             /* of the whole payload, header included */
    uint64_t checksum;                /* of everything behind the header (a sum of 64-bit words scrambled by their position) */
} mtg_wire_header;
typedef struct mtg_wire_gap {         /* mtg_gap_result */
    int32_t nb_nodes, total_nt, nb_terminal, has_solution_counts, nb_total_filled, nb_reported, n_filled;
    uint32_t first_filled;            /* index of its first mtg_wire_filled */
    uint64_t ext_off;                 /* offset of its extension sequence in the extension section (the empty string at offset 0) */
} mtg_wire_gap;
typedef struct mtg_wire_filled {      /* mtg_filled */
    uint64_t seq_off;                 /* offset in the sequence section */
    uint32_t seq_len;
    int32_t nb_errors_in_anchor, target_index, qual, solution_count, solution_rank;
    float avg_coverage, median_coverage;
} mtg_wire_filled;
int mtg_results_wire_size(const mtg_results* r, uint64_t* bytes);
int mtg_results_to_wire(const mtg_results* r, uint64_t tag, void* dst, uint64_t cap, uint64_t* bytes);
/* The reverse: a result set whose records point into `wire` (which must stay alive and unchanged until mtg_results_free); the payload is
 * validated (magic, sizes, offsets, checksum: MTG_ERR_FORMAT).  Needs no device. */
int mtg_results_from_wire(const void* wire, uint64_t bytes, mtg_results** out, uint64_t* tag);

/* The tool's writers for a run of breakpoint sites (writeFilledBreakpoint, writeVcf, writeExtensions; src/Filler.cpp:1029-1214,1275-1291)
 * on result sets that may have come over the wire: the text the sites add to <out>.insertions.fasta, .info.txt, .insertions.vcf and
 * .extensions.fasta, in site order.  rev / rev_index: the reverse attempts (src/Filler.cpp:669-680): rev_index[i] = index in `rev` of site
 * i's reverse attempt or -1 (rev may be NULL when there is none).  Needs no device. */
typedef struct mtg_site {
    const char* name;                 /* header of the left k-mer's record up to the first space (src/Filler.cpp:631-636) */
    const char* name_r;               /* the same of the right k-mer's record */
    const char* source;               /* the left k-mer's sequence */
    const char* target;               /* the right k-mer's sequence */
} mtg_site;
typedef struct mtg_text {
    char *fasta, *info, *vcf, *ext;   /* not NUL-terminated */
    uint64_t fasta_bytes, info_bytes, vcf_bytes, ext_bytes;
} mtg_text;
int mtg_format_bkpt(const mtg_site* sites, size_t n, const mtg_results* fwd, const mtg_results* rev, const int64_t* rev_index, int filter, int extend, mtg_text* out);
/* writeVcfHeader (src/Filler.cpp:349-383) into out->vcf: sample = the reads / graph the index came from, prefix = the output prefix */
int mtg_format_vcf_header(const char* sample, const char* prefix, mtg_text* out);
void mtg_text_free(mtg_text* t);

/* Stage A only (gatb IterativeExtensions::construct_linear_seqs, call site src/Filler.cpp:884): contigs of each gap
 * as ASCII, for parity tests and the info file.  sources[i] / targets[i] are NUL terminated strings. */
typedef struct mtg_contigs mtg_contigs;
int mtg_stage_a_batch(const mtg_index* idx, const mtg_params* p, const char* const* sources, const char* const* targets, size_t n,
                      mtg_contigs** out);
size_t mtg_contigs_count(const mtg_contigs* c, size_t gap);
const char* mtg_contigs_get(const mtg_contigs* c, size_t gap, size_t i);
void mtg_contigs_free(mtg_contigs* c);

/* timings / counters of the last fill or stage-A batch on this thread */
typedef struct mtg_batch_stats {
    double kernel_ms;            /* HIP-event time of the traversal kernel (k_stage_a) launches (sum over tiers / attempts) */
    double post_kernel_ms;       /* HIP-event time of the terminal-search / coverage kernel (k_post) and of the two layout scans behind it */
    double total_ms;             /* wall time of the whole call */
    double h2d_ms, d2h_ms, host_ms;
    double marshal_ms, result_ms; /* mtg_fill_batch only: argument marshalling and result-arena construction */
    uint64_t index_lines;        /* 64-byte index lines the kernels actually read */
    uint64_t n_launches;
    uint64_t n_retried_gaps;     /* gaps re-run in a larger scratch tier */
    uint64_t contig_nt;          /* nucleotides of all contigs built */
    uint64_t store_runs;         /* runs of nucleotides the traversal took from the unitig store (one header read each) */
    uint64_t run_nt;             /* nucleotides of the contigs that came out of the unitig store */
    uint64_t post_lines;         /* index buckets read by the coverage pass of k_post */
    uint64_t contig_words;       /* 8-byte words of contig arena the traversal wrote */
    uint64_t coverage_kmers;     /* k-mers whose abundance the coverage pass of k_post needed */
    uint64_t dense_words;        /* 8-byte words of contigs copied back for the multi-contig gaps */
    double emit_kernel_ms;       /* HIP-event time of the result kernel (k_emit: ASCII sequences + records); post_kernel_ms = k_post + the layout scans */
    uint64_t seq_bytes;          /* bytes of ASCII the result kernel wrote into the sequence arena */
    double copy_kernel_ms;       /* HIP-event time of the copy kernel (k_copy: long runs of the contigs out of the unitig store, one wave per gap) */
    uint64_t copy_words;         /* 8-byte words of contig arena it wrote */
    uint64_t copy_cmds;          /* runs it copied (one 24-byte command each) */
    uint64_t coverage_direct_kmers; /* of coverage_kmers: abundance bytes read at places known from the copy commands (no look-up, nothing to verify) */
    double finish_kernel_ms;     /* HIP-event time of the finishing kernel (k_finish: parked gaps resumed by groups of lanes, bubbles resolved from LDS) */
    uint64_t n_parked_gaps;      /* gaps the walk kernel parked at a branching node that is not the strict SNP pattern */
    uint64_t n_rounds;           /* bubble rounds queued between launches of the walk kernel (0: the parked gaps went straight to k_finish) */
    uint64_t n_lean_gaps;        /* gaps whose contig was never materialised: target located in the unitig store, coverage and ASCII read off the store */
    double lean_kernel_ms;       /* of copy_kernel_ms: the decision kernel (k_lean, one gap per lane); the rest is k_copy over the gaps that need it */
    uint64_t copy_words_executed; /* of copy_words: those k_copy really wrote (a lean gap's commands are never executed) */
    uint64_t copy_cmds_executed;
    uint64_t post_scanned_words; /* of contig_words: those k_post's terminal search really read (a lean gap's contig is never scanned) */
    double device_span_ms;       /* first kernel of a launch to its last, summed over the launches */
    uint64_t n_general_device;   /* multi-contig gaps finished on the device (k_general: candidate sequences, de-duplication, coverage, ASCII) */
    uint64_t n_general_host;     /* multi-contig gaps the host's path took (a candidate that does not fit the work areas, an unknown k-mer; HOST_GENERAL) */
    uint64_t n_light_walks;      /* launches whose first walk ran in the light kernel (k_walk: simple paths only; chosen when the previous launch hardly met a branching node) */
    uint64_t n_branching_gaps;   /* gaps whose first walk stood on a branching node at least once */
} mtg_batch_stats;
int mtg_last_batch_stats(mtg_batch_stats* s);

/* ------------------------------------------------------------------------------------------------------------
 * needleman_wunsch (src/Utils.cpp:87-189: match +10, mismatch -5, gap -5, traceback preference diagonal / up / left) for n pairs of
 * NUL-terminated sequences, on the device: matches[i] = matching positions along the traceback of (a[i] = rows, b[i] = columns); the
 * reference's identity is matches / max(strlen(a), strlen(b)).  Used by the fill path to de-duplicate multi-path solutions
 * (remove_almost_identical_solutions, src/Utils.cpp:208-238).
 * ---------------------------------------------------------------------------------------------------------- */
int mtg_nw_matches(const char* const* a, const char* const* b, size_t n, uint32_t* matches);

/* ------------------------------------------------------------------------------------------------------------
 * Whole tool: `MindTheGap fill ...` = Filler::run(argc, argv) (src/main.cpp:105-120).  argv[0] is the first option.
 * Returns the process exit code of the reference (0 / 1).
 * ---------------------------------------------------------------------------------------------------------- */
int mtg_fill_main(int argc, const char* const* argv);
/* The same behind Graph::load / Graph::create (src/Filler.cpp:172-226): the graph is an index that is already resident (options -in / -graph
 * are not expected; the index stays the caller's).  What a long-lived caller uses to run the tool's drivers and writers repeatedly. */
int mtg_fill_main_on_index(mtg_index* idx, int argc, const char* const* argv);

/* ------------------------------------------------------------------------------------------------------------
 * Tuning: every switch of the library -- capacities, A/B hooks of measured alternatives, hooks the tests use to force rare paths,
 * diagnostics -- is an entry of ONE table (mindthegap_amd/csrc/mtg_tuning.h: name, default, kind, what it does; INTEGRATION.md prints it).
 * An entry NAME starts from the environment variable MTG_NAME (or from MTG_TUNING="NAME=value,NAME=value"), and these calls list, read and
 * set the entries at run time.  A value set here holds for the calls that begin after it; an index keeps the capacities it was built with.
 * There is no equivalent in the reference (its options are all on the command line, src/Filler.cpp:95-135, and are mtg_params here): the
 * table only steers HOW this library computes, never WHAT -- every setting gives the same results, which is what the tests use it for.
 * ---------------------------------------------------------------------------------------------------------- */
size_t mtg_tuning_count(void);
/* entry i: its name (without the MTG_ prefix), default ("" = not set), kind ("cap", "ab", "test", "diag") and description; static strings */
int mtg_tuning_describe(size_t i, const char** name, const char** dflt, const char** kind, const char** what);
/* the current value of an entry, "" when it is not set; MTG_ERR_ARG: no such entry, or cap too small */
int mtg_tuning_get(const char* name, char* value, size_t cap);
/* value NULL or "": "not set" -- overriding the environment and the default (to return to the default, set the default string).  A flag
 * ("" by default) is ON for any value but "0": NAME=0 leaves it off.  An environment variable set to the empty string switches a flag on and
 * leaves a number unset. */
int mtg_tuning_set(const char* name, const char* value);

/* ------------------------------------------------------------------------------------------------------------
 * Bench support: measured ceiling of dependent random reads of line_bytes (16/32/64/128) over a table of the given size.
 * ---------------------------------------------------------------------------------------------------------- */
int mtg_bench_random_lines(uint64_t table_bytes, uint64_t n_chains, uint32_t chain_len, uint32_t line_bytes, double* ms, double* gbps);

#ifdef __cplusplus
}
#endif
#endif
