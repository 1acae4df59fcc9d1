/*
 * mtg_oracle_main.cpp -- command-line front end of the CPU ORACLE (TEST INFRASTRUCTURE ONLY, see
 * mtg_oracle.h).  Mirrors `MindTheGap fill` options (src/Filler.cpp:76-113) so the reference's shell
 * tests (test/simple_full_test.sh:124,163) can be replayed against the oracle.
 */
#include "mtg_oracle.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

int main(int argc, char** argv)
{
    if (argc < 2 || strcmp(argv[1], "fill") != 0) { fprintf(stderr, "usage: mtg_oracle fill (-in reads) (-bkpt f | -contig f) [-out p] ...\n"); return 1; }
    mtgo_params P; mtgo_default_params(&P);
    std::string in, bkpt, contig, out = "MindTheGap_oracle";
    int k = 31, amin = -1, amax = 0;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); } return argv[++i]; };
        if (a == "-in") in = val(); else if (a == "-bkpt") bkpt = val(); else if (a == "-contig") contig = val();
        else if (a == "-out") out = val(); else if (a == "-kmer-size") k = atoi(val());
        else if (a == "-abundance-min") { const char* v = val(); amin = strcmp(v, "auto") == 0 ? -1 : atoi(v); }
        else if (a == "-abundance-max") amax = atoi(val());
        else if (a == "-max-nodes") P.max_nodes = atoi(val()); else if (a == "-max-length") P.max_depth = atoi(val());
        else if (a == "-overlap") P.overlap = atoi(val()); else if (a == "-nb-cores") P.nb_cores = atoi(val());
        else if (a == "-fwd-only") P.fwd_only = 1; else if (a == "-filter") P.filter = 1; else if (a == "-extend") P.extend = 1;
        else if (a == "-max-memory" || a == "-max-disk" || a == "-verbose") val();
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
    }
    if (in.empty() || (bkpt.empty() == contig.empty())) { fprintf(stderr, "need -in and exactly one of -bkpt / -contig\n"); return 1; }
    mtgo_index* idx = mtgo_index_from_files(in.c_str(), k, amin, amax);
    if (!idx) { fprintf(stderr, "cannot build index\n"); return 1; }
    uint64_t nb_solid, nb_branching; mtgo_index_stats(idx, &nb_solid, &nb_branching);
    uint64_t stats[8] = {0}; double secs = 0;
    int rc = mtgo_fill_files(idx, &P, bkpt.empty() ? 1 : 0, bkpt.empty() ? contig.c_str() : bkpt.c_str(), out.c_str(), in.c_str(), stats, &secs);
    printf("kmer-size : %d\nabundance_min (used) : %d\nnb_solid_kmers : %llu\nnb_branching_nodes : %llu\n", k, mtgo_index_abundance_min(idx),
           (unsigned long long)nb_solid, (unsigned long long)nb_branching);
    printf("nb_input : %llu\nnb_filled : %llu\nas_multiple_sequence : %llu\nprobes : %llu\nabundance_lookups : %llu\nTime : %.3f s\n",
           (unsigned long long)stats[0], (unsigned long long)stats[1], (unsigned long long)stats[2], (unsigned long long)stats[3], (unsigned long long)stats[4], secs);
    mtgo_index_free(idx);
    return rc;
}
