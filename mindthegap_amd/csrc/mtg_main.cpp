/*
 * mtg_main.cpp -- the `MindTheGap` executable: module dispatch of /root/reference/src/main.cpp:62-124.
 * Only the `fill` module is in scope; `find` is the reference's other module.
 */
#include "../../include/mtg_fill.h"
#include <cstdio>
#include <cstring>

int main(int argc, char** argv)
{
    if (argc < 2 || strcmp(argv[1], "-help") == 0 || strcmp(argv[1], "-h") == 0) {
        printf("\nMindTheGap (mindthegap_amd build: fill module on MI355X)\nUsage:\n   MindTheGap fill (-in <reads.fq> | -graph <graph>) (-bkpt <breakpoints.fa> | -contig <contigs.fa>) [options]\n   MindTheGap -version\n");
        return 1;
    }
    if (strcmp(argv[1], "-version") == 0 || strcmp(argv[1], "-v") == 0) { printf("MindTheGap version 2.3.0 (mindthegap_amd, HIP gfx950)\n"); return 0; }
    if (strcmp(argv[1], "fill") == 0) return mtg_fill_main(argc - 2, (const char* const*)(argv + 2));
    if (strcmp(argv[1], "find") == 0) { fprintf(stderr, "EXCEPTION: the find module is not part of this build; use the reference MindTheGap find and pass its .breakpoints to fill\n"); return 1; }
    fprintf(stderr, "EXCEPTION: unknown module '%s'\n", argv[1]);
    return 1;
}
