"""Host-side mirror of the reference's Filler tool (src/Filler.hpp) on top of libmtgfill.so.

`Filler().run(argv)` takes the reference's own option strings (src/Filler.cpp:76-113) and writes the same output
files; `gap_fill_from_source` mirrors Filler::gapFillFromSource (src/Filler.hpp:188-189) for a batch of gaps."""
from .lib import FillParams, Gap, Index, fill_main


class Filler:
    STR_URI_BKPT, STR_URI_CONTIG, STR_URI_GRAPH, STR_URI_INPUT, STR_URI_OUTPUT = "-bkpt", "-contig", "-graph", "-in", "-out"
    STR_MAX_DEPTH, STR_MAX_NODES, STR_CONTIG_OVERLAP, STR_FILTER, STR_FWD_ONLY, STR_EXTEND = "-max-length", "-max-nodes", "-overlap", "-filter", "-fwd-only", "-extend"

    def __init__(self):
        self._nb_mis_allowed = 2  # src/Filler.cpp:56
        self._nb_gap_allowed = 0  # src/Filler.cpp:57

    def run(self, argv):
        """Equivalent of `MindTheGap fill <argv>`; returns the exit code (0 / 1)."""
        return fill_main(list(argv))

    def gap_fill_from_source(self, index: Index, gaps, max_nodes=100, max_depth=10000):
        """gaps: iterable of mindthegap_amd.Gap; returns the per-gap result dicts of Index.fill_batch."""
        return index.fill_batch(list(gaps), FillParams(max_nodes=max_nodes, max_depth=max_depth, nb_mis_allowed=self._nb_mis_allowed))
