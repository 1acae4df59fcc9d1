"""mindthegap_amd -- MI355X-native drop-in for the hot path of `MindTheGap fill`.

The compute lives in libmtgfill.so (hand-written HIP for gfx950 behind the C ABI of include/mtg_fill.h);
this package is the thin host mirror used by tests, bench.py and Python callers.  There is no CPU
fallback: every call fails loudly when the library or a HIP device is missing."""
from .lib import (Batch, TextGaps, Filler, Index, FillParams, Gap, MtgError, build_library, cpu_budget, device_count, fill_main, last_batch_stats, library_path,
                  load_library, nw_matches, random_line_ceiling, tuning, tuning_set)

__all__ = ["Batch", "TextGaps", "Index", "FillParams", "Gap", "MtgError", "Filler", "build_library", "cpu_budget", "device_count", "fill_main", "last_batch_stats",
           "library_path", "load_library", "nw_matches", "random_line_ceiling", "tuning", "tuning_set"]
