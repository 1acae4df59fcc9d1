"""mtg_fill_text against mtg_fill_batch on the same strings: shared by the emulator suite (tests/test_emu_parity.py) and the GPU suite
(tests/test_gpu_parity.py)."""
import pytest


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))


def text_vs_strings(mtg, idx, gaps):
    """the same gaps through mtg_fill_batch (strings, marshalled on the host) and mtg_fill_text (one block of text, marshalled by the device
    code): the records must be identical"""
    params = mtg.FillParams()
    a = idx.fill_batch(gaps, params)
    tg = mtg.TextGaps(gaps)
    from mindthegap_amd import lib as L
    # twice: the block copied into the library's page-locked memory, and (round 4) uploaded from where it is once the caller has page-locked it
    for registered in (False, True):
        if registered:
            tg.register()
        h, nf, buf = idx.fill_prepared(tg, params)
        b = []
        for i in range(len(gaps)):
            r = L.load_library().mtg_results_get(h, i).contents
            b.append(dict(nb_nodes=r.nb_nodes, total_nt=r.total_nt, nb_terminal=r.nb_terminal, n=r.n_filled, ext=r.extension.decode(),
                          filled=[(r.filled[j].seq.decode(), r.filled[j].nb_errors_in_anchor, r.filled[j].target_index, r.filled[j].qual, r.filled[j].avg_coverage, r.filled[j].median_coverage)
                                  for j in range(r.n_filled)]))
        idx.free_results(h)
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert (x["nb_nodes"], x["total_nt"], x["nb_terminal"], len(x["filled"]), x["extension"]) == (y["nb_nodes"], y["total_nt"], y["nb_terminal"], y["n"], y["ext"])
            assert [(f["seq"], f["nb_errors_in_anchor"], f["target_index"], f["qual"], f["avg_coverage"], f["median_coverage"]) for f in x["filled"]] == y["filled"]
    tg.unregister()
    return a


def run(mtg, oracle_lib):
    """mtg_fill_text (strings as offsets into one block, encoded by k_marshal_text / k_marshal_targets -- here their per-gap functions on the
    host) against mtg_fill_batch on the same strings, including the ones the marshalling treats specially: lower case, N in source / pattern /
    dictionary key, a source longer than k, keys shorter and longer than k, a pattern shorter than k, empty dictionaries, several entries,
    repeated anchors and reverse attempts"""
    from mindthegap_amd.synth import SynthSet
    for het in (0, 4):
        S = SynthSet(nseq=60 if not het else 80, n_sites=40, seed=11, het_snps=het)
        o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
        km, ct = o.export()
        idx = mtg.Index.from_kmers(km, ct, 31)
        gaps = []
        for i in range(S.n_sites):
            l, r, _ = S.site(i)
            name = S.site_name(i)
            v = i % 10
            if v == 0:
                gaps.append(mtg.Gap(l.lower(), r, [(r.lower(), name, False)]))
            elif v == 1:
                gaps.append(mtg.Gap(l, r[:10] + "N" + r[11:], [(r, name, False)]))          # pattern with N: never stops early
            elif v == 2:
                gaps.append(mtg.Gap(l, r, [(r[:5] + "N" + r[6:], name, False), (r, name + "_b", True)]))
            elif v == 3:
                gaps.append(mtg.Gap(l + "ACGT", r, [(r + "TTGA", name, False)]))            # longer than k: the first k characters count
            elif v == 4:
                gaps.append(mtg.Gap(l, r[:20], [(r[:20], name, False)]))                    # shorter than k: unusable key, short pattern
            elif v == 5:
                gaps.append(mtg.Gap(l, r, []))
            elif v == 6:
                gaps.append(mtg.Gap(l, r, [(r, name, False)], is_anchor_repeated=True))
            elif v == 7:
                gaps.append(mtg.Gap(_rc(r), _rc(l), [(_rc(l), name, True)], reverse=True))
            elif v == 8:
                gaps.append(mtg.Gap(l[:15] + "N" + l[16:], r, [(r, name, False)]))          # N in the source: not the fast form
            else:
                gaps.append(mtg.Gap(l, r, [(r, name, False)]))
        res = text_vs_strings(mtg, idx, gaps)
        assert sum(1 for r in res if r["filled"]) >= 10
        idx.close()
    # malformed batches are refused, not read
    from mindthegap_amd import lib as L
    S = SynthSet(nseq=20, n_sites=4, seed=2)
    o = oracle_lib.Index.from_sequences([S.ascii(j) for j in range(S.nseq)], 31, 3, 40)
    km, ct = o.export()
    idx = mtg.Index.from_kmers(km, ct, 31)
    l, r, _ = S.site(0)
    tg = mtg.TextGaps([mtg.Gap(l, r, [(r, "x", False)])])
    tg.arrays["pattern_off"][0] = len(tg.text) + 5
    with pytest.raises(L.MtgError):
        idx.fill_prepared(tg)
    tg = mtg.TextGaps([mtg.Gap(l[:20], r, [(r, "x", False)])])
    with pytest.raises(L.MtgError):
        idx.fill_prepared(tg)
    idx.close()
