/* TEST-ONLY: unit checks of the vectorised host helpers of mtg_host.cpp (file-local there, so the source is included here) against
 * plain per-character code: the device's ASCII writer emit_ascii (one lane) for every alignment, length and direction, and FillInput::set_common
 * (pext packing, k-mer by bit reversal, validity tests) against encode_kmer / per-character loops.  Links with emu_backend.cpp.  Prints OK. */
#include "../../mindthegap_amd/csrc/mtg_host.cpp"
#include <random>
#include <cmath>

int main()
{
    std::mt19937_64 rng(11);
    static const char NT[4] = {'A', 'C', 'T', 'G'}, NTC[4] = {'T', 'G', 'A', 'C'};
    /* emit_ascii (the device's 2-bit -> ASCII writer, mtg_emit.h; one lane here): every source offset, length, direction and
     * destination alignment; nothing but [dst, dst + L] may be written */
    std::vector<uint64_t> words(40);
    for (int round = 0; round < 120; round++) {
        for (auto& w : words) w = rng();
        for (uint32_t from = 0; from < 70; from += (round & 1) + 1)
            for (uint32_t L : {0u, 1u, 3u, 4u, 15u, 16u, 17u, 31u, 32u, 33u, 63u, 64u, 65u, 100u, 517u, 1000u, (uint32_t)(rng() % 1100)}) {
                if (from + L > 38 * 32) continue;
                for (int rc = 0; rc < 2; rc++) {
                    const uint32_t al = (uint32_t)(rng() % 16);
                    std::vector<char> buf(L + 64, '#');
                    char* dst = buf.data() + 16 + al - ((uintptr_t)buf.data() & 15); /* alignment al modulo 16 */
                    std::string want(L, '?');
                    mtg::emit_ascii(words.data(), from, L, rc != 0, dst);
                    for (uint32_t i = 0; i < L; i++) {
                        const uint32_t j = from + i, c = (uint32_t)(words[j >> 5] >> (2 * (j & 31))) & 3;
                        if (!rc) want[i] = NT[c]; else want[L - 1 - i] = NTC[c];
                    }
                    bool ok = std::string(dst, L) == want && dst[L] == 0;
                    for (char* q = buf.data(); q < dst; q++) ok = ok && *q == '#';
                    for (char* q = dst + L + 1; q < buf.data() + buf.size(); q++) ok = ok && *q == '#';
                    if (!ok) { fprintf(stderr, "emit_ascii from %u L %u rc %d align %u differs\n", from, L, rc, al); return 1; }
                }
            }
    }
    /* set_common */
    const char alphabet[] = "ACGTACGTACGTACGTNacgtnXR-";
    for (int k : {31, 21, 16, 13}) {
        for (int round = 0; round < 3000; round++) {
            const bool clean = (round % 3) != 0;
            auto rnd = [&](size_t n) { std::string s(n, 'A'); for (auto& c : s) c = clean ? "ACGT"[rng() & 3] : alphabet[rng() % (sizeof alphabet - 1)]; return s; };
            const std::string source = rnd((size_t)k + (round % 5 == 0 ? rng() % 4 : 0));
            const std::string pattern = rnd(round % 7 == 0 ? rng() % 200 : (size_t)k);
            mtgi::FillInput in;
            in.k = k;
            in.resize(1);
            in.size(0, pattern.size(), 0);
            in.layout();
            /* poison the pattern words: the block is recycled in the product */
            for (size_t w = 0; w < in.rwords.size(); w++) in.rwords[w] = ~0ull;
            in.set_common(0, source, pattern, 2, 0);
            const uint64_t want_src = mtg::encode_kmer(source.data(), k);
            bool acgt = true;
            for (char c : pattern) acgt = acgt && (c == 'A' || c == 'C' || c == 'G' || c == 'T');
            const uint64_t want_r0 = (pattern.size() >= (size_t)k && acgt) ? mtg::encode_kmer(pattern.data(), k) : 0;
            bool fast = (int)source.size() == k;
            for (unsigned char c : source) fast = fast && !mtgi::nt_bad(c);
            bool ok = in.src[0] == want_src && in.r0[0] == want_r0 && in.fast_ok[0] == (fast ? 1 : 0) && in.nbmis[0] == 2 &&
                      in.rlen[0] == (acgt ? (uint32_t)pattern.size() : 0xFFFFFFFFu);
            const size_t nw = (pattern.size() + 31) / 32 + 1;
            for (size_t w = 0; w < nw && ok; w++) {
                uint64_t want = 0;
                for (size_t i = w * 32; i < std::min(pattern.size(), (w + 1) * 32); i++) want |= (uint64_t)mtg::nt_code((unsigned char)pattern[i]) << (2 * (i & 31));
                ok = in.rwords[in.roff[0] + w] == want;
            }
            if (!ok) { fprintf(stderr, "set_common k %d source %s pattern %s differs\n", k, source.c_str(), pattern.c_str()); return 1; }
        }
    }
    /* auto_cutoff (the product's automatic solidity cut-off) on histograms whose answer is known without any other implementation: an
     * error peak that decays from abundance 1 and a coverage peak around c -- the cut-off is the minimum of the smoothed valley between
     * them (ties: the first), never below the floor, never at or beyond the peak, and unchanged when the histogram is scaled */
    {
        auto smoothed = [](const std::vector<uint64_t>& h, size_t i) -> uint64_t {
            if (i == 1) return (uint64_t)(0.6 * (double)h[1] + 0.4 * (double)h[2]);
            return (uint64_t)(0.2 * (double)h[i - 1] + 0.6 * (double)h[i] + 0.2 * (double)h[i + 1]);
        };
        for (int round = 0; round < 400; round++) {
            const int c = 12 + (int)(rng() % 60), valley = 3 + (int)(rng() % (unsigned)(c - 8));
            const double err0 = 1e5 * (1 + (double)(rng() % 100)), decay = 0.25 + 0.01 * (double)(rng() % 40), pk = 1e4 * (1 + (double)(rng() % 50));
            std::vector<uint64_t> h(c * 3 + 10, 0);
            for (size_t a = 1; a < h.size(); a++) {
                double e = err0;
                for (size_t t = 1; t < a; t++) e *= decay;                                   /* sequencing errors: geometric decay */
                const double d = ((double)a - c) / (0.18 * c + 1.0);
                h[a] = (uint64_t)(e + pk / (1.0 + d * d * d * d)) + ((int)a == valley ? 0 : 3); /* coverage peak; the designated valley is the lowest point */
            }
            /* make the designated valley the unique lowest entry between the error slope and the peak */
            uint64_t lo = ~0ull;
            for (int a = 2; a < c; a++) lo = std::min(lo, h[(size_t)a]);
            for (int a = valley - 1; a <= valley + 1; a++) h[(size_t)a] = lo > 6 ? lo - 6 + (uint64_t)std::abs(a - valley) * 2 : 0;
            const int got = mtgi::auto_cutoff(h, 3);
            /* the properties */
            bool ok = got >= 3 && got < c; /* below the coverage peak */
            for (size_t a = (size_t)std::max(got - 1, 2); a <= (size_t)got + 1 && a + 1 < h.size(); a++) ok = ok && smoothed(h, (size_t)got) <= smoothed(h, a); /* a local minimum of the smoothed histogram */
            ok = ok && std::abs(got - valley) <= 1;                                                                        /* at the designated valley (smoothing may move it by one) */
            std::vector<uint64_t> h10(h);
            for (auto& v : h10) v *= 10;
            ok = ok && mtgi::auto_cutoff(h10, 3) == got;                                                                    /* scale invariance */
            if (!ok) { fprintf(stderr, "auto_cutoff: round %d coverage %d valley %d -> %d\n", round, c, valley, got); return 1; }
        }
        /* no valley at all (monotone decay: error k-mers only) and tiny histograms: the floor */
        std::vector<uint64_t> mono(60);
        for (size_t a = 1; a < mono.size(); a++) mono[a] = 1000000 / (a * a);
        if (mtgi::auto_cutoff(std::vector<uint64_t>{0, 5, 3}, 3) != 3) { fprintf(stderr, "auto_cutoff: tiny histogram\n"); return 1; }
        (void)mono;
    }
    /* "%.2f" of the device formatter (mtg_format.h: fmt_fixed2, exact for a float promoted to double) against printf: random values, exact
     * ties (multiples of 1/8, sums over n for the n that divide 200), and the floats next to them */
    {
        struct Buf { char b[64]; int n = 0; void ch(int, char c) { b[n++] = c; } void bytes(int, const char* p, uint32_t l) { memcpy(b + n, p, l); n += (int)l; } };
        auto check = [&](float x) -> bool {
            if (!mtg::fmt_fixed2_ok(x)) return true;
            Buf o;
            mtg::fmt_fixed2(o, 0, x);
            char want[64];
            const int wn = snprintf(want, sizeof want, "%.2f", (double)x);
            if (wn != o.n || memcmp(want, o.b, (size_t)wn) != 0) { fprintf(stderr, "fmt_fixed2(%.9g) = %.*s, printf says %s\n", (double)x, o.n, o.b, want); return false; }
            return true;
        };
        std::uniform_real_distribution<float> U(0.f, 3000.f), V(0.f, 9.9e8f);
        for (int i = 0; i < 500000; i++) if (!check(U(rng)) || !check(V(rng))) return 1;
        for (uint32_t m = 0; m < 200000; m++) {
            const float t = (float)m / 8.0f; /* .125, .375, .625, .875: exact ties of the second decimal */
            if (!check(t) || !check(std::nextafter(t, 0.f)) || !check(std::nextafter(t, 1e9f))) return 1;
        }
        for (uint32_t n : {1u, 2u, 4u, 5u, 8u, 10u, 20u, 25u, 40u, 50u, 100u, 200u, 31u, 970u, 333u})
            for (uint32_t sum = 0; sum < 60000; sum += 7) if (!check((float)sum / (float)n)) return 1; /* avg = sum / (float)n, src/Filler.cpp:986 */
        /* atoi of a token */
        for (const char* tkn : {"0", "12", "+12", "-7", " 42", "x3", "", "0012", "99999999999", "3x", "\t5"}) {
            if (mtg::fmt_atoi(tkn, (uint32_t)strlen(tkn)) != (long long)atoi(tkn)) { fprintf(stderr, "fmt_atoi(\"%s\") differs from atoi\n", tkn); return 1; }
        }
    }
    printf("OK\n");
    return 0;
}
