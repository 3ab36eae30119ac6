// pfp.cpp -- Engine methods of the PFP producer: parse (A2), dictionary / parse
// structures (A3) and the suffix array of the text from them (A4).
//
// Text-sized quantities (trigger positions, phrase starts, stream offsets of groups and entries, the suffix-array
// column) are 32 bits wide in a narrow run and 40 / 64 bits in a wide one (wide.hpp); everything indexed by phrase,
// dictionary position or parse position is 32 bits in both.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "engine.hpp"
#include "pfp_kernels.hpp"
#include "pool.hpp"
#include "prims.hpp"

namespace mmt {

static int bit_width_u64(uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

static uint32_t read_u32(const uint32_t* d, hipStream_t s) {
    uint32_t v = 0;
    MMT_HIP(hipMemcpyAsync(&v, d, 4, hipMemcpyDeviceToHost, s));
    MMT_HIP(hipStreamSynchronize(s));
    return v;
}

// exclusive prefix sum of 32-bit counts into a table of stream offsets
static void offsets_from_counts(DevBuf<uint8_t>& temp, const uint32_t* counts, PosBuf& out, size_t n, hipStream_t s) {
    if (out.wide()) prims::exclusive_sum_u32_to_u64(temp, counts, out.p64(), n, s);
    else prims::exclusive_sum_u32(temp, counts, out.p32(), n, s);
}

// MMT_MEM_TRACE=1: live / peak bytes of the device heap at the stage boundaries (stderr)
static void mem_mark(int device, const char* what) {
    static const bool on = std::getenv("MMT_MEM_TRACE") != nullptr;
    if (!on) return;
    const pool::Stats s = pool::stats(device);
    std::fprintf(stderr, "[mem] %-28s live %7.2f GB  peak %7.2f GB  mapped %7.2f GB\n", what, s.live / 1e9, s.peak / 1e9, s.mapped / 1e9);
}

// A2 + the dictionary half of A3: phrases, distinct phrases, dictionary text with its suffix
// array / LCP, phrase ranks, parse.  Requires build_text() to have run.
// keep_dict_inputs: the phrase table and V stay (PREFIX.dict is written from them: -P / parse_only).
void Engine::pfp_parse(uint32_t w, uint32_t p, bool keep_dict_inputs) {
    PfpState& S = *pfp_;
    const uint64_t n = n_;
    const bool W = wide_;
    const bool slim = (lean_ || wide_) && !keep_dict_inputs;       // give scratch back as soon as it is dead
    if (w < 1 || w > 32 || p < 1) throw std::runtime_error("PFP window must be in [1, 32] and the modulus positive");
    std::vector<uint64_t> hist;
    d2h(hist, d_hist_.get(), 256, stream_);
    if (hist[0] || hist[1] || hist[2])          // newscan.hpp:318: characters <= Dollar are not allowed
        throw std::runtime_error("input contains bytes <= 0x02, which the prefix-free parse reserves");
    S.w = w; S.p = p; S.have_parse = false; S.bwt_ready = false;
    hipStream_t st = stream_;
    EventPair e0, e1, e2, e3, e4;

    // -- triggers, phrase boundaries
    e0.start(st);
    const TextRef v = text_ref();                  // V = Dollar . T . Dollar^w lives in the text buffer (Engine::text_ptr), or packed (textref.hpp)
    const uint32_t tb = pk::trigger_blocks(n);
    // (the cut bits double as the rank / successor structure of the guided sort: whole blocks of 4096 positions, zero padded)
    auto trigger_pass = [&]() {
        S.tmask.ensure((size_t)(((n + 64) / 4096 + 2) * 256)); S.tcnt.ensure((size_t)tb + 1);
        MMT_HIP(hipMemsetAsync(S.tmask.get(), 0, S.tmask.bytes(), st));
        pk::trigger_masks(v, n, w, p, S.tmask.get(), S.tcnt.get(), st);
    };
    S.toff.ensure((size_t)tb + 1);
    trigger_pass();
    prims::exclusive_sum_u32(d_temp_, S.tcnt.get(), S.toff.get(), tb, st);
    S.err.ensure(16);
    {
        const uint64_t cuts64 = (uint64_t)read_u32(S.toff.get() + (tb - 1), st) + read_u32(S.tcnt.get() + (tb - 1), st);
        if (cuts64 >= 0x7ffffffdull) throw std::runtime_error("more than 2^31 - 2 phrases (newscan.hpp:44)");
        S.n_cuts = (uint32_t)cuts64;                                   // size the cut list exactly
    }
    S.cuts.ensure((size_t)S.n_cuts + 1, W);
    pk::trigger_cuts(S.tmask.get(), n, S.toff.get(), S.cuts.get(), W, st);
    const uint32_t m = S.n_phrases = S.n_cuts + 1;
    S.pstart.ensure(m, W); S.plen.ensure(m);
    pk::phrase_bounds(S.cuts.get(), S.n_cuts, n, w, S.pstart.get(), S.plen.get(), W, st);
    if (slim) { S.tcnt.release(); S.toff.release(); S.cuts.release(); }
    e0.stop(st);
    mem_mark(device_, "triggers + phrases");

    // -- distinct phrases: fingerprints, sort, verified grouping
    e1.start(st);
    // A packed text is a text that fills the device (configs[4]: 143 GB packed + 72 GB of cut bits): the cut bits -- an eighth
    // of a byte per character, needed again by the bucket-wise producer -- make room for the 52 bytes per phrase of this
    // stage and are computed once more behind it (one more pass over the text: ~1 ms per G characters)
    const bool tmask_again = packed_ && slim;
    if (tmask_again) { MMT_HIP(hipStreamSynchronize(st)); S.tmask.release(); }
    S.h1.ensure(m); S.pinfo.ensure((size_t)m * 16 + 16); S.hk_b.ensure(m);
    S.iota.ensure(m); S.order.ensure(m);
    pk::phrase_hash(v, S.pstart.get(), S.plen.get(), m, S.h1.get(), S.pinfo.get(), W, st);
    pk::iota(S.iota.get(), m, st);
    S.dflags.ensure(m); S.scan.ensure(m);
    for (int attempt = std::getenv("MMT_PFP_TWO_FINGERPRINTS") ? 1 : 0;; attempt++) {     // the variable forces the rare path (tests)
        if (attempt == 0) {
            // order by the first fingerprint alone; equal phrases are adjacent unless two different phrases share it
            prims::sort_pairs_u64_u32(d_temp_, S.h1.get(), S.hk_b.get(), S.iota.get(), S.order.get(), m, 0, 64, st);
        } else {
            // (rare) order by both fingerprints: stable sort by the second, then by the first
            S.h2.ensure(m); S.hk_a.ensure(m); S.ord_a.ensure(m);
            pk::second_fingerprint(S.pinfo.get(), m, S.h2.get(), st);
            prims::sort_pairs_u64_u32(d_temp_, S.h2.get(), S.hk_a.get(), S.iota.get(), S.ord_a.get(), m, 0, 64, st);
            pk::gather_u64(S.h1.get(), S.ord_a.get(), m, S.hk_a.get(), st);
            prims::sort_pairs_u64_u32(d_temp_, S.hk_a.get(), S.hk_b.get(), S.ord_a.get(), S.order.get(), m, 0, 64, st);
        }
        MMT_HIP(hipMemsetAsync(S.err.get(), 0, 16, st));
        pk::mark_distinct(S.order.get(), S.hk_b.get(), S.pinfo.get(), v, m, S.dflags.get(), S.err.get(), st);
        prims::inclusive_sum_u32(d_temp_, S.dflags.get(), S.scan.get(), m, st);
        uint32_t flags2[2] = {0, 0};
        MMT_HIP(hipMemcpyAsync(flags2, S.err.get(), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        if (flags2[0])
            throw std::runtime_error("phrase fingerprint collision (both fingerprints); refusing to merge different phrases");
        if (attempt == 0 && flags2[1]) continue;
        break;
    }
    const uint32_t D = S.n_distinct = read_u32(S.scan.get() + (m - 1), st);
    S.pid.ensure(m); S.rep.ensure(D); S.dlen.ensure(D); S.dstart.ensure(D);
    pk::assign_distinct(S.order.get(), S.scan.get(), S.dflags.get(), S.plen.get(), m, S.pid.get(), S.rep.get(),
                        S.dlen.get(), st);
    if (slim) {
        S.h1.release(); S.h2.release(); S.pinfo.release(); S.hk_a.release(); S.hk_b.release(); S.iota.release();
        S.ord_a.release(); S.order.release(); S.dflags.release(); S.scan.release();
    }
    if (tmask_again) { trigger_pass(); S.tcnt.release(); }
    e1.stop(st);
    mem_mark(device_, "distinct phrases");

    // -- dictionary text (distinct phrases in fingerprint order; ranks come from its suffix array)
    e2.start(st);
    prims::exclusive_sum_u32(d_temp_, S.dlen.get(), S.dstart.get(), D, st);
    {
        // the 32-bit prefix sum wraps silently: the length of the dictionary is added up in 64 bits, always
        uint64_t dict_len64 = 0;
        {
            DevBuf<uint64_t> total;
            total.ensure(1);
            pk::sum_u32(S.dlen.get(), D, total.get(), st);
            MMT_HIP(hipMemcpyAsync(&dict_len64, total.get(), 8, hipMemcpyDeviceToHost, st));
            MMT_HIP(hipStreamSynchronize(st));
            dict_len64 += 1;                           // (the lengths count their terminators; + the final 0x00)
        }
        // Little redundancy between the documents (the anchor next to one other whole genome): the dictionary is as
        // large as half the text, and neither its 32-bit suffix array nor the tables of its suffixes fit.  The text
        // suffixes are then sorted themselves, with the parse as the tie-breaker (guided.cpp).
        S.guided = false;
        if (!keep_dict_inputs) {
            const char* env = std::getenv("MUMEMTO_PRODUCER");
            const double tables = 46.0 * (double)dict_len64, sorter = 49.0 * (double)std::max<uint64_t>(dict_len64, m);
            const double need = tables + sorter;
            S.guided = producer_ == 3 || producer_ == 4 || pfp_want_guided_ || (env && (std::string(env) == "guided" || std::string(env) == "expand")) ||
                       dict_len64 >= 0xffffff00ull ||
                       (producer_ == 0 && need > 0.9 * (double)pool::available(device_));
            // Expansion: the bucket-wise producer sorts ONE representative per (distinct phrase, offset) -- the valid suffixes of
            // the dictionary nobody holds -- and the emitter expands them by the phrases' inverted lists (guided.cpp).  It pays
            // by the redundancy of the collection: a rank's share of whole genomes, {anchor + 12}, has 79 G text suffixes and 14 G
            // of those.  A producer named "guided" stays the plain one; MUMEMTO_EXPAND=0 | 1 overrides.
            S.expand = false;
            if (S.guided) {
                const bool named_plain = producer_ == 3 || (env && std::string(env) == "guided");
                // (... below 2^38 characters: a text that fills the device -- every rank of configs[4] holds 573 G characters --
                // leaves the inverted lists and the batches of representatives too little room next to each other; there the
                // plain producer with its staging list stays)
                // (round 6: ... and beyond as well -- with the staging list of representatives, the trigger bits made slice by
                // slice and the heap at nine tenths a rank's share of 573 G characters runs through expansion in 69 s where the plain
                // producer took 104, at the same peak)
                S.expand = producer_ == 4 || (env && std::string(env) == "expand") ||
                           (!named_plain && (double)dict_len64 < 0.5 * (double)n);
                if (const char* x = std::getenv("MUMEMTO_EXPAND")) S.expand = std::atoi(x) != 0;
            }
            if (S.guided) {
                S.dict_len = 0;
                e2.stop(st);
                mem_mark(device_, "dictionary text");
                S.ms[0] = e0.ms(); S.ms[1] = e1.ms(); S.ms[2] = e2.ms();
                S.have_parse = false;
                return;
            }
        }
        if (slim) S.tmask.release();
        if (dict_len64 >= 0xffffff00ull)
            throw std::runtime_error("PFP dictionary of " + std::to_string(dict_len64) + " bytes exceeds the 32-bit "
                                     "dictionary of this build (too little redundancy between the documents)");
        S.dict_len = (uint32_t)dict_len64;
    }
    const uint32_t nd = S.dict_len;
    S.dict.ensure((size_t)nd + 64); S.dinfo.ensure(nd);
    MMT_HIP(hipMemsetAsync(S.dict.get() + nd, 0, 64, st));
    // room for the byte before each position in the phrase-id word (MMT_PFP_NO_PACK: the other path, for tests)
    const bool pack_prev = D < (1u << 24) && !std::getenv("MMT_PFP_NO_PACK");
    pk::copy_dict(v, S.pstart.get(), S.plen.get(), S.rep.get(), S.dstart.get(), D, S.dict.get(),
                  S.dinfo.get(), nd, pack_prev, W, st);
    if (slim) S.dstart.release();
    e2.stop(st);
    mem_mark(device_, "dictionary text");

    // -- suffix array of the dictionary (dictionary.hpp:133) ...
    e3.start(st);
    uint8_t code[256];
    int sigma = 0;
    // (0x00 -- once, the last byte of the dictionary -- and the phrase terminator 0x01 share code 0 with the padding behind
    // the end: nothing is compared behind a terminator, terminators are ordered by position.  Dollar, the document
    // separator, A C G T and N then fit three bits: 21 characters per 64-bit key instead of 15)
    for (int c = 0; c < 256; c++) code[c] = (c > 1 && (hist[c] || c == 2)) ? (uint8_t)(++sigma) : 0;
    const int bits = std::max(1, bit_width_u64((uint64_t)sigma));
    // every 0x01 (end of a phrase) is a unique terminator, ordered by position: what follows it never matters, so
    // a suffix is final as soon as the compared prefix reaches the end of its phrase (one key bit says so)
    int chars = std::min(63 / bits, 63);
    if (const char* e = std::getenv("MMT_DICT_KEY_CHARS")) chars = std::max(4, std::min(chars, std::atoi(e)));      // (tuning aid)
    d_code_.ensure(256);
    MMT_HIP(hipMemcpyAsync(d_code_.get(), code, 256, hipMemcpyHostToDevice, st));
    S.sa_d.ensure(nd); S.rank_d.ensure(nd);
    sorter_.reserve(slim ? nd : std::max(nd, m));
    // long runs of one symbol (assembly gaps, homopolymers): their ends are listed while the keys are packed, and the sorter
    // orders the suffixes inside them by (end of the run, what follows) instead of doubling its way through (sorter.hpp)
    const uint32_t run_cap = 1u << 20;
    DevBuf<uint32_t> run_ends, run_sorted, run_count;
    const bool want_runs = bits <= 3 && !std::getenv("MMT_NO_RUN_REFINE");          // (MMT_NO_RUN_REFINE: plain doubling, A/B)
    if (want_runs) {
        run_ends.ensure(run_cap); run_count.ensure(4);
        MMT_HIP(hipMemsetAsync(run_count.get(), 0, 16, st));
    }
    k::pack_keys(S.dict.get(), nd, d_code_.get(), bits, chars, (uint32_t)code[1], sorter_.keys_in(), sorter_.vals_in(), st,
                 want_runs ? run_ends.get() : nullptr, want_runs ? run_count.get() : nullptr, run_cap);
    RunRefine runs;
    if (want_runs) {
        const uint32_t found = read_u32(run_count.get(), st);
        if (found && found <= run_cap) {                     // (more than the list holds: plain doubling)
            run_sorted.ensure((size_t)found * 3);            // sorted ends | two columns of dummy values
            MMT_HIP(hipMemsetAsync(run_sorted.get() + found, 0, (size_t)found * 4, st));
            prims::sort_pairs_u32_u32(d_temp_, run_ends.get(), run_sorted.get(), run_sorted.get() + found,
                                      run_sorted.get() + 2 * (size_t)found, found, 0, 32, st);
            runs.text = S.dict.get(); runs.n = nd; runs.code = d_code_.get(); runs.bits = bits; runs.chars = chars; runs.sigma = sigma;
            runs.ends = run_sorted.get(); runs.n_ends = found;
        }
    }
    S.rounds_dict = sorter_.sort(nd, bits * chars + 1, (uint64_t)chars, S.sa_d.get(), S.rank_d.get(), d_temp_, st, true,
                                 runs.n_ends ? &runs : nullptr);
    S.run_refined = sorter_.run_refined();
    // (the sorter's 49 bytes per dictionary character were the idle half of the run's peak while the tables of the dictionary
    // were built: a fresh process maps, and gives back at its end, every byte of that peak)
    if (slim) { MMT_HIP(hipStreamSynchronize(st)); sorter_.release(); S.rank_d.release(); }
    e3.stop(st);
    mem_mark(device_, "dictionary suffix array");
    // ... the groups of equal proper phrase suffixes and the phrase ranks
    e4.start(st);
    S.esuf.ensure(nd); S.ephr.ensure(nd); S.ebw.ensure(nd);
    pk::entry_info(S.sa_d.get(), S.dinfo.get(), S.dict.get(), nd, pack_prev, S.esuf.get(), S.ephr.get(), S.ebw.get(), st);
    if (slim) S.dinfo.release();
    S.gflag.ensure(nd); S.pflag.ensure(nd); S.vflag.ensure(nd); S.gscan.ensure(nd); S.pscan.ensure(nd);
    S.prank.ensure(D); S.parse.ensure(m);
    // LCP array of the dictionary (dictionary.hpp:133 takes it from gsacak): irreducible entries compared directly, the
    // others by the PLCP chain along the dictionary, then gathered into suffix-array order
    {
        DevBuf<uint32_t> plcp_d, counts, huge;
        DevBuf<uint8_t> longs;
        plcp_d.ensure((size_t)nd + 16); counts.ensure(4); S.lcp_d.ensure(((size_t)nd + 3) / 4 * 4 + 16);
        uint32_t cap = std::max<uint32_t>(nd / 256 + 4096, 1u << 16), found = 0;
        for (int attempt = 0;; attempt++) {
            longs.ensure((size_t)cap * sizeof(k::LongLcpLim));
            pk::dict_irreducible(S.dict.get(), nd, S.sa_d.get(), S.esuf.get(), S.ebw.get(), plcp_d.get(), longs.get(),
                                 counts.get(), cap, st);
            found = read_u32(counts.get(), st);
            if (found <= cap) break;
            if (attempt) throw std::runtime_error("long-match list overflow in the dictionary's LCP construction");
            cap = found + 1024;                              // rare: once more with the exact size
        }
        if (found) {
            huge.ensure((size_t)found + 1);
            k::long_lcp_lim(S.dict.get(), nd, longs.get(), found, plcp_d.get(), huge.get(), counts.get() + 1, st);
        }
        d_temp_.ensure(k::plcp_running_max_scratch(nd));
        k::plcp_running_max(plcp_d.get(), nd, d_temp_.get(), st);
        SaCol sd; sd.lo = S.sa_d.get(); sd.hi = nullptr;
        k::lcp_gather(plcp_d.get(), sd, 0, nd, S.lcp_d.get(), st);
        pk::dict_lcp_clamp(S.lcp_d.get(), S.esuf.get(), nd, st);
        MMT_HIP(hipStreamSynchronize(st));
    }
    S.segmin.ensure(nd);
    pk::group_flags(S.esuf.get(), S.lcp_d.get(), nd, w, S.gflag.get(), S.pflag.get(), S.vflag.get(), S.segmin.get(), st);
    prims::inclusive_segmin_u64(d_temp_, S.segmin.get(), S.segmin.get(), nd, st);
    if (slim) { MMT_HIP(hipStreamSynchronize(st)); S.lcp_d.release(); }
    prims::inclusive_sum_u32(d_temp_, S.gflag.get(), S.gscan.get(), nd, st);
    prims::inclusive_sum_u32(d_temp_, S.pflag.get(), S.pscan.get(), nd, st);
    pk::phrase_ranks(S.esuf.get(), S.ephr.get(), S.pscan.get(), nd, S.prank.get(), st);
    pk::parse_ranks(S.pid.get(), S.prank.get(), m, S.parse.get(), st);
    S.n_groups = read_u32(S.gscan.get() + (nd - 1), st);
    if (slim) { S.pflag.release(); S.pscan.release(); S.sa_d.release(); S.dict.release(); }
    // (the dictionary's suffix sort was the run's peak: 49 bytes per dictionary character)
    if (slim && one_shot_) pool::shrink_async(device_);
    e4.stop(st);
    mem_mark(device_, "dictionary groups + LCP");
    S.ms[0] = e0.ms(); S.ms[1] = e1.ms(); S.ms[2] = e2.ms(); S.ms[3] = e3.ms(); S.ms[4] = e4.ms();
    S.have_parse = true;
}

// A3 (parse half) + A4: the text suffixes ordered by (group of the phrase suffix, rank of the following parse suffix).
// pfp_prepare builds every table the emitter needs (or hands over to guided_prepare); the columns themselves are
// emitted window by window while the scan consumes them (pfp_stream below) and are never stored as a whole --
// pfp_lcp_mum.hpp:197: one update() per suffix.
void Engine::pfp_prepare(uint32_t w, uint32_t p) {
    PfpState& S = *pfp_;
    auto t0 = std::chrono::steady_clock::now();
    S.emit_ready = false;
    pfp_parse(w, p, false);
    if (S.guided) guided_prepare();
    else pfp_prepare_emitter(w);
    S.ms[7] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void Engine::pfp_prepare_emitter(uint32_t w) {
    PfpState& S = *pfp_;
    const uint64_t n = n_;
    const bool W = wide_;
    const bool slim = lean_ || wide_;
    hipStream_t st = stream_;
    const uint32_t m = S.n_phrases, D = S.n_distinct;
    EventPair e5, e6;
    e5.start(st);
    S.sa_p.ensure(m); S.isa_p.ensure(m);
    const int pbits = std::max(1, bit_width_u64((uint64_t)D));
    const int pchars = std::max(1, 64 / pbits);
    sorter_.reserve(m);
    pk::pack_keys_u32(S.parse.get(), m, pbits, pchars, sorter_.keys_in(), sorter_.vals_in(), st);
    S.rounds_parse = sorter_.sort(m, pbits * pchars, (uint64_t)pchars, S.sa_p.get(), S.isa_p.get(), d_temp_, st);
    if (slim) { MMT_HIP(hipStreamSynchronize(st)); sorter_.release(); S.isa_p.release(); S.parse.release(); }
    // LCP of adjacent parse suffixes + range minima: every LCP value of the stream follows from them locally
    S.plcp.build(text_ref(), n + 1 + w, S.sa_p.get(), S.pid.get(), S.pstart.get(), W, m, d_temp_, st);
    e5.stop(st);
    mem_mark(device_, "parse suffix array + LCP");

    e6.start(st);
    const int shift = bit_width_u64((uint64_t)m + 1);
    const uint32_t nd = S.dict_len;
    // inverted lists: parse positions ordered by (phrase, rank of the following parse suffix)
    const uint32_t pos_bits = W ? (uint32_t)bit_width_u64(n + w + 1) : 32u;
    if ((uint32_t)shift + pos_bits > 64)
        throw std::runtime_error("parse rank and text position do not fit one 64-bit occurrence record");
    S.occ_start.ensure((size_t)D + 2); S.occ.ensure(m);
    S.occ_ids.ensure((size_t)m + 1); S.occ_ts.ensure((size_t)m + 1);
    {
        DevBuf<uint32_t> k_own, v_own;                                   // slim runs gave the sorter's columns back
        uint32_t *k_in, *v_in;
        if (slim) { k_own.ensure((size_t)m + 1); v_own.ensure((size_t)m + 1); k_in = k_own.get(); v_in = v_own.get(); }
        else {
            sorter_.u32_a().ensure((size_t)m + 1); sorter_.u32_b().ensure((size_t)m + 1);     // scratch
            k_in = sorter_.u32_a().get(); v_in = sorter_.u32_b().get();
        }
        pk::occ_sequence(S.sa_p.get(), S.pid.get(), m, D, k_in, v_in, st);
        prims::sort_pairs_u32_u32(d_temp_, k_in, S.occ_ids.get(), v_in, S.occ_ts.get(), (size_t)m + 1, 0,
                                  std::max(1, bit_width_u64((uint64_t)D)), st);
        S.occ_sl.ensure(m);
        pk::occ_finish(S.occ_ids.get(), S.occ_ts.get(), S.sa_p.get(), S.pstart.get(), m, S.occ_start.get(),
                       S.occ.get(), pos_bits, S.plcp.sl.get(), S.occ_sl.get(), W, st);
        MMT_HIP(hipStreamSynchronize(st));
    }
    if (slim) { S.occ_ids.release(); S.occ_ts.release(); S.sa_p.release(); S.pid.release(); S.pstart.release(); }
    // valid dictionary suffixes in dictionary suffix-array order, compacted ("entries")
    S.vscan.ensure(nd); S.ptab.ensure((size_t)D * 16 + 16);
    prims::exclusive_sum_u32(d_temp_, S.vflag.get(), S.vscan.get(), nd, st);
    const uint32_t E = S.n_entries = read_u32(S.vscan.get() + (nd - 1), st) + read_u32(S.vflag.get() + (nd - 1), st);
    pk::phrase_table(S.occ_start.get(), S.plen.get(), S.rep.get(), D, S.ptab.get(), st);
    S.ce_cnt.ensure(E); S.ce_eoff.ensure(E, W); S.ce_first.ensure(E); S.ce_offm1.ensure(E); S.ce_gs.ensure(E);
    S.ce_bwt.ensure(E); S.ce_hl.ensure(E); S.ce_slen.ensure(E);
    if (!S.gscan.get() || !S.segmin.get()) throw std::runtime_error("PFP tables released too early");
    pk::entry_compact(S.esuf.get(), S.ephr.get(), S.ebw.get(), S.gflag.get(), S.gscan.get(), S.vflag.get(), S.vscan.get(),
                      S.segmin.get(), S.ptab.get(), nd, S.ce_cnt.get(), S.ce_first.get(), S.ce_offm1.get(), S.ce_bwt.get(),
                      S.ce_gs.get(), S.ce_hl.get(), S.ce_slen.get(), st);
    offsets_from_counts(d_temp_, S.ce_cnt.get(), S.ce_eoff, E, st);
    {
        const uint64_t total = S.ce_eoff.read(E - 1, st) + read_u32(S.ce_cnt.get() + (E - 1), st);
        if (total != n + 1) throw std::runtime_error("PFP expansion does not cover the text exactly once");
    }
    if (slim) {
        S.esuf.release(); S.ephr.release(); S.ebw.release(); S.gflag.release(); S.vflag.release(); S.vscan.release();
        S.ptab.release(); S.plen.release(); S.segmin.release();
    }
    S.occ12.release(); S.emit_pos_bits = pos_bits; S.emit_w = w;
    pfp_emit_codes(shift);
    pfp_group_tables(E, S.n_groups, 0, n + 1, slim);
    S.emit_launches = 0;
    S.emit_ready = true;
    S.bwt_ready = true;
    e6.stop(st);
    mem_mark(device_, "lists + emitter tables");
    S.ms[5] = e5.ms(); S.ms[6] = e6.ms();
    sort_rounds_ = S.rounds_dict;
}

// Group tables of the emitter from the entry tables S.ce_* (E entries in suffix-array order of their phrase suffixes, G groups
// of equal phrase suffixes; the entries' output offsets cover [out_lo, out_hi)): first entry and first output position of
// each group, |alpha| and the LCP at the head of the group, the oversized groups, the first group of every output tile.
// The parse proper calls it once for the whole stream (out_lo = 0, out_hi = n + 1); the expansion of the bucket-wise producer
// (guided.cpp) once per batch, for the piece of the stream the batch covers.
void Engine::pfp_group_tables(uint32_t E, uint32_t G, uint64_t out_lo, uint64_t out_hi, bool slim) {
    PfpState& S = *pfp_;
    const bool W = wide_;
    hipStream_t st = stream_;
    const uint64_t TILE = pk::emit_tile();
    S.n_groups = G; S.n_entries = E;
    S.sege.ensure((size_t)G + 2); S.segb.ensure((size_t)G + 2, W);
    prims::select_indices_u32flags(d_temp_, S.ce_gs.get(), S.sege.get(), S.err.get(), E, st);
    if (read_u32(S.err.get(), st) != G) throw std::runtime_error("PFP group count mismatch");
    pk::gather_pos(S.ce_eoff.get(), S.sege.get(), G, S.segb.get(), W, st);
    MMT_HIP(hipMemcpyAsync(S.sege.get() + G, &E, 4, hipMemcpyHostToDevice, st));
    S.segb.write(G, out_hi, st);
    // per group: |alpha| and the LCP with the phrase suffix of the group before (the minimum of the dictionary's LCP array)
    S.ghead.ensure((size_t)G * 2);
    pk::group_heads(S.sege.get(), S.ce_hl.get(), S.ce_slen.get(), G, S.ghead.get(), st);
    if (slim) { MMT_HIP(hipStreamSynchronize(st)); S.ce_hl.release(); S.ce_slen.release(); }
    // groups larger than one LDS tile of the emitter get compact slots in the fallback arrays
    S.gscan.ensure(std::max<size_t>(G, 1));
    uint32_t* osize = S.gscan.get();
    MMT_HIP(hipMemsetAsync(S.err.get(), 0, 16, st));
    pk::oversize(S.segb.get(), G, osize, S.err.get(), W, st);
    S.fb_group.ensure((size_t)G + 1);
    prims::select_indices_u32flags(d_temp_, osize, S.fb_group.get(), S.err.get() + 3, G, st);
    {
        uint32_t e4[4];
        MMT_HIP(hipMemcpyAsync(e4, S.err.get(), 16, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        if (e4[2]) throw std::runtime_error("a group of 2^32 or more equal phrase suffixes is not supported");
        S.n_fallback = e4[3];
    }
    const uint32_t F = S.n_fallback;
    S.fb_size.ensure((size_t)F + 2); S.fb_off.ensure((size_t)F + 2, W); S.fb_start.ensure((size_t)F + 2, W);
    S.h_fb_off.assign(1, 0); S.h_fb_start.clear();
    if (F) {
        k::gather_u32_idx32(osize, S.fb_group.get(), F, S.fb_size.get(), st);
        MMT_HIP(hipMemsetAsync(S.fb_size.get() + F, 0, 4, st));
        offsets_from_counts(d_temp_, S.fb_size.get(), S.fb_off, (size_t)F + 1, st);
        pk::gather_pos(S.segb.get(), S.fb_group.get(), F, S.fb_start.get(), W, st);
        // the launch plan of every window needs both tables on the host
        S.h_fb_off.assign((size_t)F + 1, 0); S.h_fb_start.assign(F, 0);
        if (W) {
            MMT_HIP(hipMemcpyAsync(S.h_fb_off.data(), S.fb_off.get(), ((size_t)F + 1) * 8, hipMemcpyDeviceToHost, st));
            MMT_HIP(hipMemcpyAsync(S.h_fb_start.data(), S.fb_start.get(), (size_t)F * 8, hipMemcpyDeviceToHost, st));
            MMT_HIP(hipStreamSynchronize(st));
        } else {
            std::vector<uint32_t> a32, b32;
            d2h(a32, S.fb_off.p32(), (size_t)F + 1, st);
            d2h(b32, S.fb_start.p32(), F, st);
            for (size_t i = 0; i <= F; i++) S.h_fb_off[i] = a32[i];
            for (size_t i = 0; i < F; i++) S.h_fb_start[i] = b32[i];
        }
    }
    // chunks of pk::EMIT_BIG_CHUNK output positions per oversized group, as a running count (k_emit_big)
    S.h_fb_chunk0.assign((size_t)F + 1, 0);
    for (uint32_t f = 0; f < F; f++)
        S.h_fb_chunk0[f + 1] = S.h_fb_chunk0[f] + (S.h_fb_off[f + 1] - S.h_fb_off[f] + pk::EMIT_BIG_CHUNK - 1) / pk::EMIT_BIG_CHUNK;
    S.fb_chunk0.ensure((size_t)F + 2);
    MMT_HIP(hipMemcpyAsync(S.fb_chunk0.get(), S.h_fb_chunk0.data(), ((size_t)F + 1) * 8, hipMemcpyHostToDevice, st));
    MMT_HIP(hipStreamSynchronize(st));
    if (slim) S.gscan.release();

    // the emitter's arguments, shared by every window of these tables
    S.tiles = (out_hi + TILE - 1) / TILE;
    S.tile_base = out_lo / TILE;
    pk::EmitArgs& ea = S.ea;
    ea = pk::EmitArgs();
    ea.wide = W;
    ea.segb = S.segb.get(); ea.sege = S.sege.get(); ea.n_groups = G;
    ea.ce_eoff = S.ce_eoff.get(); ea.ce_cnt = S.ce_cnt.get(); ea.ce_first = S.ce_first.get();
    ea.ce_offm1 = S.ce_offm1.get(); ea.ce_bwt = S.ce_bwt.get(); ea.ce_gs = S.ce_gs.get();
    ea.n = n_;
    ea.fb_group = S.fb_group.get(); ea.fb_off = S.fb_off.get(); ea.n_fb = F; ea.fb_base = 0;
    ea.err = S.err.get();
    ea.ghead = S.ghead.get();
    ea.occ = S.occ.get(); ea.occ_sl = S.occ_sl.get(); ea.occ12 = S.occ12.get(); ea.pos_bits = S.emit_pos_bits;
    ea.rmq = S.plcp.view(); ea.w = S.emit_w;
    ea.bwt_code = S.bwt_code.get(); ea.fb_bits = S.fb_bits;
    S.tile_first.ensure((size_t)(S.tiles - S.tile_base) + 4);
    pk::tile_first(S.segb.get(), G, S.tiles, S.tile_first.get(), W, st, S.tile_base);
    MMT_HIP(hipMemsetAsync(S.err.get(), 0, 64, st));
}

// the byte before every suffix of an oversized group rides in the low bits of its sort key when the text has at most 16
// different bytes and the parse rank leaves room (MMT_PFP_NO_BWT_CODE: read it from the text instead)
void Engine::pfp_emit_codes(int shift) {
    PfpState& S = *pfp_;
    hipStream_t st = stream_;
    S.decode = pk::BwtDecode{};
    S.fb_bits = 0; S.key_shift = shift;
    if (!std::getenv("MMT_PFP_NO_BWT_CODE")) {
        std::vector<uint64_t> hist;
        d2h(hist, d_hist_.get(), 256, st);
        std::vector<uint8_t> code(256, 0);
        uint32_t kinds = 0;
        bool fits = true;
        for (int b = 0; b < 256; b++)
            if (b == 0 || hist[b]) {                              // 0 stands before the first text position
                if (kinds == 16) { fits = false; break; }
                S.decode.byte[kinds] = (uint8_t)b; code[b] = (uint8_t)kinds++;
            }
        uint32_t bits = 1;
        while ((1u << bits) < kinds) bits++;
        if (fits && shift + (int)bits <= 32) {
            S.fb_bits = bits;
            S.bwt_code.ensure(256);
            MMT_HIP(hipMemcpyAsync(S.bwt_code.get(), code.data(), 256, hipMemcpyHostToDevice, st));
            MMT_HIP(hipStreamSynchronize(st));
        }
    }
}

// Suffix-array entries [b0, c1) of the stream -- suffix array, BWT byte and LCP value of each -- into window set `set`
// (entry b0 at index 0).  The output tiles that cover the range are run; the groups that begin in them may reach beyond
// the range on either side and are clipped (pfp_kernels.hip).
uint64_t Engine::pfp_first_tile(uint64_t b0) {
    PfpState& S = *pfp_;
    hipStream_t st = stream_;
    const uint64_t TILE = pk::emit_tile();
    if (b0 == 0) return 0;
    const uint64_t X = b0 + 1, tX = X / TILE;
    const uint32_t gA = read_u32(S.tile_first.get() + tX, st);        // first group that begins at or after tX * TILE
    uint64_t t_lo = tX;
    if (gA > 0 && (gA >= S.n_groups || S.segb.read(gA, st) > X)) t_lo = S.segb.read(gA - 1, st) / TILE;
    return t_lo;
}

void Engine::pfp_emit_window(uint64_t b0, uint64_t c1, int set) {
    PfpState& S = *pfp_;
    if (!S.emit_ready) throw std::runtime_error("the emitter's tables are gone");
    const bool W = wide_;
    hipStream_t st = stream_;
    const uint64_t TILE = pk::emit_tile();
    pk::EmitArgs ea = S.ea;
    ea.sa.lo = w_sa_[set].get(); ea.sa.hi = W ? w_hi_[set].get() : nullptr;
    ea.bwt = w_bwt_[set].get(); ea.lcp = w_lcp_[set].get();
    ea.out_base = b0; ea.win_lo = b0; ea.win_hi = c1;
    // first tile: the one in which the group that covers stream entry b0 + 1 begins (looked up ahead of the windows
    // by pfp_stream: a read here waits behind the output piece of the window before, which is on its way to the host)
    uint64_t t_lo = 0;
    const auto known = S.first_tile.find(b0);
    if (known != S.first_tile.end()) t_lo = known->second;
    else t_lo = pfp_first_tile(b0);
    const uint64_t t_hi = std::min<uint64_t>(S.tiles, c1 / TILE + 1);      // stream entry c1 (suffix-array entry c1 - 1) is the last one
    auto first_group_at = [&](uint64_t out_pos) {                         // first oversized group that begins at or after out_pos
        return (uint32_t)(std::lower_bound(S.h_fb_start.begin(), S.h_fb_start.end(), out_pos) - S.h_fb_start.begin());
    };
    const uint64_t FB_LIMIT = 0xfffffff0ull;                              // 32-bit offsets inside one launch
    uint64_t per_launch = t_hi - t_lo;
    if (const char* c = std::getenv("MMT_EMIT_TILES")) per_launch = std::max<uint64_t>(1, std::strtoull(c, nullptr, 10));
    if (per_launch > 0x7fffffffull) per_launch = 0x7fffffffull;
    for (uint64_t t0 = t_lo; t0 < t_hi;) {
        uint64_t t1 = std::min(t_hi, t0 + per_launch);
        uint32_t f0 = first_group_at(t0 * TILE), f1 = first_group_at(t1 * TILE);
        while (S.h_fb_off[f1] - S.h_fb_off[f0] > FB_LIMIT && t1 - t0 > 1) {   // too many oversized suffixes: halve the range
            t1 = t0 + (t1 - t0) / 2;
            f1 = first_group_at(t1 * TILE);
        }
        const uint64_t fb_count = S.h_fb_off[f1] - S.h_fb_off[f0];
        if (fb_count > FB_LIMIT) throw std::runtime_error("the oversized suffix groups of one emitter tile exceed 2^32 elements");
        S.xk_a.ensure((size_t)fb_count + 1); S.xv_a.ensure((size_t)fb_count + 1, W);
        if (fb_count) { S.xk_b.ensure((size_t)fb_count + 1); S.xv_b.ensure((size_t)fb_count + 1, W); }
        ea.fb_keys = S.xk_a.get(); ea.fb_vals = S.xv_a.get();
        ea.fb_base = S.h_fb_off[f0];
        S.emit_plan.ensure(pk::emit_plan_bytes(t1 - t0));
        pk::emit(ea, S.tile_first.get(), S.tile_base, S.emit_plan.get(), t0, t1, st);
        S.emit_launches++;
        const uint32_t nf = f1 - f0;
        if (nf) pk::emit_big(ea, S.fb_chunk0.get(), f0, nf, S.h_fb_chunk0[f1] - S.h_fb_chunk0[f0], st);
        if (nf) {
            // one segmented radix sort over just the oversized groups of this launch
            const uint32_t count = (uint32_t)fb_count;
            S.fb_rel.ensure((size_t)nf + 2);
            pk::relative_offsets(S.fb_off.get(), f0, nf, S.fb_rel.get(), W, st);
            if (W)
                prims::segmented_sort_pairs_u32_u64vals_ranges(d_temp_, S.xk_a.get(), S.xk_b.get(), S.xv_a.p64(), S.xv_b.p64(),
                                                               count, nf, S.fb_rel.get(), S.fb_rel.get() + 1,
                                                               S.key_shift + (int)S.fb_bits, st);
            else
                prims::segmented_sort_pairs_u32_ranges(d_temp_, S.xk_a.get(), S.xk_b.get(), S.xv_a.p32(), S.xv_b.p32(), count,
                                                       nf, S.fb_rel.get(), S.fb_rel.get() + 1, S.key_shift + (int)S.fb_bits, st);
            pk::fallback_finish(S.fb_group.get(), S.fb_off.get(), f0, f1, S.h_fb_off[f0], S.segb.get(), S.xk_b.get(),
                                S.xv_b.get(), S.fb_bits, S.decode, text_ref(), n_, ea, count, W, st);
        }
        t0 = t1;
    }
}

// The stream, window by window: emit -> scan -> verify -> (rows keep their suffix-array entries) -> next window.
void Engine::pfp_stream(ScanState& SS, const mmt_params& p) {
    PfpState& S = *pfp_;
    const uint64_t n = n_;
    hipStream_t st = stream_;
    const uint64_t ALIGN_R = 4096;
    uint64_t range = wide_ ? (1ull << 28) : std::min<uint64_t>(n, 1ull << 28);
    if (const char* c = std::getenv("MMT_SCAN_RANGE")) range = std::max<uint64_t>(1, std::strtoull(c, nullptr, 10));
    range = (std::max<uint64_t>(range, 1) + ALIGN_R - 1) / ALIGN_R * ALIGN_R;
    uint64_t lo = 0, hi = n;
    shard_range(lo, hi);
    sort_pieces_.clear();
    for (uint32_t k = 0; k < shard_count_; k++) {
        Engine* self = this;
        const uint32_t keep = shard_index_;
        self->shard_index_ = k;
        uint64_t a = 0, b = 0;
        shard_range(a, b);
        sort_pieces_.emplace_back(a, b - a);
        self->shard_index_ = keep;
    }
    const uint64_t anchor = std::min<uint64_t>(doc_len_[0], n);
    // the first emitter tile of every window, while nothing else is queued (a window that has to be repeated with a longer
    // extension looks its tile up when it gets there)
    S.first_tile.clear();
    for (uint64_t c0 = lo; c0 < hi; c0 += range) {
        const uint64_t b0 = c0 - std::min<uint64_t>(c0 ? SS.ext0 : 0, c0);
        S.first_tile[b0] = pfp_first_tile(b0);
    }
    for (uint64_t c0 = lo; c0 < hi; c0 += range) {
        const uint64_t c1 = std::min(hi, c0 + range);
        uint64_t ext = c0 ? SS.ext0 : 0;
        for (;;) {
            if (ext > c0) ext = c0;
            const uint64_t b0 = c0 - ext, len = c1 - b0;
            if (len >= 0xffffe000ull) throw std::runtime_error("scan window with its left extension exceeds 2^32 entries");
            window_reserve(0, len);
            EventPair& ee = next_range_event(SS, 3);
            ee.start(st);
            pfp_emit_window(b0, c1, 0);
            ee.stop(st);
            stream_entries_ += len;
            if (std::getenv("MMT_EMIT_ABLATE")) break;       // timing of a crippled emitter (tests/micro/emit_ablate.sh): its windows are garbage
            ColWindow w = window_view(0, b0, (uint32_t)len, (uint32_t)ext);
            if (!scan_window(SS, w, p)) { ext = std::max<uint64_t>(ext * 4, SS.ext0); continue; }   // a walk ran off the extension
            if (want_anchor_ranks_) {
                SaCol piece = w.sa; piece.lo += ext; if (piece.hi) piece.hi += ext;
                k::anchor_ranks(piece, c0, c1 - c0, anchor, wide_ ? (void*)d_rank64_.get() : (void*)d_rank_.get(), st);
            }
            keep_window(w);
            sink_flush(SS);
            break;
        }
    }
    if (read_u32(S.err.get(), st) && !std::getenv("MMT_EMIT_ABLATE")) {
        std::vector<uint32_t> er;
        d2h(er, S.err.get(), 16, st);
        auto u64 = [&](int i) { return (unsigned long long)er[i] | ((unsigned long long)er[i + 1] << 32); };
        char msg[600];
        std::snprintf(msg, sizeof(msg), "PFP order is inconsistent: %u entries (sentinel not first %u; text position past the "
                      "end: %u in tile groups [first: position %llu at stream entry %llu], %u / %u in oversized groups [first: "
                      "position %llu at entry %llu]; neighbours of a group without ascending parse ranks: %u); text %llu "
                      "characters", er[0], er[4], er[5], u64(8), u64(10), er[6], er[7], u64(12), u64(14), er[3],
                      (unsigned long long)n);
        throw std::runtime_error(msg);
    }
}

// PREFIX.dict bytes: phrases in lexicographic order, 0x01 after each, final 0x00 (newscan.hpp:386-397)
void Engine::pfp_copy_dict(std::vector<uint8_t>& out) {
    PfpState& S = *pfp_;
    if (!S.have_parse || !have_text() || !S.pstart.get())
        throw std::runtime_error("no parse available (run parse_only first)");
    const uint32_t D = S.n_distinct, nd = S.dict_len;
    DevBuf<uint32_t> which, slen, sstart;
    DevBuf<uint8_t> sorted;
    which.ensure(D); slen.ensure(D); sstart.ensure(D); sorted.ensure(nd);
    pk::invert_ranks(S.prank.get(), S.rep.get(), S.dlen.get(), D, which.get(), slen.get(), stream_);
    prims::exclusive_sum_u32(d_temp_, slen.get(), sstart.get(), D, stream_);
    pk::copy_dict(text_ref(), S.pstart.get(), S.plen.get(), which.get(), sstart.get(), D, sorted.get(), nullptr, nd,
                  false, S.pstart.wide(), stream_);
    d2h(out, sorted.get(), nd, stream_);
}

void Engine::pfp_copy_parse(std::vector<uint32_t>& out) {
    if (!pfp_->have_parse) throw std::runtime_error("no parse available");
    d2h(out, pfp_->parse.get(), pfp_->n_phrases, stream_);
}

}  // namespace mmt
