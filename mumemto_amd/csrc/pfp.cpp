// pfp.cpp -- Engine methods of the PFP producer: parse (A2), dictionary / parse
// structures (A3) and the suffix array of the text from them (A4).
#include <algorithm>
#include <chrono>
#include <stdexcept>

#include "engine.hpp"
#include "pfp_kernels.hpp"
#include "prims.hpp"

namespace mmt {

static int bit_width_u64(uint64_t v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

static uint32_t read_u32(const uint32_t* d, hipStream_t s) {
    uint32_t v = 0;
    MMT_HIP(hipMemcpyAsync(&v, d, 4, hipMemcpyDeviceToHost, s));
    MMT_HIP(hipStreamSynchronize(s));
    return v;
}

// A2 + the dictionary half of A3: phrases, distinct phrases, dictionary text with its suffix
// array / LCP, phrase ranks, parse.  Requires build_text() to have run.
void Engine::pfp_parse(uint32_t w, uint32_t p) {
    PfpState& S = *pfp_;
    const uint32_t n = (uint32_t)n_;
    if (w < 1 || w > 32 || p < 1) throw std::runtime_error("PFP window must be in [1, 32] and the modulus positive");
    std::vector<uint32_t> hist;
    d2h(hist, d_hist_.get(), 256, stream_);
    if (hist[0] || hist[1] || hist[2])          // newscan.hpp:318: characters <= Dollar are not allowed
        throw std::runtime_error("input contains bytes <= 0x02, which the prefix-free parse reserves");
    S.w = w; S.p = p; S.have_parse = false; S.bwt_ready = false;
    hipStream_t st = stream_;
    EventPair e0, e1, e2, e3, e4;

    // -- triggers, phrase boundaries
    e0.start(st);
    const uint32_t vlen = n + 1 + w;
    S.vtext.ensure((size_t)vlen + 64);
    pk::make_vtext(d_text_.get(), n, w, S.vtext.get(), vlen + 64, st);
    const uint32_t tb = pk::trigger_blocks(n);
    S.tmask.ensure(((size_t)n + 15) / 16 + 1); S.tcnt.ensure((size_t)tb + 1); S.toff.ensure((size_t)tb + 1);
    pk::trigger_masks(d_text_.get(), n, w, p, S.tmask.get(), S.tcnt.get(), st);
    prims::exclusive_sum_u32(d_temp_, S.tcnt.get(), S.toff.get(), tb, st);
    S.err.ensure(4);
    S.n_cuts = read_u32(S.toff.get() + (tb - 1), st) + read_u32(S.tcnt.get() + (tb - 1), st);   // size the cut list exactly
    S.cuts.ensure((size_t)S.n_cuts + 1);
    pk::trigger_cuts(S.tmask.get(), n, S.toff.get(), S.cuts.get(), st);
    const uint32_t m = S.n_phrases = S.n_cuts + 1;
    S.pstart.ensure(m); S.plen.ensure(m);
    pk::phrase_bounds(S.cuts.get(), S.n_cuts, n, w, S.pstart.get(), S.plen.get(), st);
    e0.stop(st);

    // -- distinct phrases: fingerprints, sort, verified grouping
    e1.start(st);
    S.h1.ensure(m); S.pinfo.ensure((size_t)m * 16 + 16); S.hk_a.ensure(m); S.hk_b.ensure(m);
    S.iota.ensure(m); S.ord_a.ensure(m); S.order.ensure(m);
    pk::phrase_hash(S.vtext.get(), S.pstart.get(), S.plen.get(), m, S.h1.get(), S.pinfo.get(), st);
    pk::iota(S.iota.get(), m, st);
    S.dflags.ensure(m); S.scan.ensure(m);
    for (int attempt = std::getenv("MMT_PFP_TWO_FINGERPRINTS") ? 1 : 0;; attempt++) {     // the variable forces the rare path (tests)
        if (attempt == 0) {
            // order by the first fingerprint alone; equal phrases are adjacent unless two different phrases share it
            prims::sort_pairs_u64_u32(d_temp_, S.h1.get(), S.hk_b.get(), S.iota.get(), S.order.get(), m, 0, 64, st);
        } else {
            // (rare) order by both fingerprints: stable sort by the second, then by the first
            S.h2.ensure(m);
            pk::second_fingerprint(S.pinfo.get(), m, S.h2.get(), st);
            prims::sort_pairs_u64_u32(d_temp_, S.h2.get(), S.hk_a.get(), S.iota.get(), S.ord_a.get(), m, 0, 64, st);
            pk::gather_u64(S.h1.get(), S.ord_a.get(), m, S.hk_a.get(), st);
            prims::sort_pairs_u64_u32(d_temp_, S.hk_a.get(), S.hk_b.get(), S.ord_a.get(), S.order.get(), m, 0, 64, st);
        }
        MMT_HIP(hipMemsetAsync(S.err.get(), 0, 16, st));
        pk::mark_distinct(S.order.get(), S.hk_b.get(), S.pinfo.get(), S.vtext.get(), m, S.dflags.get(), S.err.get(), st);
        prims::inclusive_sum_u32(d_temp_, S.dflags.get(), S.scan.get(), m, st);
        uint32_t flags2[2] = {0, 0};
        MMT_HIP(hipMemcpyAsync(flags2, S.err.get(), 8, hipMemcpyDeviceToHost, st));
        MMT_HIP(hipStreamSynchronize(st));
        if (flags2[0])
            throw std::runtime_error("phrase fingerprint collision (128-bit); refusing to merge different phrases");
        if (attempt == 0 && flags2[1]) continue;
        break;
    }
    const uint32_t D = S.n_distinct = read_u32(S.scan.get() + (m - 1), st);
    S.pid.ensure(m); S.rep.ensure(D); S.dlen.ensure(D); S.dstart.ensure(D);
    pk::assign_distinct(S.order.get(), S.scan.get(), S.dflags.get(), S.plen.get(), m, S.pid.get(), S.rep.get(),
                        S.dlen.get(), st);
    e1.stop(st);

    // -- dictionary text (distinct phrases in fingerprint order; ranks come from its suffix array)
    e2.start(st);
    prims::exclusive_sum_u32(d_temp_, S.dlen.get(), S.dstart.get(), D, st);
    const uint64_t dict_len64 = (uint64_t)read_u32(S.dstart.get() + (D - 1), st) + read_u32(S.dlen.get() + (D - 1), st) + 1;
    if (dict_len64 >= 0xffffff00ull) throw std::runtime_error("PFP dictionary exceeds the 32-bit build");
    const uint32_t nd = S.dict_len = (uint32_t)dict_len64;
    S.dict.ensure((size_t)nd + 64); S.dinfo.ensure(nd);
    MMT_HIP(hipMemsetAsync(S.dict.get() + nd, 0, 64, st));
    // room for the byte before each position in the phrase-id word (MMT_PFP_NO_PACK: the other path, for tests)
    const bool pack_prev = D < (1u << 24) && !std::getenv("MMT_PFP_NO_PACK");
    pk::copy_dict(S.vtext.get(), S.pstart.get(), S.plen.get(), S.rep.get(), S.dstart.get(), D, S.dict.get(),
                  S.dinfo.get(), nd, pack_prev, st);
    e2.stop(st);

    // -- suffix array of the dictionary (dictionary.hpp:133) ...
    e3.start(st);
    uint8_t code[256];
    int sigma = 0;
    for (int c = 0; c < 256; c++) code[c] = (hist[c] || c <= 2) ? (uint8_t)(++sigma) : 0;
    const int bits = std::max(1, bit_width_u64((uint64_t)sigma));
    // every 0x01 (end of a phrase) is a unique terminator, ordered by position: what follows it never matters, so
    // a suffix is final as soon as the compared prefix reaches the end of its phrase (one key bit says so)
    const int chars = std::min(63 / bits, 63);
    d_code_.ensure(256);
    MMT_HIP(hipMemcpyAsync(d_code_.get(), code, 256, hipMemcpyHostToDevice, st));
    S.sa_d.ensure(nd); S.rank_d.ensure(nd);
    sorter_.reserve(std::max(nd, m));
    k::pack_keys(S.dict.get(), nd, d_code_.get(), bits, chars, (uint32_t)code[1], sorter_.keys_in(), sorter_.vals_in(), st);
    S.rounds_dict = sorter_.sort(nd, bits * chars + 1, (uint64_t)chars, S.sa_d.get(), S.rank_d.get(), d_temp_, st, true);
    e3.stop(st);
    // ... the groups of equal proper phrase suffixes and the phrase ranks
    e4.start(st);
    S.esuf.ensure(nd); S.ephr.ensure(nd); S.ebw.ensure(nd);
    pk::entry_info(S.sa_d.get(), S.dinfo.get(), S.dict.get(), nd, pack_prev, S.esuf.get(), S.ephr.get(), S.ebw.get(), st);
    S.gflag.ensure(nd); S.pflag.ensure(nd); S.vflag.ensure(nd); S.gscan.ensure(nd); S.pscan.ensure(nd);
    S.prank.ensure(D); S.parse.ensure(m);
    pk::group_flags(S.esuf.get(), S.sa_d.get(), S.dict.get(), nd, w, S.gflag.get(), S.pflag.get(), S.vflag.get(), st);
    prims::inclusive_sum_u32(d_temp_, S.gflag.get(), S.gscan.get(), nd, st);
    prims::inclusive_sum_u32(d_temp_, S.pflag.get(), S.pscan.get(), nd, st);
    pk::phrase_ranks(S.esuf.get(), S.ephr.get(), S.pscan.get(), nd, S.prank.get(), st);
    pk::parse_ranks(S.pid.get(), S.prank.get(), m, S.parse.get(), st);
    S.n_groups = read_u32(S.gscan.get() + (nd - 1), st);
    e4.stop(st);
    S.ms[0] = e0.ms(); S.ms[1] = e1.ms(); S.ms[2] = e2.ms(); S.ms[3] = e3.ms(); S.ms[4] = e4.ms();
    S.have_parse = true;
}

// A3 (parse half) + A4: suffix array of the text = positions sorted by
// (group of the phrase suffix, rank of the following parse suffix).
void Engine::suffix_sort_pfp(uint32_t w, uint32_t p) {
    PfpState& S = *pfp_;
    const uint32_t n = (uint32_t)n_;
    auto t0 = std::chrono::steady_clock::now();
    pfp_parse(w, p);
    hipStream_t st = stream_;
    const uint32_t m = S.n_phrases, D = S.n_distinct;
    EventPair e5, e6;
    e5.start(st);
    S.sa_p.ensure(m); S.isa_p.ensure(m);
    const int pbits = std::max(1, bit_width_u64((uint64_t)D));
    const int pchars = std::max(1, 64 / pbits);
    pk::pack_keys_u32(S.parse.get(), m, pbits, pchars, sorter_.keys_in(), sorter_.vals_in(), st);
    S.rounds_parse = sorter_.sort(m, pbits * pchars, (uint64_t)pchars, S.sa_p.get(), S.isa_p.get(), d_temp_, st);
    if (lean_) { MMT_HIP(hipStreamSynchronize(st)); sorter_.release(); }   // dictionary-sized doubling scratch: done
    e5.stop(st);

    e6.start(st);
    const int shift = bit_width_u64((uint64_t)m + 1);
    const uint32_t nd = S.dict_len;
    d_sa_.ensure(n); d_bwt_.ensure((size_t)n + 16);     // no inverse suffix array on this path (see Engine::lcp_bwt)
    // inverted lists: parse positions ordered by (phrase, rank of the following parse suffix)
    S.occ_start.ensure((size_t)D + 2); S.occ_pos.ensure((size_t)m * 2);       // (t, position) records
    S.occ_ids.ensure((size_t)m + 1); S.occ_ts.ensure((size_t)m + 1);
    {
        sorter_.u32_a().ensure((size_t)m + 1); sorter_.u32_b().ensure((size_t)m + 1);     // scratch
        uint32_t* k_in = sorter_.u32_a().get();
        uint32_t* v_in = sorter_.u32_b().get();
        pk::occ_sequence(S.sa_p.get(), S.pid.get(), m, D, k_in, v_in, st);
        prims::sort_pairs_u32_u32(d_temp_, k_in, S.occ_ids.get(), v_in, S.occ_ts.get(), (size_t)m + 1, 0,
                                  std::max(1, bit_width_u64((uint64_t)D)), st);
        pk::occ_finish(S.occ_ids.get(), S.occ_ts.get(), S.sa_p.get(), S.pstart.get(), m, S.occ_start.get(),
                       S.occ_pos.get(), st);
    }
    // valid dictionary suffixes in dictionary suffix-array order, compacted ("entries")
    S.vscan.ensure(nd); S.ptab.ensure((size_t)D * 16 + 16);
    prims::exclusive_sum_u32(d_temp_, S.vflag.get(), S.vscan.get(), nd, st);
    const uint32_t E = S.n_entries = read_u32(S.vscan.get() + (nd - 1), st) + read_u32(S.vflag.get() + (nd - 1), st);
    pk::phrase_table(S.occ_start.get(), S.plen.get(), S.rep.get(), D, S.ptab.get(), st);
    S.ce_cnt.ensure(E); S.ce_eoff.ensure(E); S.ce_first.ensure(E); S.ce_offm1.ensure(E); S.ce_gs.ensure(E);
    S.ce_bwt.ensure(E);
    pk::entry_compact(S.esuf.get(), S.ephr.get(), S.ebw.get(), S.gflag.get(), S.vflag.get(), S.vscan.get(),
                      S.ptab.get(), nd, S.ce_cnt.get(), S.ce_first.get(), S.ce_offm1.get(), S.ce_bwt.get(),
                      S.ce_gs.get(), st);
    prims::exclusive_sum_u32(d_temp_, S.ce_cnt.get(), S.ce_eoff.get(), E, st);
    {
        const uint64_t total = (uint64_t)read_u32(S.ce_eoff.get() + (E - 1), st) + read_u32(S.ce_cnt.get() + (E - 1), st);
        if (total != (uint64_t)n + 1) throw std::runtime_error("PFP expansion does not cover the text exactly once");
    }
    // groups of equal phrase suffixes: first entry and first output position of each
    const uint32_t G = S.n_groups;
    S.sege.ensure((size_t)G + 2); S.segb.ensure((size_t)G + 2);
    prims::select_indices_u32flags(d_temp_, S.ce_gs.get(), S.sege.get(), S.err.get(), E, st);
    if (read_u32(S.err.get(), st) != G) throw std::runtime_error("PFP group count mismatch");
    k::gather_u32_idx32(S.ce_eoff.get(), S.sege.get(), G, S.segb.get(), st);
    {
        const uint32_t endv[2] = {E, n + 1};
        MMT_HIP(hipMemcpyAsync(S.sege.get() + G, &endv[0], 4, hipMemcpyHostToDevice, st));
        MMT_HIP(hipMemcpyAsync(S.segb.get() + G, &endv[1], 4, hipMemcpyHostToDevice, st));
        MMT_HIP(hipStreamSynchronize(st));
    }
    // groups larger than one LDS tile of the emitter get compact slots in the fallback arrays
    uint32_t* osize = S.gscan.get();                       // scratch of >= G entries (G <= dictionary length)
    pk::oversize(S.segb.get(), G, osize, st);
    S.fb_group.ensure((size_t)G + 1);
    prims::select_indices_u32flags(d_temp_, osize, S.fb_group.get(), S.err.get(), G, st);
    const uint32_t F = S.n_fallback = read_u32(S.err.get(), st);
    S.fb_size.ensure((size_t)F + 2); S.fb_off.ensure((size_t)F + 2);
    uint32_t fb_total = 0;
    if (F) {
        k::gather_u32_idx32(osize, S.fb_group.get(), F, S.fb_size.get(), st);
        MMT_HIP(hipMemsetAsync(S.fb_size.get() + F, 0, 4, st));
        prims::exclusive_sum_u32(d_temp_, S.fb_size.get(), S.fb_off.get(), (size_t)F + 1, st);
        fb_total = read_u32(S.fb_off.get() + F, st);
    }
    S.xk_a.ensure((size_t)fb_total + 1); S.xv_a.ensure((size_t)fb_total + 1);
    // the emitter
    MMT_HIP(hipMemsetAsync(S.err.get(), 0, 16, st));
    pk::EmitArgs ea;
    ea.segb = S.segb.get(); ea.sege = S.sege.get(); ea.n_groups = G;
    ea.ce_eoff = S.ce_eoff.get(); ea.ce_cnt = S.ce_cnt.get(); ea.ce_first = S.ce_first.get();
    ea.ce_offm1 = S.ce_offm1.get(); ea.ce_bwt = S.ce_bwt.get(); ea.ce_gs = S.ce_gs.get();
    ea.occ = reinterpret_cast<const uint2*>(S.occ_pos.get());
    ea.n = n; ea.sa = d_sa_.get(); ea.bwt = d_bwt_.get();
    ea.fb_group = S.fb_group.get(); ea.fb_off = S.fb_off.get(); ea.n_fb = F;
    ea.fb_keys = S.xk_a.get(); ea.fb_vals = S.xv_a.get(); ea.err = S.err.get();
    // the byte before every suffix of an oversized group rides in the low bits of its sort key when the text has at
    // most 16 different bytes and the parse rank leaves room (MMT_PFP_NO_BWT_CODE: read it from the text instead)
    pk::BwtDecode decode{};
    uint32_t fb_bits = 0;
    if (F && !std::getenv("MMT_PFP_NO_BWT_CODE")) {
        std::vector<uint32_t> hist;
        d2h(hist, d_hist_.get(), 256, st);
        std::vector<uint8_t> code(256, 0);
        uint32_t kinds = 0;
        bool fits = true;
        for (int b = 0; b < 256; b++)
            if (b == 0 || hist[b]) {                              // 0 stands before the first text position
                if (kinds == 16) { fits = false; break; }
                decode.byte[kinds] = (uint8_t)b; code[b] = (uint8_t)kinds++;
            }
        uint32_t bits = 1;
        while ((1u << bits) < kinds) bits++;
        if (fits && shift + (int)bits <= 32) {
            fb_bits = bits;
            S.bwt_code.ensure(256);
            MMT_HIP(hipMemcpyAsync(S.bwt_code.get(), code.data(), 256, hipMemcpyHostToDevice, st));
            MMT_HIP(hipStreamSynchronize(st));
        }
    }
    ea.bwt_code = S.bwt_code.get(); ea.fb_bits = fb_bits;
    S.tile_first.ensure(((size_t)n + 1) / pk::EMIT_TILE + 4);
    pk::emit(ea, n + 1, S.tile_first.get(), st);
    if (F) {      // one segmented radix sort over just the oversized groups
        S.xk_b.ensure((size_t)fb_total + 1); S.xv_b.ensure((size_t)fb_total + 1);
        prims::segmented_sort_pairs_u32_ranges(d_temp_, S.xk_a.get(), S.xk_b.get(), S.xv_a.get(), S.xv_b.get(),
                                               fb_total, F, S.fb_off.get(), S.fb_off.get() + 1, shift + (int)fb_bits, st);
        pk::fallback_finish(S.fb_group.get(), S.fb_off.get(), F, S.segb.get(), S.xk_b.get(), S.xv_b.get(), fb_bits, decode,
                            d_text_.get(), n,
                            d_sa_.get(), d_bwt_.get(), S.err.get(), st);
    }
    if (read_u32(S.err.get(), st)) throw std::runtime_error("PFP order: the end sentinel is not first");
    S.bwt_ready = true;
    e6.stop(st);
    S.ms[5] = e5.ms(); S.ms[6] = e6.ms();
    S.ms[7] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    sort_rounds_ = S.rounds_dict;
}

// PREFIX.dict bytes: phrases in lexicographic order, 0x01 after each, final 0x00 (newscan.hpp:386-397)
void Engine::pfp_copy_dict(std::vector<uint8_t>& out) {
    PfpState& S = *pfp_;
    if (!S.have_parse) throw std::runtime_error("no parse available");
    const uint32_t D = S.n_distinct, nd = S.dict_len;
    DevBuf<uint32_t> which, slen, sstart;
    DevBuf<uint8_t> sorted;
    which.ensure(D); slen.ensure(D); sstart.ensure(D); sorted.ensure(nd);
    pk::invert_ranks(S.prank.get(), S.rep.get(), S.dlen.get(), D, which.get(), slen.get(), stream_);
    prims::exclusive_sum_u32(d_temp_, slen.get(), sstart.get(), D, stream_);
    pk::copy_dict(S.vtext.get(), S.pstart.get(), S.plen.get(), which.get(), sstart.get(), D, sorted.get(), nullptr, nd,
                  false, stream_);
    d2h(out, sorted.get(), nd, stream_);
}

void Engine::pfp_copy_parse(std::vector<uint32_t>& out) {
    if (!pfp_->have_parse) throw std::runtime_error("no parse available");
    d2h(out, pfp_->parse.get(), pfp_->n_phrases, stream_);
}

}  // namespace mmt
