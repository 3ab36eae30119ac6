// pfp.hpp -- device state of the prefix-free-parsing producer (rows A2-A4).
#pragma once
#include <unordered_map>
#include <cstdint>
#include <vector>

#include "device_utils.hpp"
#include "parse_lcp.hpp"
#include "guided_kernels.hpp"
#include "pfp_kernels.hpp"

namespace mmt {

struct PfpState {
    uint32_t w = 0, p = 0;
    uint32_t n_cuts = 0, n_phrases = 0, n_distinct = 0, dict_len = 0, n_groups = 0;
    bool have_parse = false;
    bool guided = false;    // the dictionary stage was skipped: Engine::suffix_sort_guided sorts the text suffixes themselves
    bool expand = false;    // ... one representative per (distinct phrase, offset) only, expanded by the emitter (guided.cpp)
    int rounds_dict = 0, rounds_parse = 0;
    uint64_t run_refined = 0;         // dictionary suffixes ordered by their long run of one symbol (sorter.hpp, RunRefine)
    float ms[8] = {0};   // parse, dedup, dict build, dict SA, dict LCP + groups, parse SA, inverted lists + emitter, total
    DevBuf<uint8_t> dict, ptab, pinfo;   // pinfo: 16-byte record per phrase (k_phrase_hash)  // ptab: 16-byte record per distinct phrase (phrase_table)
    DevBuf<uint16_t> tmask;             // trigger masks, one per 16 text positions
    DevBuf<uint32_t> tcnt, toff;        // triggers per workgroup of the trigger pass and their exclusive scan
    PosBuf cuts, pstart;                // trigger positions, phrase starts (in V): text positions
    DevBuf<uint32_t> plen, iota, ord_a, order, scan, dflags, pid, rep, dlen, dstart, esuf, ephr;
    DevBuf<uint64_t> dinfo;
    DevBuf<uint8_t> ebw;
    DevBuf<uint32_t> sa_d, rank_d, gflag, pflag, gscan, pscan, prank, parse, sa_p, isa_p, err;
    DevBuf<uint64_t> h1, h2, hk_a, hk_b;
    DevBuf<uint32_t> occ_start, occ_ids, occ_ts, vflag, vscan;
    DevBuf<uint64_t> occ;               // (t << pos_bits) | V position, per phrase occurrence
    DevBuf<uint32_t> occ12;             // ... or 12-byte records (pk::occ_finish12) when the two exceed 64 bits
    uint32_t emit_pos_bits = 0, emit_w = 0;
    DevBuf<uint32_t> ce_cnt, ce_first, ce_offm1, ce_gs, sege, xk_a, xk_b, fb_group, fb_size, fb_rel, tile_first;
    DevBuf<uint8_t> emit_plan;          // one record per output tile of the launch in progress (pk::emit)
    PosBuf ce_eoff, segb, fb_off, fb_start, xv_a, xv_b;     // stream offsets / text positions
    DevBuf<uint8_t> ce_bwt, bwt_code;
    // LCP without a text-order column: LCP of adjacent parse suffixes (+ range minima), per group of equal phrase
    // suffixes the length of alpha and its LCP with the group before; ce_dpos / ce_slen: scratch of those
    ParseLcp plcp;
    DevBuf<uint32_t> ghead, ce_hl, ce_slen, occ_sl, lcp_d;   // lcp_d: LCP array of the dictionary (suffix-array order)
    DevBuf<uint64_t> segmin;                                // (flag, minimum of lcp_d since the valid entry before) per entry
    uint32_t n_entries = 0, n_fallback = 0, emit_launches = 0;
    std::unordered_map<uint64_t, uint64_t> first_tile;     // window begin (stream entry) -> first emitter tile of the window
    bool bwt_ready = false;
    // the emitter between its windows (Engine::pfp_emit_window): arguments shared by every launch, the oversized groups'
    // offsets / begin positions on the host, the BWT code of their sort keys
    bool emit_ready = false;
    pk::EmitArgs ea;
    pk::BwtDecode decode{};
    uint32_t fb_bits = 0;
    int key_shift = 0;
    uint64_t tiles = 0, tile_base = 0;  // output tiles [tile_base, tiles) of the tables at hand (tile_first begins at tile_base)
    std::vector<uint64_t> h_fb_off, h_fb_start, h_fb_chunk0;
    DevBuf<uint64_t> fb_chunk0;
    // the bucket-wise producer between its batches (guided.cpp): rank / successor tables over the phrase ends, the
    // histogram of the suffixes' leading characters
    gk::Ctx gctx{};
    DevBuf<uint32_t> g_rdir, g_brank;
    DevBuf<uint16_t> g_coff;               // the phrase ends as a list (gk::Ctx::coff): packed texts
    DevBuf<uint64_t> g_nxt;
    std::vector<uint64_t> g_bins, g_bins_rep;     // text suffixes per bin of leading characters; of them representatives (expansion)
    DevBuf<uint32_t> g_repbits;                   // bit k: phrase k is the representative occurrence of its distinct phrase
    int g_prefix = 0;
    uint32_t g_nbins = 0;
    // the bins the last stream of this producer took (a rank's share: Engine::set_scan_shard), and the symbol codes they are made of
    uint32_t g_share_lo = 0, g_share_hi = 0;
    int g_bits = 0;
    uint8_t g_code[256] = {0};
    bool g_share_valid = false;
    // giant phrases of the bucket-wise producer (guided.cpp::build_giant; gk::Ctx::g_*)
    DevBuf<uint32_t> gi_k, gi_base, gi_isa, gi_grp, gi_lcp, gi_bmin, gi_bits, gi_rank;
    DevBuf<uint64_t> gi_ps;
    uint32_t gi_occ = 0, gi_distinct = 0, gi_chars = 0, gi_nb = 0, gi_levels = 0;
};

}  // namespace mmt
