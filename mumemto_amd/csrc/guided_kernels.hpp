// guided_kernels.hpp -- launch wrappers of guided_kernels.hip: the parse-guided suffix sort (guided.cpp).
#pragma once
#include <cstdint>

#include <hip/hip_runtime_api.h>

#include "parse_lcp.hpp"
#include "textref.hpp"
#include "wide.hpp"

namespace mmt { namespace gk {

// One element of a batch: V index of the string's first character (40 bits) and, above it, the number of characters
// up to the end of its phrase, saturated (an element with a saturated length looks the end up again).
constexpr uint64_t POS_MASK = (1ull << 40) - 1;
constexpr uint32_t LEN_SAT = 0xffffffu;
// keys of a refinement round: up to 63 bits of symbol codes, or bit 63 + the rank of the parse suffix that follows
constexpr uint64_t RANK_KEY = 1ull << 63;
constexpr uint64_t GIANT_KEY = 1ull << 62;  // ... or bit 62 + the group of the rest of alpha in the giant dictionary
constexpr uint32_t TILE = 4096;            // text positions per workgroup of the text-order kernels
constexpr uint32_t SORT_CAP = 2048;        // elements of one LDS tile of the round sort
constexpr uint32_t NO_BOUND = 0xffffffffu;

struct Ctx {
    TextRef T;                 // V = Dollar . T . Dollar^w (pfp_kernels.hip) behind the accessor of textref.hpp: bytes, or two
                               // bits per character + exceptions; V index q = text position + 1
    uint64_t n;                // text length
    uint32_t w;                // trigger window
    const uint64_t* mask;      // bit c set <=> a phrase ends at text position c (64 positions per word, zero padded)
    const uint32_t* rdir;      // number of phrase ends before text position 512 * j
    const uint64_t* nxt;       // first phrase end at or after text position 4096 * j (n + w - 1: none)
    const uint32_t* isa_p;     // rank of the parse suffix that starts at phrase k
    uint32_t m;                // number of phrases
    const uint8_t* code;       // byte -> symbol code (1 = Dollar, 0 = past the padding)
    int bits, chars;           // bits per code, characters per 63-bit key
    uint32_t skip;             // 0: elements are text suffixes, 1: elements are whole phrases
    uint32_t pos_bits;         // element records: position in the low pos_bits bits, above it ...
    uint32_t rec_rank;         // ... 1: the rank of the following parse suffix, 0: the length of alpha (pos_bits = 40)
    uint32_t tile0 = 0;        // first tile of this launch (the text-order kernels run in slices of 2^23 tiles)
    // symbol codes of A C G T (0: the text holds none of that base) -- the text-order kernels read the tiles of a PACKED text that
    // hold no exception straight from the words of 2-bit codes (guided_kernels.hip for_tile_dense); dense_ok: all four are set
    uint8_t acgt[4] = {0, 0, 0, 0}; uint8_t dense_ok = 0;
    uint32_t acgt_lut = 0;     // the four codes in one word, code of base d in bits 8 d .. 8 d + 7 (a lookup by a computed index)
    const uint32_t* pid = nullptr;   // text suffixes: id of the distinct phrase at parse position k (m entries), or nullptr
    // the phrase ends as a list (mask = rdir = nullptr then): coff[k] = offset of phrase end k inside its block of 4096 text
    // positions, brank[b] = phrase ends before block b (blocks + 2 entries); nxt as above
    const uint16_t* coff = nullptr; const uint32_t* brank = nullptr;
    // Giant phrases (longer than g_depth characters: a run of N, a microsatellite -- no trigger of the parse falls inside a
    // periodic run, newscan.hpp:265-325): their suffixes are sorted once, as a small dictionary of their own, and a
    // comparison that is still undecided g_depth characters into alpha continues on those ranks instead of on characters.
    //   g_k[j] (ascending) = phrase index of the j-th giant occurrence, g_ps[j] = its first V index, g_base[j] = position of
    //   its phrase in the giant dictionary; g_isa[position] = entry of the giant dictionary's suffix array; g_grp[entry] =
    //   id of its group of equal strings; g_rmq over the giant dictionary's LCP array.
    const uint32_t* g_k = nullptr; const uint64_t* g_ps = nullptr; const uint32_t* g_base = nullptr; uint32_t g_n = 0;
    const uint32_t* g_isa = nullptr; const uint32_t* g_grp = nullptr; RmqView g_rmq; uint32_t g_depth = 0;
    const uint32_t* g_bits = nullptr;                        // bit k: phrase k of the parse is a giant occurrence
    const uint32_t* g_rank = nullptr;                        // giant occurrences before phrase 32 j (one entry per word of g_bits)
    // Expansion (guided.cpp): bit k <=> phrase k is the representative occurrence of its distinct phrase; the text-order
    // kernels (bin_hist, batch_count, batch_fill) then see only the suffixes that start in such an occurrence.  nullptr: all.
    const uint32_t* repbits = nullptr;
    // ... and `expand` says that the elements are such representatives: no two of them spell the same (distinct phrase, offset),
    // their order inside a group of equal phrase suffixes does not matter (the emitter merges the groups' inverted lists by
    // parse rank), so ties are broken by position -- no rank of a following parse suffix is ever looked up -- and the LCP
    // of two members of a group is reported as |alpha| (all that the group tables ask: "at least |alpha|").
    uint32_t expand = 0;
    // MMT_GUIDED_PROF: 16 counters of k_resolve_medium (clock ticks per phase summed over the waves, waves, walks); else nullptr
    unsigned long long* prof = nullptr;
};

// A slice of a bin of ONE repeated symbol (guided_kernels.hip, "slices of a bin of one repeated symbol"): the suffixes c^r X of the
// bin whose K = (X0 < c ? r : 2^41 - r) falls into the buckets [blo, bhi) -- a contiguous piece of the suffix array.  sym = 0: none.
constexpr uint32_t RUN_BUCKETS_HALF = 8448;          // buckets per class: r exact below 256, then 256 per power of two up to 2^40
struct RunSlice {
    uint32_t sym = 0, blo = 0, bhi = 0;
    const uint64_t* lead = nullptr;                  // per tile of 4096 text positions: length of the run of `first` that begins at its
    const uint8_t* first = nullptr;                  // first position (may span many tiles), the symbol there, and the symbol behind the run
    const uint8_t* follow = nullptr;
    uint64_t n_tiles = 0;
};
// per tile: symbol at the first position, number of leading positions that hold it (4096: the whole tile), the symbol behind them
void tile_lead(const Ctx& c, uint8_t* first, uint16_t* lead, uint8_t* follow, hipStream_t s);
// hist[b], b < 2 * RUN_BUCKETS_HALF: suffixes of the run bin per bucket; hist[2 * RUN_BUCKETS_HALF + b]: those c.repbits keeps
void run_hist(const Ctx& c, int prefix_chars, uint32_t bin, const RunSlice& rs, uint64_t* hist, hipStream_t s);

// cut bits -> rank directory counts (one per 512 positions) and the first cut of every block of 4096 positions
void rank_counts(const uint64_t* mask, uint64_t n_words, uint32_t* counts, uint64_t n_counts, hipStream_t s);
void block_first_cut(const uint64_t* mask, uint64_t n_words, uint64_t* first, uint64_t n_blocks, hipStream_t s);
void block_cut_counts(const uint64_t* mask, uint64_t n_words, uint32_t* counts, uint64_t n_blocks, hipStream_t s);
void block_cut_offsets(const uint64_t* mask, uint64_t n_words, const uint32_t* brank, uint16_t* coff, uint64_t n_blocks, hipStream_t s);

// bins = leading `prefix_chars` symbols of every text suffix (at most 4096 bins)
void bin_hist(const Ctx& c, int prefix_chars, uint64_t* hist, hipStream_t s);
void batch_count(const Ctx& c, int prefix_chars, uint32_t bin_lo, uint32_t bin_hi, uint32_t* tile_count, hipStream_t s,
                 const RunSlice& rs = RunSlice());
// the suffixes whose bin lies in [bin_lo, bin_hi), in text order: first key and element record
// (next_count, optional: per-tile counts of the next batch's bins [next_lo, next_hi), taken along)
void batch_fill(const Ctx& c, int prefix_chars, uint32_t bin_lo, uint32_t bin_hi, const uint32_t* tile_off, uint64_t* keys,
                uint64_t* pos, uint32_t next_lo, uint32_t next_hi, uint32_t* next_count, hipStream_t s, const RunSlice& rs = RunSlice());
// several batches per pass over the text: the suffixes of bins [bin_lo, bin_hi) in text order as (offset in the tile | bin << 12):
// four bytes an entry; its tile is where it lies in the list (tile_off, the prefix sums of the pass's per-tile counts)
// (next_count, optional: the per-tile counts of the next pass's bins [next_lo, next_hi), taken along)
void stage_fill(const Ctx& c, int prefix_chars, uint32_t bin_lo, uint32_t bin_hi, const uint32_t* tile_off, uint32_t* staged,
                uint32_t next_lo, uint32_t next_hi, uint32_t* next_count, hipStream_t s);
// blk_tile[B] = the tile of the list's entry 4096 B (one more entry behind the last block: the last tile)
void stage_block_tiles(const uint32_t* tile_off, uint32_t n_tiles, uint64_t n, uint32_t* blk_tile, hipStream_t s);
// ... and what a batch -- bins [b0, b1) -- takes from that list: counts per block of 4096 entries, then keys and records
void stage_count(const uint32_t* staged, uint64_t n, uint32_t b0, uint32_t b1, uint32_t* block_count, hipStream_t s);
// (next_count, optional: per-block counts of the next batch's bins [nb0, nb1), taken along)
void stage_take(const Ctx& c, const uint32_t* staged, uint64_t n, uint32_t b0, uint32_t b1, const uint32_t* block_off, uint64_t* keys,
                uint64_t* pos, uint32_t nb0, uint32_t nb1, uint32_t* next_count, const uint32_t* tile_off, const uint32_t* blk_tile,
                hipStream_t s);
// one element per distinct phrase (c.skip = 1)
void phrase_items(const Ctx& c, const void* pstart, bool wide, const uint32_t* rep, uint32_t D, uint64_t* keys,
                  uint64_t* pos, hipStream_t s);

// after the first sort: head[j] = 1 where a new key starts, active[j] = 1 inside groups of two or more
// lcp (optional, B entries): the LCP of a slot with the slot before it where their first keys differ
void heads0(const uint64_t* keys, uint32_t B, uint8_t* head, uint8_t* active, uint32_t* lcp, int bits, int chars, hipStream_t s);
void gather_active(const uint32_t* idx, uint32_t m, const uint64_t* pos_sorted, const uint8_t* head, uint32_t* slot,
                   uint64_t* pos, uint32_t* headval, hipStream_t s);
void round_keys(const Ctx& c, const uint64_t* pos, uint32_t m, uint64_t offset, uint64_t* keys, uint32_t* err, hipStream_t s,
                const uint32_t* gr = nullptr, const uint8_t* pure = nullptr, const uint32_t* ghead = nullptr);
// expansion: gr[e] = the entry of the giant dictionary's suffix array for member e at `offset` (0xffffffff: none), pure[first member
// of a group] = 1 when every member has one -- round_keys then hands such groups their entries as keys (m bytes of `pure` are set)
void giant_probe(const Ctx& c, const uint64_t* pos, const uint32_t* ghead, uint32_t m, uint64_t offset, uint32_t* gr, uint8_t* pure,
                 hipStream_t s);
// groups of at most 8 elements, finished by direct comparison (their sorted records go to `out` at the group's slots)
void resolve_small(const Ctx& c, const uint64_t* pos, const uint32_t* ghead, const uint32_t* slot, uint32_t m, uint64_t offset,
                   uint64_t* out, uint8_t* flags, uint32_t* err, hipStream_t s,
                   uint32_t* lcp_out = nullptr, uint32_t limit = 0);   // (lcp_out, optional, by output slot: the LCP of a member with the
                   // member before it; limit: the largest group finished here, 0 = the eight of the first step)
// groups of 9 .. 128 elements, in four size classes (<= 16, 32, 64, 128): list = 4 regions of `cap` (first element, size)
// pairs (uint2), count4 = 4 counters; then 4 / 2 / 1 / 1 groups per wave; flags must be preset to 1 (elements of larger
// groups keep it); lcp_out (optional, indexed by slot): the LCP of every member but the first with the member before it
void medium_groups(const uint32_t* ghead, uint32_t m, void* list, uint32_t cap, uint32_t* count4, hipStream_t s);
void resolve_medium(const Ctx& c, const RmqView& R, const uint64_t* pos, const uint32_t* slot, const void* list, uint32_t cap,
                    const uint32_t n_groups[4], uint64_t offset, uint64_t* out, uint8_t* flags, uint32_t* lcp_out, uint32_t* err,
                    hipStream_t s);
void tile_bounds(const uint32_t* ghead, uint32_t m, uint32_t target, uint32_t limit, uint32_t n_tiles, uint32_t* bound,
                 hipStream_t s);
// sorts every tile that fits LDS by (group, key); lists the others (big_*) and copies them through unsorted
void local_sort(const uint64_t* kin, const uint64_t* pin, const uint32_t* ghead, uint64_t* kout, uint64_t* pout,
                const uint32_t* bound, uint32_t n_tiles, uint32_t* big_begin, uint32_t* big_end, uint32_t* big_count,
                uint32_t big_cap, hipStream_t s);
// the groups (begin, end) inside the listed ranges, for the segmented sort of groups longer than a tile
// (lcp_out, optional, indexed by batch slot `slot[c]`: the LCP of two neighbours that part by the characters of this round's keys)
void round_heads(const Ctx& ctx, const uint64_t* keys, const uint32_t* ghead, uint32_t m, uint32_t* headval, uint32_t* err, hipStream_t s,
                 uint32_t* lcp_out = nullptr, const uint32_t* slot = nullptr, uint64_t offset = 0, const uint8_t* pure = nullptr);
void round_apply(const uint64_t* pos_sorted, const uint32_t* newhead, const uint32_t* slot, uint32_t m, uint64_t* out,
                 uint8_t* flags, hipStream_t s);
void round_compact(const uint32_t* idx, uint32_t m2, const uint32_t* slot, const uint64_t* pos_sorted,
                   const uint32_t* newhead, uint32_t* slot_out, uint64_t* pos_out, uint32_t* headval_out, hipStream_t s);
// groups of a listed range: segment list for the segmented sort
void range_groups(const uint32_t* ghead, const uint32_t* big_begin, const uint32_t* big_end, uint32_t big, uint32_t* seg_begin,
                  uint32_t* seg_end, uint32_t* seg_count, hipStream_t s);

// giant phrases: flags[i] = v[i] > thr; the giant dictionary's bookkeeping (guided.cpp::build_giant)
void flag_greater(const uint32_t* v, uint32_t n, uint32_t thr, uint32_t* flags, hipStream_t s);
void flag_spread(const uint32_t* in, uint32_t n, uint32_t* out, hipStream_t s);                       // out[k] = in[k - 1] | in[k] | in[k + 1]
void flag_to_distinct(const uint32_t* occ_flag, const uint32_t* pid, uint32_t m, uint32_t* dflag, hipStream_t s);
void flag_from_distinct(const uint32_t* dflag, const uint32_t* pid, uint32_t m, uint32_t* occ_flag, hipStream_t s);
void flag_scatter_ones(const uint32_t* ids, uint32_t n, uint32_t* flags, hipStream_t s);                    // flags[ids[i]] = 1
void giant_distinct(const uint32_t* gids, uint32_t n, const uint32_t* rep, const uint32_t* dlen, uint32_t* which, uint32_t* glen,
                    hipStream_t s);
// bits[k / 32] bit k % 32 = flags[k] != 0
void giant_bits(const uint32_t* flags, uint32_t m, uint32_t* bits, hipStream_t s);
void popcount_words(const uint32_t* bits, uint32_t n, uint32_t* out, hipStream_t s);                        // out[i] = popcount(bits[i])
void giant_map(const uint32_t* gids, const uint32_t* gstart, uint32_t n, uint32_t* dmap, hipStream_t s);
void giant_occurrences(const uint32_t* gk, uint32_t n, const uint32_t* pid, const void* pstart, bool wide, const uint32_t* dmap,
                       uint64_t* gps, uint32_t* gbase, hipStream_t s);
void giant_group_flags(const uint32_t* esuf, const uint32_t* lcp, uint32_t nd, uint32_t* flags, hipStream_t s);
// sorted batch -> suffix array and BWT columns at [base, base + B)
void write_columns(const Ctx& c, const uint64_t* pos, uint32_t B, uint64_t base, SaCol sa, uint8_t* bwt, hipStream_t s);
// sorted batch -> its piece of the LCP column (lcp[j] for element j; entries that are not 0xffffffff were filled in by the
// sort and are kept); carry[0] = last element record of the batch before
// (have_carry = false: the batch starts the suffix array, lcp[0] = 0); err[2] counts pairs that are equal up to the end
// of alpha without being ordered by their parse ranks
void batch_lcp(const Ctx& c, const RmqView& R, const uint64_t* pos, uint32_t B, const uint64_t* carry, bool have_carry,
               uint32_t* lcp, uint32_t* err, hipStream_t s);
// expansion: the entry tables of the emitter (pfp_kernels.hpp EmitArgs) from a sorted batch of representative suffixes;
// tab: one uint4 per distinct phrase (occurrences, first slot of its inverted list, phrase length, -); ce_gs leaves as a
// 0 / 1 flag "first entry of a group of equal phrase suffixes" (group_ids turns it into group id + 1); err[1] counts
// elements that are not proper phrase suffixes of at least w characters
void expand_entries(const Ctx& c, const uint64_t* pos, const uint32_t* lcp, uint32_t B, const void* tab, uint32_t* ce_cnt,
                    uint32_t* ce_first, uint32_t* ce_offm1, uint8_t* ce_bwt, uint32_t* ce_gs, uint32_t* ce_hl, uint32_t* ce_slen,
                    uint32_t* err, hipStream_t s);
void group_ids(uint32_t* ce_gs, const uint32_t* gscan, uint32_t B, hipStream_t s);
void add_offset(void* table, bool wide, uint32_t n, uint64_t add, hipStream_t s);
void rep_bits(const uint32_t* pid, const uint32_t* rep, uint32_t m, uint32_t* bits, hipStream_t s);
// sorted phrases -> 1-based lexicographic rank per distinct phrase
void phrase_ranks(const Ctx& c, const uint64_t* pos, uint32_t D, const uint32_t* pid, uint32_t* prank, hipStream_t s);

}}  // namespace mmt::gk
