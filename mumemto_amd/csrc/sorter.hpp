// sorter.hpp -- suffix sorting by prefix doubling (Manber-Myers / Larsson-Sadakane
// scheme) on top of a device radix sort.  Used for the whole text (A8, replaces
// gsacak(text), include/direct_gsacak.hpp:62), for the PFP dictionary (replaces
// gsacak(d), include/dictionary.hpp:133) and for the parse (replaces sacak_int,
// include/parse.hpp:85).
#pragma once
#include <cstdint>

#include "device_utils.hpp"

namespace mmt {

class DoublingSorter {
public:
    DoublingSorter() = default;
    DoublingSorter(const DoublingSorter&) = delete;
    ~DoublingSorter();
    // The caller fills keys_in() / vals_in() for n suffixes: keys = first h0 symbols of every
    // suffix packed big-endian into `key_bits` bits (past-the-end = 0 = smallest), vals = 0..n-1.
    void reserve(uint32_t n);
    uint64_t* keys_in() { return keys_a_.get(); }
    uint32_t* vals_in() { return sac_a_.get(); }
    // Sorts; sa_out[j] = j-th smallest suffix, rank_out = its inverse.  Returns #doubling rounds.
    // lsb_unique: keys carry a low "this suffix reached its unique terminator" bit (k_pack_keys with sep_code):
    // such suffixes are final after the first sort, in position order among equal keys.
    int sort(uint32_t n, int key_bits, uint64_t h0, uint32_t* sa_out, uint32_t* rank_out, DevBuf<uint8_t>& temp,
             hipStream_t s, bool lsb_unique = false);
    // scratch columns, reusable by the caller between sorts (ensure() what is needed)
    DevBuf<uint64_t>& keys_a() { return keys_a_; }
    DevBuf<uint64_t>& keys_b() { return keys_b_; }
    DevBuf<uint32_t>& u32_a() { return sac_a_; }
    DevBuf<uint32_t>& u32_b() { return sac_b_; }
    // give every scratch column back (one-shot runs: later stages then reuse the memory)
    void release();

private:
    hipStream_t side_ = nullptr;         // the rank scatter of a round runs beside the compaction of the active list
    hipEvent_t ev_main_ = nullptr, ev_side_ = nullptr;
    void sort_round(uint32_t m, int shift, DevBuf<uint8_t>& temp, hipStream_t s);
    DevBuf<uint64_t> keys_a_, keys_b_;
    DevBuf<uint32_t> sac_a_, sac_b_, pos_a_, pos_b_, headc_, headc_b_, count_, bound_, big_begin_, big_end_;
    DevBuf<uint8_t> hf_, tile_big_;      // new heads (4 B) + flags (1 B) per tied suffix of a round; marks of the long ranges' tiles
};

}  // namespace mmt
