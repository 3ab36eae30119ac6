// sorter.hpp -- suffix sorting by prefix doubling (Manber-Myers / Larsson-Sadakane
// scheme) on top of a device radix sort.  Used for the whole text (A8, replaces
// gsacak(text), include/direct_gsacak.hpp:62), for the PFP dictionary (replaces
// gsacak(d), include/dictionary.hpp:133) and for the parse (replaces sacak_int,
// include/parse.hpp:85).
#pragma once
#include <cstdint>
#include <vector>

#include "device_utils.hpp"

namespace mmt {

// Long runs of one symbol in the text that is being sorted (optional; the PFP dictionary passes them).  Prefix doubling needs
// log2(run length / h0) rounds for the suffixes inside a run c^j X, all of them tied in ONE bucket at first -- assembly gaps
// broken by insertions put a fifth of a realistic dictionary there, ten rounds of a device-wide sort each.  Their order is a
// function of (class of X0 against c, j, X): with the ends of the long runs listed, the bucket c^h0 of the first sort is
// sorted once by that key (kernels.hip, k_run_keys) before the doubling rounds begin.
struct RunRefine {
    const uint8_t* text = nullptr; uint32_t n = 0;        // the text and the symbol codes of the first keys
    const uint8_t* code = nullptr; int bits = 0, chars = 0, sigma = 0;
    const uint32_t* ends = nullptr; uint32_t n_ends = 0;  // ascending last positions of the runs of `chars` symbols or more (device)
};

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
             hipStream_t s, bool lsb_unique = false, const RunRefine* runs = nullptr);
    uint64_t run_refined() const { return run_refined_; }   // suffixes the last sort ordered by their runs
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
    void refine_runs(uint32_t n, const RunRefine& R, uint32_t* sa, std::vector<uint32_t>& forced, DevBuf<uint8_t>& temp, hipStream_t s);
    uint64_t run_refined_ = 0;
    DevBuf<uint64_t> run_probe_;
    DevBuf<uint32_t> run_range_;
    DevBuf<uint8_t> hf_, tile_big_;      // new heads (4 B) + flags (1 B) per tied suffix of a round; marks of the long ranges' tiles
};

}  // namespace mmt
