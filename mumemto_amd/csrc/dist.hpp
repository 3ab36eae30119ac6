// dist.hpp -- the multi-GPU exchange against RCCL (dist.cpp); C ABI: mmt_comm_* / mmt_dist_* in mumemto_gpu.h.
#pragma once
#include <cstdint>
#include <string>

#include "engine.hpp"
#include "merge_types.hpp"

namespace mmt {

struct Comm;
void comm_unique_id(uint8_t out[128]);                                            // rank 0, handed to the others out of band
Comm* comm_create(Engine& e, int rank, int world, const uint8_t id[128]);         // collective
void comm_destroy(Comm* c);
MergedRows dist_merge(Comm& c, uint32_t min_len, bool* is_root);                  // collective; rows on rank 0
MergedRows dist_merge_ranges(Comm& c, uint32_t min_len, bool* is_root);           // the same, every rank folds its slice of the anchor
int comm_world(const Comm& c);
std::string dist_gather_text(Comm& c);                                            // collective; bytes on rank 0
void dist_loopback(Comm& c, uint64_t out[8]);
void dist_selftest(Comm& c, uint64_t elements, uint32_t width, uint64_t out[4]);     // one message of that size to this rank itself, in pieces                                     // the exchange's messages with this rank as its own peer
// (no exchange of columns: a rank of a sharded run produces, scans and drops its own share of the stream)

}  // namespace mmt
