// pool.hpp -- one growing device heap per GPU behind every DevBuf.
//
// Why: on this stack device memory that a process gives back is wiped by the driver (~33 GB/s), and the next
// allocation that receives those pages waits for the wipe -- a stage that frees 48 GB and allocates 48 GB of a
// different size stalls for 1.5 s (tests/micro/alloc_probe.cpp; round 1 saw the same thing as "hipMalloc costs 40-180
// ms per GB after a hipFree").  A run therefore takes physical memory from the driver once and recycles it in user
// space: the heap is one reserved virtual range (hipMemAddressReserve) whose mapped prefix grows in 1 GiB chunks
// (hipMemCreate / hipMemMap), with a coalescing best-fit free list on top.  Stages can release their scratch as soon
// as it is dead, so the footprint of a run is its largest stage, not the sum of all stages.
// MUMEMTO_POOL=0 (or a driver without the virtual-memory API) falls back to hipMalloc / hipFree per buffer.
#pragma once
#include <cstddef>

namespace mmt { namespace pool {

void* alloc(size_t bytes);        // current device; throws HipError when the device is full
void release(void* p);
struct Stats { bool pooled; size_t mapped, live, peak, largest_free; double map_seconds; };
Stats stats(int device);
size_t available(int device);     // driver-free bytes + unused bytes of the heap
// maps the large region up to `bytes` (or what the driver has, less 8 GB) NOW: a one-shot process that knows about how much it
// will need does it while the GPU is idle -- a chunk mapped beside running work waits for it (30 ms a time, measured)
void premap(int device, size_t bytes);
// device memory the heap leaves alone from now on, as MUMEMTO_HEAP_RESERVE does from the start (~0: back to the environment's
// value): how a test makes a 288 GB device a smaller one in the middle of a process
void set_reserve(size_t bytes);
void trim();                      // unmap everything if no block is live (engine teardown in long-lived processes)
// One-shot processes: the wholly free chunks at the top of the heap go back to the driver on a helper thread while the run
// goes on (what is still mapped at exit is torn down on the way out: 0.57 s for 116 GB).  shrink_wait joins the helper.
void shrink_async(int device);
void shrink_wait();

}}  // namespace mmt::pool
