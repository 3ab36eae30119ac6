// fasta.hpp -- FASTA/FASTQ(.gz) reader for the CLI (host plumbing of A1).
// Behaviour follows what the reference gets from kseq.h (include/kseq.h:176-216):
// a record starts at a line whose first character is '>' or '@'; its name is the
// header up to the first whitespace; sequence lines are concatenated with line
// ends (and one trailing '\r') removed, empty lines skipped, until a line that
// starts with '>', '@' or '+'; after '+' as many quality characters as bases are
// skipped.  Bases are kept verbatim (upper-casing happens on the GPU).
#pragma once
#include <cstdint>
#include <memory>
#include <functional>
#include <string>
#include <vector>

namespace mmt {

struct FastaDoc {
    std::string path;                 // as given (normalised)
    std::vector<std::string> names;
    std::vector<uint64_t> lengths;    // per record
    uint64_t total = 0;
};

// Appends the bases of every record of `path` to `bases`; throws on I/O errors.
FastaDoc read_fasta(const std::string& path, std::vector<uint8_t>& bases);

// The bases of `path` INSTEAD of what `bases` held: plain FASTA files through the block reader (read() in cache-sized blocks,
// lines copied straight to their place: several times the stream reader's speed), everything else as read_fasta does.
FastaDoc read_fasta_replace(const std::string& path, std::vector<uint8_t>& bases);

// The concatenated bases of a collection: sized once, never zero-filled.
struct HostBytes {
    std::unique_ptr<uint8_t[]> p;
    size_t n = 0;
    void allocate(size_t bytes) { p.reset(new uint8_t[bytes ? bytes : 1]); n = bytes; }
    uint8_t* data() { return p.get(); }
    const uint8_t* data() const { return p.get(); }
    size_t size() const { return n; }
    const uint8_t* begin() const { return p.get(); }
    const uint8_t* end() const { return p.get() + n; }
};

// RefBuilder::build_input_file's reading half (src/ref_builder.cpp:211-314) for a whole collection: the files are
// independent, so they are inflated and parsed on as many host threads as the machine offers, then every thread
// copies the files it parsed to their place in the concatenation (file order).  doc_len[i] = bases of file i.
// Returns the index of the first file without bases (the caller reports it the way the reference does), or -1.
long read_fasta_files(const std::vector<std::string>& inputs, std::vector<FastaDoc>& docs, HostBytes& bases,
                      std::vector<uint64_t>& doc_len);

// The same without a second copy of the bases: every plain file is parsed straight into its own slot of one large
// buffer (huge pages, kept between calls), compressed files into vectors; the caller uploads document by document.
class HostArena {
public:
    HostArena() = default;
    HostArena(const HostArena&) = delete;
    ~HostArena();
    uint8_t* ensure(size_t bytes);      // grows only
private:
    uint8_t* p_ = nullptr;
    size_t cap_ = 0;
};
struct HostDocs {
    std::vector<const uint8_t*> ptr;    // bases of document i
    std::vector<uint64_t> len;
    std::vector<std::vector<uint8_t>> owned;
};
// Threads the readers use: the cgroup's CPU quota if there is one, else the machine's; MUMEMTO_READ_THREADS overrides.
size_t reader_threads();

// What a caller may do while the files are still being read (mmt_engine_run_files: send every document to the device as
// soon as it is parsed).  layout: once, before any file is read -- bytes of the arena, offset of every document's slot in it
// (N + 1 entries), whether every document lives in the arena (no compressed input).  ready: from a reader thread, document
// i is complete at arena + slot[i] with `len` bases.
//
// Chunked mode (optional, the command line's one-shot runs): the bases of a plain file do not go to the arena at all but
// through fixed-size buffers the caller owns -- page-locked, sent to the document's slot on the DEVICE as they fill up.  A
// process then never holds the collection in host memory: 6 GB of anonymous pages cost 0.2 s to fault in and 0.3 - 0.4 s to
// give back, wherever that happens (tests/micro/exit_probe2.cpp).  plan_chunks: once, before any file is read and before the
// arena exists (bytes of all slots, the slot table, whether every file is plain); true = every file goes through
// chunks(i).  ChunkTarget::swap(filled, n, offset, more): `filled` (nullptr at first) holds n bases that begin at base
// `offset` of the document; returns the buffer to fill next (nullptr when `more` is false -- the document is complete -- or
// to abort the whole read: read_fasta_collection then returns -2 and the caller reads again without chunks).
struct ChunkTarget {
    size_t chunk = 0;
    std::function<uint8_t*(uint8_t* filled, size_t n, uint64_t offset, bool more)> swap;
};
struct ReadHooks {
    std::function<void(const uint8_t* arena, size_t bytes, const std::vector<size_t>& slot, bool all_in_arena)> layout;
    std::function<void(size_t i, uint64_t len)> ready;
    std::function<bool(size_t bytes, const std::vector<size_t>& slot, bool all_plain)> plan_chunks;
    std::function<ChunkTarget(size_t i)> chunks;
};
long read_fasta_collection(const std::vector<std::string>& inputs, std::vector<FastaDoc>& docs, HostArena& arena,
                           HostDocs& out, const ReadHooks* hooks = nullptr);

// Writes n bytes to `path` (created / truncated) and closes it.
void write_file_bytes(const std::string& path, const void* data, size_t n);

// RefBuilder::write_lengths_file (src/ref_builder.cpp:193-209)
void write_lengths_file(const std::string& prefix, const std::vector<FastaDoc>& docs);

}  // namespace mmt
