// fasta.hpp -- FASTA/FASTQ(.gz) reader for the CLI (host plumbing of A1).
// Behaviour follows what the reference gets from kseq.h (include/kseq.h:176-216):
// a record starts at a line whose first character is '>' or '@'; its name is the
// header up to the first whitespace; sequence lines are concatenated with line
// ends (and one trailing '\r') removed, empty lines skipped, until a line that
// starts with '>', '@' or '+'; after '+' as many quality characters as bases are
// skipped.  Bases are kept verbatim (upper-casing happens on the GPU).
#pragma once
#include <cstdint>
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

}  // namespace mmt
