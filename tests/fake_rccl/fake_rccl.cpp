// fake_rccl.cpp -- TEST DOUBLE for the ten RCCL entry points mumemto_amd/csrc/dist.cpp binds at run time.
//
// Why: the test boxes have ONE GPU, and RCCL wants one device per rank -- so the glue of dist.cpp (which table goes to
// whom, offsets `thresh + base[r]`, counts `hi[r] - base[r]` against the receiver's span, the order the pieces are gathered
// in) had only ever run with world = 1, where every loop is empty.  This library gives the same ten symbols with ranks =
// PROCESSES SHARING GPU 0: a message is a device-to-host copy into a file under /dev/shm, renamed into place; the
// receiver polls for it, copies it host-to-device and unlinks it.  A send is a RENDEZVOUS, as in NCCL: it is complete only when
// its receive has taken the message.  Operations between ncclGroupStart / ncclGroupEnd progress together (all sends are posted,
// all receives served, then every send waits for its message to be taken), so a group cannot deadlock on the order of its
// calls; a send OUTSIDE a group blocks until it is matched -- two ranks that send to each other before they receive hang
// (here: fail after FAKE_RCCL_TIMEOUT seconds, default 60, with "probable deadlock"), on this double as on the links.
// Message order per (source, destination) pair follows the order of the calls, as in NCCL.  A receive whose
// message has another size than the receiver asked for FAILS (ncclInvalidArgument): that is the check RCCL itself would
// not make and the reason this double is stricter than the real thing.
//
// Selected with MUMEMTO_RCCL_LIB=<path to this .so>.  Not a transport anybody should use for anything but tests.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct FakeComm {
    std::string dir;
    int rank = 0, world = 1;
    std::vector<uint64_t> sent, received;     // per peer: messages so far
};

struct Op { int kind; const void* src; void* dst; size_t bytes; int peer; FakeComm* comm; hipStream_t stream; };   // kind 0 send, 1 recv
thread_local int g_depth = 0;
thread_local std::vector<Op> g_queue;
thread_local std::string g_error;

size_t elem_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

std::string msg_path(const FakeComm& c, int src, int dst, uint64_t seq) {
    return c.dir + "/m_" + std::to_string(src) + "_" + std::to_string(dst) + "_" + std::to_string(seq);
}

ncclResult_t do_send(const Op& o, std::vector<std::string>* posted) {
    FakeComm& c = *o.comm;
    std::vector<char> host(o.bytes ? o.bytes : 1);
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (o.bytes && hipMemcpy(host.data(), o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const std::string path = msg_path(c, c.rank, o.peer, c.sent[o.peer]++);
    const std::string tmp = path + ".tmp";
    const int fd = ::open(tmp.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
    if (fd < 0) return ncclSystemError;
    size_t at = 0;
    while (at < o.bytes) {
        const ssize_t w = ::write(fd, host.data() + at, o.bytes - at);
        if (w <= 0) { ::close(fd); return ncclSystemError; }
        at += (size_t)w;
    }
    ::close(fd);
    if (::rename(tmp.c_str(), path.c_str()) != 0) return ncclSystemError;
    if (posted) posted->push_back(path);
    return ncclSuccess;
}

int timeout_s() { const char* e = std::getenv("FAKE_RCCL_TIMEOUT"); return e ? std::max(1, std::atoi(e)) : 60; }

// a posted send is complete when its receiver has taken (unlinked) the message
ncclResult_t wait_taken(const FakeComm& c, const std::string& path) {
    const auto t0 = std::chrono::steady_clock::now();
    struct stat sb;
    while (::stat(path.c_str(), &sb) == 0) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s())) {
            g_error = "fake rccl: rank " + std::to_string(c.rank) + ": nobody received " + path + " within " + std::to_string(timeout_s()) +
                      " s: probable deadlock (a send is complete only when its receive has run)";
            return ncclSystemError;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    return ncclSuccess;
}

ncclResult_t do_recv(const Op& o) {
    FakeComm& c = *o.comm;
    const std::string path = msg_path(c, o.peer, c.rank, c.received[o.peer]++);
    const auto t0 = std::chrono::steady_clock::now();
    struct stat sb;
    while (::stat(path.c_str(), &sb) != 0) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s())) {
            g_error = "fake rccl: rank " + std::to_string(c.rank) + " waited " + std::to_string(timeout_s()) + " s for " + path + ": probable deadlock";
            return ncclSystemError;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if ((size_t)sb.st_size != o.bytes) {
        g_error = "fake rccl: rank " + std::to_string(c.rank) + " expects " + std::to_string(o.bytes) + " bytes from rank " +
                  std::to_string(o.peer) + ", the message has " + std::to_string((size_t)sb.st_size);
        return ncclInvalidArgument;
    }
    std::vector<char> host(o.bytes ? o.bytes : 1);
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return ncclSystemError;
    size_t at = 0;
    while (at < o.bytes) {
        const ssize_t r = ::read(fd, host.data() + at, o.bytes - at);
        if (r <= 0) { ::close(fd); return ncclSystemError; }
        at += (size_t)r;
    }
    ::close(fd);
    ::unlink(path.c_str());
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (o.bytes && hipMemcpy(o.dst, host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

// the operations of one group (or of one collective) progress together: every send is posted, every receive served, then the
// sends wait for their receivers
ncclResult_t run_ops(std::vector<Op>& ops) {
    std::vector<std::string> posted;
    for (const Op& o : ops) if (o.kind == 0) { const ncclResult_t r = do_send(o, &posted); if (r != ncclSuccess) return r; }
    for (const Op& o : ops) if (o.kind == 1) { const ncclResult_t r = do_recv(o); if (r != ncclSuccess) return r; }
    if (!ops.empty()) for (const std::string& p : posted) { const ncclResult_t r = wait_taken(*ops[0].comm, p); if (r != ncclSuccess) return r; }
    return ncclSuccess;
}

ncclResult_t submit(std::vector<Op> ops) {
    if (g_depth > 0) { g_queue.insert(g_queue.end(), ops.begin(), ops.end()); return ncclSuccess; }
    return run_ops(ops);
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof(*id));
    char tmpl[] = "/dev/shm/mmt_fake_rccl_XXXXXX";
    if (!::mkdtemp(tmpl)) return ncclSystemError;
    std::snprintf(id->internal, sizeof(id->internal), "%s", tmpl);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    FakeComm* c = new FakeComm();
    c->dir = id.internal; c->rank = rank; c->world = nranks;
    c->sent.assign((size_t)nranks, 0); c->received.assign((size_t)nranks, 0);
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (c && c->rank == 0) ::rmdir(c->dir.c_str());      // succeeds once every message has been consumed
    delete c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm,
                                                             hipStream_t s) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (peer < 0 || peer >= c->world || peer == c->rank || !elem_size(t)) return ncclInvalidArgument;
    return submit({Op{0, buf, nullptr, count * elem_size(t), peer, c, s}});
}

__attribute__((visibility("default"))) ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm,
                                                             hipStream_t s) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (peer < 0 || peer >= c->world || peer == c->rank || !elem_size(t)) return ncclInvalidArgument;
    return submit({Op{1, nullptr, buf, count * elem_size(t), peer, c, s}});
}

__attribute__((visibility("default"))) ncclResult_t ncclBroadcast(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t t, int root,
                                                                  ncclComm_t comm, hipStream_t s) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    const size_t bytes = count * elem_size(t);
    if (root < 0 || root >= c->world || !elem_size(t)) return ncclInvalidArgument;
    std::vector<Op> ops;
    if (c->rank == root) {
        for (int r = 0; r < c->world; r++) if (r != root) ops.push_back(Op{0, sendbuf, nullptr, bytes, r, c, s});
        if (recvbuf != sendbuf && bytes) {
            if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(recvbuf, sendbuf, bytes, hipMemcpyDeviceToDevice) != hipSuccess)
                return ncclUnhandledCudaError;
        }
    } else {
        ops.push_back(Op{1, nullptr, recvbuf, bytes, root, c, s});
    }
    return submit(ops);
}

__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void* sendbuf, void* recvbuf, size_t sendcount, ncclDataType_t t,
                                                                  ncclComm_t comm, hipStream_t s) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    // FAKE_RCCL_FAIL: every rank's first collective of an exchange fails (how bench.py's trial step is tested)
    if (std::getenv("FAKE_RCCL_FAIL")) return ncclSystemError;
    const size_t bytes = sendcount * elem_size(t);
    if (!elem_size(t)) return ncclInvalidArgument;
    std::vector<Op> ops;
    for (int r = 0; r < c->world; r++) {
        if (r == c->rank) continue;
        ops.push_back(Op{0, sendbuf, nullptr, bytes, r, c, s});
        ops.push_back(Op{1, nullptr, static_cast<char*>(recvbuf) + (size_t)r * bytes, bytes, r, c, s});
    }
    if (bytes) {
        if (hipStreamSynchronize(s) != hipSuccess ||
            hipMemcpy(static_cast<char*>(recvbuf) + (size_t)c->rank * bytes, sendbuf, bytes, hipMemcpyDeviceToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    }
    return submit(ops);
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }

__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_queue);
    return run_ops(ops);
}

__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t r) {
    if (!g_error.empty()) return g_error.c_str();
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake rccl: HIP error";
        case ncclSystemError: return "fake rccl: system error";
        case ncclInvalidArgument: return "fake rccl: invalid argument";
        case ncclInvalidUsage: return "fake rccl: invalid usage";
        default: return "fake rccl: error";
    }
}

}  // extern "C"
