// fake_rccl.cpp -- TEST INFRASTRUCTURE.  The eight RCCL entry points the engine binds
// (pydcop_amd/csrc/engine.hip: load_rccl), implemented over files so that the native
// exchange of the sharded path can be exercised on a box without GPUs, between threads
// of one process or between processes: a send of a group becomes one file
// <dir>/<id>/m_<src>_<dst>_<seq> (written under another name, then renamed), a receive
// polls for its file, reads and removes it.  "Device" memory of the emulated engine is
// host memory and its streams are synchronous, so ncclGroupEnd can do the copies on the
// spot.  Buffered sends first, then receives: no rendezvous, no deadlock.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace {

struct Comm {
    std::string dir;
    int rank = 0, world = 0;
    std::map<std::pair<int, int>, uint64_t> seq;  // (src, dst) -> next message number
};

struct Op {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
    Comm* comm;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
std::atomic<unsigned> g_ids{0};

std::string base_dir() {
    const char* d = getenv("FAKE_RCCL_DIR");
    return d && *d ? d : "/tmp";
}

size_t dt_size(int dt) { return dt == 8 ? 8 : dt == 7 ? 4 : dt <= 1 ? 1 : 0; }

bool exists(const std::string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

int wait_for(const std::string& p, double seconds) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!exists(p)) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return 1;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    return 0;
}

int write_file(const std::string& p, const void* data, size_t n) {
    const std::string tmp = p + ".part";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return 1;
    if (n && fwrite(data, 1, n, f) != n) {
        fclose(f);
        return 1;
    }
    fclose(f);
    return rename(tmp.c_str(), p.c_str()) != 0;
}

int run_ops(std::vector<Op>& ops) {
    for (const Op& o : ops)
        if (o.send) {
            Comm* c = o.comm;
            const uint64_t s = c->seq[{c->rank, o.peer}]++;
            char name[96];
            snprintf(name, sizeof name, "/m_%d_%d_%llu", c->rank, o.peer, (unsigned long long)s);
            if (write_file(c->dir + name, o.buf, o.bytes)) return 2;
        }
    for (const Op& o : ops)
        if (!o.send) {
            Comm* c = o.comm;
            const uint64_t s = c->seq[{o.peer, c->rank}]++;
            char name[96];
            snprintf(name, sizeof name, "/m_%d_%d_%llu", o.peer, c->rank, (unsigned long long)s);
            const std::string p = c->dir + name;
            if (wait_for(p, 120.0)) return 6;
            FILE* f = fopen(p.c_str(), "rb");
            if (!f) return 2;
            const size_t got = o.bytes ? fread(o.buf, 1, o.bytes, f) : 0;
            // a count mismatch between the two sides is the bug this fake exists to catch
            fseek(f, 0, SEEK_END);
            const long len = ftell(f);
            fclose(f);
            unlink(p.c_str());
            if (got != o.bytes || (size_t)len != o.bytes) return 4;
        }
    ops.clear();
    return 0;
}

}  // namespace

extern "C" {

struct ncclUniqueId {
    char internal[128];
};

int ncclGetUniqueId(ncclUniqueId* id) {
    memset(id->internal, 0, sizeof id->internal);
    snprintf(id->internal, sizeof id->internal, "fake_rccl_%d_%u_%lld", (int)getpid(), g_ids++,
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return 4;
    id.internal[127] = 0;
    Comm* c = new Comm();
    c->dir = base_dir() + "/" + id.internal;
    c->rank = rank;
    c->world = nranks;
    mkdir(c->dir.c_str(), 0700);  // every rank may be first
    char name[64];
    snprintf(name, sizeof name, "/joined_%d", rank);
    if (write_file(c->dir + name, "", 0)) return 2;
    for (int q = 0; q < nranks; ++q) {  // ncclCommInitRank is a collective
        snprintf(name, sizeof name, "/joined_%d", q);
        if (wait_for(c->dir + name, 120.0)) return 6;
    }
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm) {
    delete (Comm*)comm;
    return 0;
}

int ncclGroupStart() {
    ++g_depth;
    return 0;
}

int ncclGroupEnd() {
    if (g_depth <= 0) return 5;
    if (--g_depth == 0) return run_ops(g_ops);
    return 0;
}

int ncclSend(const void* buf, size_t count, int dt, int peer, void* comm, void*) {
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->world || !dt_size(dt)) return 4;
    g_ops.push_back({true, (void*)buf, count * dt_size(dt), peer, c});
    return g_depth ? 0 : run_ops(g_ops);
}

int ncclRecv(void* buf, size_t count, int dt, int peer, void* comm, void*) {
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->world || !dt_size(dt)) return 4;
    g_ops.push_back({false, buf, count * dt_size(dt), peer, c});
    return g_depth ? 0 : run_ops(g_ops);
}

const char* ncclGetErrorString(int r) {
    switch (r) {
        case 0: return "no error";
        case 2: return "fake rccl: file error";
        case 4: return "fake rccl: invalid argument / count mismatch";
        case 5: return "fake rccl: invalid usage";
        case 6: return "fake rccl: peer timeout";
        default: return "fake rccl: error";
    }
}

}  // extern "C"
