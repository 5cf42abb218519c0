// Test-only stand-in for the RCCL entry points libmhpmvo.so binds at run time (monohair_amd/csrc/capi.cpp: rccl_load).
//
// Why: RCCL refuses two ranks on one device, and the boxes the tests run on have ONE GPU, so the nranks > 1 branches of
// mh_volume_reduce / mh_volume_gather (group semantics, peer numbers, slab offsets and counts) could never execute there.
// With MH_RCCL_LIB=<this library> they do: several processes share the GPU, and this library moves the bytes between them
// through POSIX shared memory (device -> host segment -> device).  It is NOT a product path and nothing under
// monohair_amd/ knows about it beyond the MH_RCCL_LIB override.
//
// What makes it a meaningful check and not a mock that agrees with whatever the caller does:
//  * it is compiled against <rccl/rccl.h>: every function below has the real header's prototype, so the hand-written
//    function-pointer types and enum values of capi.cpp are exercised against the real ABI (ncclUniqueId by value,
//    ncclFloat32 == 7, ncclSum == 0, argument order);
//  * NCCL's semantics are enforced where real RCCL would hang or silently corrupt: ncclSend/ncclRecv inside a group are
//    deferred to ncclGroupEnd, a receive must find a send of exactly the same byte count from exactly that peer, every
//    wait has a time-out and returns ncclSystemError instead of hanging, ranks outside [0, nranks) are rejected;
//  * operations are ordered on the caller's stream (the stream is synchronised before the buffer is read and the
//    receiving copy is issued on it).
//
// Build: hipcc -shared -fPIC -O2 tests/fake_rccl.cpp -o tests/lib/libfake_rccl.so   (tests/conftest.py does it on demand)
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
const int MAX_RANKS = 16;
const double WAIT_SECONDS = 120.0;

struct Mailbox {                       // one per ordered pair (src, dst)
    std::atomic<unsigned long long> posted;     // messages the sender has published
    std::atomic<unsigned long long> consumed;   // messages the receiver has taken
    unsigned long long bytes;                   // size of message number `posted`
};
struct Control {
    std::atomic<int> joined;
    std::atomic<int> left;
    int nranks;
    Mailbox box[MAX_RANKS][MAX_RANKS];
};
struct Op {
    int kind;          // 0 send, 1 recv
    const void *src;
    void *dst;
    size_t bytes;
    int peer;
    hipStream_t stream;
};
struct FakeComm {
    int rank, nranks;
    std::string name;
    Control *ctl;
};
thread_local int g_depth = 0;
thread_local std::vector<std::pair<FakeComm *, Op>> g_queue;

double now() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}
void nap() {
    timespec t = {0, 200000};
    nanosleep(&t, nullptr);
}
std::string seg_name(const FakeComm *c, int src, int dst, unsigned long long seq) {
    char b[200];
    snprintf(b, sizeof b, "%s_%d_%d_%llu", c->name.c_str(), src, dst, seq);
    return b;
}

ncclResult_t do_send(FakeComm *c, const Op &op) {
    Mailbox &m = c->ctl->box[c->rank][op.peer];
    const double t0 = now();
    while (m.consumed.load(std::memory_order_acquire) != m.posted.load(std::memory_order_acquire)) {   // one in flight per pair
        if (now() - t0 > WAIT_SECONDS) return ncclSystemError;
        nap();
    }
    const unsigned long long seq = m.posted.load() + 1;
    const std::string n = seg_name(c, c->rank, op.peer, seq);
    int fd = shm_open(n.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (op.bytes && ftruncate(fd, (off_t)op.bytes) != 0) { close(fd); return ncclSystemError; }
    if (op.bytes) {
        void *p = mmap(nullptr, op.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (p == MAP_FAILED) { close(fd); return ncclSystemError; }
        // stream order: everything queued on the caller's stream before this call has produced the buffer
        if (hipStreamSynchronize(op.stream) != hipSuccess || hipMemcpy(p, op.src, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) {
            munmap(p, op.bytes); close(fd); return ncclUnhandledCudaError;
        }
        munmap(p, op.bytes);
    }
    close(fd);
    m.bytes = op.bytes;
    m.posted.store(seq, std::memory_order_release);
    return ncclSuccess;
}

ncclResult_t do_recv(FakeComm *c, const Op &op) {
    Mailbox &m = c->ctl->box[op.peer][c->rank];
    const unsigned long long want = m.consumed.load() + 1;
    const double t0 = now();
    while (m.posted.load(std::memory_order_acquire) < want) {
        if (now() - t0 > WAIT_SECONDS) return ncclSystemError;      // real RCCL would hang here
        nap();
    }
    const std::string n = seg_name(c, op.peer, c->rank, want);
    if (m.bytes != op.bytes) {                                      // real RCCL would hang or corrupt memory
        fprintf(stderr, "fake_rccl: rank %d expects %zu bytes from rank %d, which sent %llu\n", c->rank, op.bytes, op.peer,
                m.bytes);
        shm_unlink(n.c_str());                                      // the message is dropped, the pair stays usable
        m.consumed.store(want, std::memory_order_release);
        return ncclInvalidArgument;
    }
    int fd = shm_open(n.c_str(), O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    ncclResult_t rc = ncclSuccess;
    if (op.bytes) {
        void *p = mmap(nullptr, op.bytes, PROT_READ, MAP_SHARED, fd, 0);
        if (p == MAP_FAILED) { close(fd); return ncclSystemError; }
        if (hipMemcpyAsync(op.dst, p, op.bytes, hipMemcpyHostToDevice, op.stream) != hipSuccess ||
            hipStreamSynchronize(op.stream) != hipSuccess)
            rc = ncclUnhandledCudaError;
        munmap(p, op.bytes);
    }
    close(fd);
    shm_unlink(n.c_str());
    m.consumed.store(want, std::memory_order_release);
    return rc;
}

ncclResult_t run(FakeComm *c, const Op &op) { return op.kind == 0 ? do_send(c, op) : do_recv(c, op); }

ncclResult_t submit(FakeComm *c, const Op &op) {
    if (!c || op.peer < 0 || op.peer >= c->nranks) return ncclInvalidArgument;
    if (g_depth > 0) {
        g_queue.emplace_back(c, op);
        return ncclSuccess;
    }
    return run(c, op);
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}
}   // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, sizeof id->internal);
    unsigned char rnd[12] = {0};
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, rnd, sizeof rnd) != (ssize_t)sizeof rnd) { if (fd >= 0) close(fd); return ncclSystemError; }
    close(fd);
    char *o = id->internal;
    o += sprintf(o, "/mhfake_");
    for (unsigned char b : rnd) o += sprintf(o, "%02x", b);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (strncmp(id.internal, "/mhfake_", 8) != 0 || id.internal[sizeof id.internal - 1] != 0) return ncclInvalidArgument;
    FakeComm *c = new FakeComm;
    c->rank = rank;
    c->nranks = nranks;
    c->name = id.internal;
    int fd = shm_open((c->name + "_ctl").c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Control)) != 0) { delete c; return ncclSystemError; }   // fresh segments are zero
    void *p = mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->ctl = (Control *)p;
    c->ctl->nranks = nranks;
    c->ctl->joined.fetch_add(1);
    const double t0 = now();
    while (c->ctl->joined.load() < nranks) {          // ncclCommInitRank is collective: it returns once everybody is in
        if (now() - t0 > WAIT_SECONDS) return ncclSystemError;
        nap();
    }
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    FakeComm *c = (FakeComm *)comm;
    if (!c) return ncclInvalidArgument;
    const int left = c->ctl->left.fetch_add(1) + 1;
    const bool last = left == c->nranks;
    munmap(c->ctl, sizeof(Control));
    if (last) shm_unlink((c->name + "_ctl").c_str());
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    // sends first (they are buffered and never wait for the receiver's matching call), then receives: any legal
    // combination of grouped sends and receives completes, as in NCCL
    ncclResult_t rc = ncclSuccess;
    for (auto &q : g_queue)
        if (q.second.kind == 0 && rc == ncclSuccess) rc = run(q.first, q.second);
    for (auto &q : g_queue)
        if (q.second.kind == 1 && rc == ncclSuccess) rc = run(q.first, q.second);
    g_queue.clear();
    return rc;
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream) {
    const size_t tb = type_bytes(datatype);
    if (!tb || (!sendbuff && count)) return ncclInvalidArgument;
    Op op = {0, sendbuff, nullptr, count * tb, peer, stream};
    return submit((FakeComm *)comm, op);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    const size_t tb = type_bytes(datatype);
    if (!tb || (!recvbuff && count)) return ncclInvalidArgument;
    Op op = {1, nullptr, recvbuff, count * tb, peer, stream};
    return submit((FakeComm *)comm, op);
}

// fp32 sum only (what mh_volume_reduce mode 1 issues); the root adds the peers' buffers in rank order on the host
ncclResult_t ncclReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root,
                        ncclComm_t comm, hipStream_t stream) {
    FakeComm *c = (FakeComm *)comm;
    if (!c || datatype != ncclFloat32 || op != ncclSum || root < 0 || root >= c->nranks || !sendbuff) return ncclInvalidArgument;
    if (g_depth > 0) return ncclInvalidUsage;       // (not needed by the caller; keeps the stand-in small)
    const size_t bytes = count * sizeof(float);
    if (c->rank != root) {
        Op s = {0, sendbuff, nullptr, bytes, root, stream};
        return do_send(c, s);
    }
    if (!recvbuff) return ncclInvalidArgument;
    std::vector<float> acc(count), tmp(count);
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(acc.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    float *stage = nullptr;
    if (hipMalloc(&stage, bytes ? bytes : 4) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t rc = ncclSuccess;
    for (int r = 0; r < c->nranks && rc == ncclSuccess; ++r) {
        if (r == root) continue;
        Op q = {1, nullptr, stage, bytes, r, stream};
        rc = do_recv(c, q);
        if (rc == ncclSuccess && hipMemcpy(tmp.data(), stage, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = ncclUnhandledCudaError;
        if (rc == ncclSuccess)
            for (size_t i = 0; i < count; ++i) acc[i] += tmp[i];
    }
    (void)hipFree(stage);
    if (rc == ncclSuccess && hipMemcpy(recvbuff, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    return rc;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "fake_rccl: success";
        case ncclUnhandledCudaError: return "fake_rccl: HIP error";
        case ncclSystemError: return "fake_rccl: system error or time-out waiting for a peer";
        case ncclInvalidArgument: return "fake_rccl: invalid argument (peer out of range, or send/recv sizes do not match)";
        case ncclInvalidUsage: return "fake_rccl: invalid usage";
        default: return "fake_rccl: error";
    }
}

}   // extern "C"
