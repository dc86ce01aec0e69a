// rccl_fake.cpp -- TEST STAND-IN for the NCCL entry points libvamp_hip.so resolves at run time
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclGetErrorString; optional
// ncclCommCount, ncclCommUserRank).
//
// RCCL refuses two ranks on one GPU, and the test box has one GPU.  This library lets several
// processes on ONE device form a "communicator": the all-gather goes device -> POSIX shared memory
// -> device with a process-shared barrier on either side.  It is synchronous (it drains the stream it
// is given), so it checks the in-library exchange LOGIC at world > 1 -- shard layout, pieces, pack
// and scatter, step bookkeeping, the call sequence of bench.py under torch.distributed.run -- not the
// asynchronous ordering of the real collective (that is what the world-1 RCCL test and the stream
// tests cover).  Selected with VAMP_RCCL_LIB; never used outside tests/.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

namespace {
constexpr size_t DATA_BYTES = 96u << 20;          // room for one gathered piece
struct Header {
    volatile int magic;
    int count;
    int gen;
    int attached;
};
struct Comm {
    int rank, n;
    char name[64];
    Header* hdr;
    char* data;
    size_t total;
};
void barrier(Comm* c) {
    const int g = __atomic_load_n(&c->hdr->gen, __ATOMIC_ACQUIRE);
    if (__atomic_add_fetch(&c->hdr->count, 1, __ATOMIC_ACQ_REL) == c->n) {
        __atomic_store_n(&c->hdr->count, 0, __ATOMIC_RELEASE);
        __atomic_add_fetch(&c->hdr->gen, 1, __ATOMIC_ACQ_REL);
    } else {
        while (__atomic_load_n(&c->hdr->gen, __ATOMIC_ACQUIRE) == g) sched_yield();
    }
}
}  // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId* id) {
    if (getenv("VAMP_FAKE_RCCL_FAIL_ID")) return 2;       // tests of bench.py's fallback: the id cannot be made
    std::memset(id->internal, 0, sizeof(id->internal));
    std::snprintf(id->internal, sizeof(id->internal), "/vampfake-%d-%ld", (int)getpid(), (long)time(nullptr));
    return 0;
}

int ncclCommInitRank(void** comm, int n, ncclUniqueId id, int rank) {
    if (!comm || n < 1 || rank < 0 || rank >= n) return 4;
    if (const char* bad = getenv("VAMP_FAKE_RCCL_FAIL_INIT_RANK"))      // ... or ONE rank fails to join
        if (std::atoi(bad) == rank) return 2;
    Comm* c = new Comm();
    c->rank = rank;
    c->n = n;
    std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->total = sizeof(Header) + 64 + DATA_BYTES;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 2;
    if (ftruncate(fd, (off_t)c->total) != 0) return 2;
    void* p = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    c->hdr = (Header*)p;
    c->data = (char*)p + sizeof(Header) + 64;
    if (rank == 0) {
        c->hdr->count = 0;
        c->hdr->gen = 0;
        __atomic_store_n(&c->hdr->magic, 0x56414d50, __ATOMIC_RELEASE);
    } else {
        while (__atomic_load_n(&c->hdr->magic, __ATOMIC_ACQUIRE) != 0x56414d50) sched_yield();
    }
    __atomic_add_fetch(&c->hdr->attached, 1, __ATOMIC_ACQ_REL);
    barrier(c);
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return 0;
    barrier(c);
    if (c->rank == 0) shm_unlink(c->name);
    munmap((void*)c->hdr, c->total);
    delete c;
    return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t bytes = count * (dtype == 8 || dtype == 4 || dtype == 5 ? 8 : 4);
    if (bytes * c->n > DATA_BYTES) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (hipMemcpy(c->data + bytes * c->rank, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    barrier(c);
    if (hipMemcpy(recv, c->data, bytes * c->n, hipMemcpyHostToDevice) != hipSuccess) return 1;
    barrier(c);
    return 0;
}

int ncclCommCount(void* comm, int* n) {
    if (!comm || !n) return 4;
    *n = ((Comm*)comm)->n;
    return 0;
}
int ncclCommUserRank(void* comm, int* r) {
    if (!comm || !r) return 4;
    *r = ((Comm*)comm)->rank;
    return 0;
}

const char* ncclGetErrorString(int r) {
    return r == 0 ? "no error" : r == 1 ? "fake rccl: HIP error" : r == 2 ? "fake rccl: shared memory error" : "fake rccl: invalid usage";
}

}  // extern "C"
