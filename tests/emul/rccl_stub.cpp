// TEST INFRASTRUCTURE (never part of libdpx_hip.so): a stand-in for librccl that moves bytes between PROCESSES of one host through
// a POSIX shared-memory segment, so that the multi-rank branches of delta-prox_amd/csrc/dpx_comm.hip (world - 1 direct sends of the
// all-gather, the root's sends of the scatter, broadcast) execute with world > 1 in the CPU test suite.  It implements the subset
// of the public NCCL / RCCL C API that dpx_comm.hip binds with dlsym -- ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
// ncclBroadcast, ncclAllGather, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString -- with the library's
// semantics as far as a host-memory transport has them: sends and receives inside a group are queued and progressed together
// at ncclGroupEnd (no deadlock whatever the posting order), pairs match in posting order per (source, destination), "device"
// pointers are host pointers (the SIMT emulator's address space) and the stream argument is ignored (the emulator's streams
// are synchronous).  Selected with dpx_comm_use_library() / DPX_RCCL_LIB.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

constexpr int kMaxWorld = 16;
constexpr size_t kRing = 256 * 1024;                    // bytes per (source -> destination) channel
constexpr double kTimeout = 120.0;                      // seconds without progress before a collective gives up

struct Channel {                                        // single-producer / single-consumer byte stream
  std::atomic<uint64_t> head;                           // bytes written by the source
  std::atomic<uint64_t> tail;                           // bytes consumed by the destination
  char pad[48];
  unsigned char data[kRing];
};

struct Segment {
  std::atomic<int> ready;                               // set by the creating rank once the header is initialised
  std::atomic<int> joined;                              // ranks that have mapped the segment
  std::atomic<int> left;                                // ranks that have destroyed their communicator
  int world;
  Channel ch[1];                                        // [world][world], row = source
};

struct Comm {
  Segment* seg;
  size_t bytes;
  int rank, world;
  char name[64];
};

struct Op {
  Comm* c;
  bool send;
  int peer;
  unsigned char* p;
  size_t n, done;
};

thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;

enum { ncclSuccess = 0, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 };

double now() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

size_t type_size(int dt) {
  switch (dt) {
    case 0: case 1: return 1;                           // int8 / uint8
    case 2: case 3: case 7: return 4;                   // int32 / uint32 / float32
    case 4: case 5: case 8: return 8;                   // int64 / uint64 / float64
    case 6: case 9: return 2;                           // float16 / bfloat16
    default: return 0;
  }
}

Channel& chan(Comm* c, int src, int dst) { return c->seg->ch[(size_t)src * c->world + dst]; }

// moves as many bytes of `o` as the channel allows right now; returns true when something moved
bool progress(Op& o) {
  if (o.done == o.n) return false;
  Channel& C = o.send ? chan(o.c, o.c->rank, o.peer) : chan(o.c, o.peer, o.c->rank);
  const uint64_t h = C.head.load(std::memory_order_acquire), t = C.tail.load(std::memory_order_acquire);
  size_t room = o.send ? kRing - (size_t)(h - t) : (size_t)(h - t);
  if (room > o.n - o.done) room = o.n - o.done;
  if (!room) return false;
  size_t pos = (size_t)((o.send ? h : t) % kRing), moved = 0;
  while (moved < room) {
    const size_t piece = (room - moved < kRing - pos) ? room - moved : kRing - pos;
    if (o.send) std::memcpy(C.data + pos, o.p + o.done + moved, piece);
    else std::memcpy(o.p + o.done + moved, C.data + pos, piece);
    moved += piece;
    pos = (pos + piece) % kRing;
  }
  o.done += room;
  if (o.send) C.head.store(h + room, std::memory_order_release);
  else C.tail.store(t + room, std::memory_order_release);
  return true;
}

int run(std::vector<Op>& ops) {
  double last = now();
  for (;;) {
    bool all = true, any = false;
    for (size_t i = 0; i < ops.size(); ++i) {
      Op& o = ops[i];
      bool first = true;                                // a channel is a byte stream: its operations complete in posting order
      for (size_t j = 0; j < i && first; ++j)
        first = !(ops[j].c == o.c && ops[j].send == o.send && ops[j].peer == o.peer && ops[j].done < ops[j].n);
      if (first) any |= progress(o);
      all &= (o.done == o.n);
    }
    if (all) return ncclSuccess;
    if (any) last = now();
    else {
      if (now() - last > kTimeout) {
        std::fprintf(stderr, "rccl_stub: no progress for %.0f s (a peer died or the collectives are mismatched)\n", kTimeout);
        return ncclSystemError;
      }
      sched_yield();
    }
  }
}

int post(Comm* c, bool send, int peer, const void* p, size_t n) {
  if (!c || peer < 0 || peer >= c->world || (!p && n)) return ncclInvalidArgument;
  Op o{c, send, peer, (unsigned char*)p, n, 0};
  if (g_group_depth > 0) {
    g_ops.push_back(o);
    return ncclSuccess;
  }
  std::vector<Op> one{o};
  return run(one);
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  static std::atomic<unsigned> counter{0};
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "/dpx_rccl_stub_%d_%u_%llx", (int)getpid(), counter.fetch_add(1),
                (unsigned long long)(now() * 1e6));
  return ncclSuccess;
}

int ncclCommInitRank(void** comm, int world, ncclUniqueId id, int rank) {
  if (!comm || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || id.internal[0] != '/') return ncclInvalidArgument;
  const size_t bytes = sizeof(Segment) + ((size_t)world * world - 1) * sizeof(Channel);
  bool creator = true;
  int fd = shm_open(id.internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) {
    creator = false;
    const double t0 = now();
    while ((fd = shm_open(id.internal, O_RDWR, 0600)) < 0)
      if (now() - t0 > kTimeout) return ncclSystemError;
  }
  if (creator && ftruncate(fd, (off_t)bytes) != 0) {
    close(fd);
    return ncclSystemError;
  }
  if (!creator) {                                       // wait until the creator has sized the segment
    struct stat st;
    const double t0 = now();
    while (fstat(fd, &st) == 0 && (size_t)st.st_size < bytes)
      if (now() - t0 > kTimeout) { close(fd); return ncclSystemError; }
  }
  void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return ncclSystemError;
  Segment* S = (Segment*)m;                             // (a fresh POSIX segment is zero-filled: heads, tails and counters start at 0)
  if (creator) {
    S->world = world;
    S->ready.store(1, std::memory_order_release);
  }
  const double t0 = now();
  while (!S->ready.load(std::memory_order_acquire))
    if (now() - t0 > kTimeout) return ncclSystemError;
  if (S->world != world) return ncclInvalidArgument;
  S->joined.fetch_add(1);
  while (S->joined.load() < world)                      // ncclCommInitRank is collective: returns once every rank has joined
    if (now() - t0 > kTimeout) return ncclSystemError;
    else sched_yield();
  Comm* c = new Comm{S, bytes, rank, world, {0}};
  std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
  *comm = c;
  return ncclSuccess;
}

int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclInvalidArgument;
  const bool last = c->seg->left.fetch_add(1) + 1 == c->world;
  munmap(c->seg, c->bytes);
  if (last) shm_unlink(c->name);
  delete c;
  return ncclSuccess;
}

int ncclGroupStart() {
  ++g_group_depth;
  return ncclSuccess;
}

int ncclGroupEnd() {
  if (g_group_depth <= 0) return ncclInvalidArgument;
  if (--g_group_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run(ops);
}

int ncclSend(const void* buf, size_t count, int dt, int peer, void* comm, void* /*stream*/) {
  return post((Comm*)comm, true, peer, buf, count * type_size(dt));
}

int ncclRecv(void* buf, size_t count, int dt, int peer, void* comm, void* /*stream*/) {
  return post((Comm*)comm, false, peer, buf, count * type_size(dt));
}

int ncclBroadcast(const void* send, void* recv, size_t count, int dt, int root, void* comm, void* /*stream*/) {
  Comm* c = (Comm*)comm;
  if (!c || root < 0 || root >= c->world) return ncclInvalidArgument;
  const size_t n = count * type_size(dt);
  std::vector<Op> ops;
  if (c->rank == root) {
    if (recv != send) std::memmove(recv, send, n);
    for (int r = 0; r < c->world; ++r)
      if (r != root) ops.push_back(Op{c, true, r, (unsigned char*)send, n, 0});
  } else {
    ops.push_back(Op{c, false, root, (unsigned char*)recv, n, 0});
  }
  return run(ops);
}

int ncclAllGather(const void* send, void* recv, size_t count, int dt, void* comm, void* /*stream*/) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclInvalidArgument;
  const size_t n = count * type_size(dt);
  unsigned char* own = (unsigned char*)recv + (size_t)c->rank * n;
  if (own != send) std::memmove(own, send, n);
  std::vector<Op> ops;
  for (int r = 0; r < c->world; ++r)
    if (r != c->rank) {
      ops.push_back(Op{c, true, r, own, n, 0});
      ops.push_back(Op{c, false, r, (unsigned char*)recv + (size_t)r * n, n, 0});
    }
  return run(ops);
}

const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "rccl_stub: system error / peer timeout";
    case ncclInvalidArgument: return "rccl_stub: invalid argument";
    default: return "rccl_stub: error";
  }
}

}  // extern "C"
