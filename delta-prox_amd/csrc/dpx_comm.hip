// RCCL behind the C ABI: the one-time broadcast of shared constants (OTF / denominator tables, denoiser
// weights, sampling masks, schedules), the scatter of a batch held by one rank and the final all-gather of the per-rank results.
// Independent images never exchange data inside the iteration, so these three collectives are all the communication there is.
//
// librccl is bound lazily (dlopen at dpx_comm_unique_id / dpx_comm_init): single-GPU users and the CPU-side build check never load it.
// A communicator is tied to the HIP device that is current when dpx_comm_init is called (one process per GPU).
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

#include "dpx_common.h"

namespace {

typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
enum { RCCL_INT8 = 0 };                                    // ncclInt8 / ncclChar

struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(rcclUniqueId*) = nullptr;
  int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

// the collective library: dpx_comm_use_library(path) or, as its initial value, the environment variable DPX_RCCL_LIB name a shared
// object that is tried before the system's librccl (any library with the NCCL / RCCL C API: a site build of RCCL, or the
// shared-memory stand-in of tests/emul that lets the multi-rank branches below run between CPU processes)
char g_lib_path[1024] = "";
bool g_lib_tried = false;

Rccl* rccl() {
  static Rccl R;
  if (!g_lib_tried) {
    g_lib_tried = true;
    if (!g_lib_path[0] && getenv("DPX_RCCL_LIB")) snprintf(g_lib_path, sizeof(g_lib_path), "%s", getenv("DPX_RCCL_LIB"));
    if (g_lib_path[0]) R.so = dlopen(g_lib_path, RTLD_NOW | RTLD_GLOBAL);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (R.so || g_lib_path[0]) break;                  // (a named library that does not load is an error, not a reason to fall back)
      R.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (R.so) {
#define DPX_BIND(field, sym) *(void**)(&R.field) = dlsym(R.so, sym)
      DPX_BIND(GetUniqueId, "ncclGetUniqueId");
      DPX_BIND(CommInitRank, "ncclCommInitRank");
      DPX_BIND(CommDestroy, "ncclCommDestroy");
      DPX_BIND(Broadcast, "ncclBroadcast");
      DPX_BIND(AllGather, "ncclAllGather");
      DPX_BIND(Send, "ncclSend");
      DPX_BIND(Recv, "ncclRecv");
      DPX_BIND(GroupStart, "ncclGroupStart");
      DPX_BIND(GroupEnd, "ncclGroupEnd");
      DPX_BIND(GetErrorString, "ncclGetErrorString");
#undef DPX_BIND
    }
  }
  const bool ok = R.so && R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.Broadcast && R.AllGather && R.Send && R.Recv && R.GroupStart &&
                  R.GroupEnd;
  return ok ? &R : nullptr;
}

struct Comm {
  rcclComm_t c;
  int rank, world;
};

int fail(const char* what, int rc) {
  Rccl* R = rccl();
  dpx::set_error("%s: RCCL error %d (%s)", what, rc, (R && R->GetErrorString) ? R->GetErrorString(rc) : "?");
  return DPX_ERR_LAUNCH;
}

}  // namespace

extern "C" int dpx_comm_use_library(const char* path) {
  DPX_REQUIRE(path && strlen(path) < sizeof(g_lib_path), "dpx_comm_use_library: bad path");
  DPX_REQUIRE(!g_lib_tried || !strcmp(path, g_lib_path), "dpx_comm_use_library: the collective library is already loaded (%s)",
              g_lib_path[0] ? g_lib_path : "librccl");
  snprintf(g_lib_path, sizeof(g_lib_path), "%s", path);
  return DPX_OK;
}

extern "C" int dpx_comm_unique_id(void* out128) {
  DPX_REQUIRE(out128, "dpx_comm_unique_id: null pointer");
  Rccl* R = rccl();
  DPX_REQUIRE(R, "dpx_comm_unique_id: librccl.so could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing");
  rcclUniqueId id;
  const int rc = R->GetUniqueId(&id);
  if (rc) return fail("dpx_comm_unique_id", rc);
  std::memcpy(out128, &id, sizeof(id));
  return DPX_OK;
}

extern "C" int dpx_comm_init(void** comm, const void* id128, int rank, int world) {
  DPX_REQUIRE(comm && id128 && world >= 1 && rank >= 0 && rank < world, "dpx_comm_init: bad arguments (rank %d of %d)", rank, world);
  Rccl* R = rccl();
  DPX_REQUIRE(R, "dpx_comm_init: librccl.so could not be loaded");
  rcclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  Comm* C = new Comm{nullptr, rank, world};
  const int rc = R->CommInitRank(&C->c, world, id, rank);
  if (rc) {
    delete C;
    return fail("dpx_comm_init", rc);
  }
  *comm = C;
  return DPX_OK;
}

extern "C" int dpx_comm_destroy(void* comm) {
  if (!comm) return DPX_OK;
  Comm* C = (Comm*)comm;
  Rccl* R = rccl();
  const int rc = R ? R->CommDestroy(C->c) : 0;
  delete C;
  return rc ? fail("dpx_comm_destroy", rc) : DPX_OK;
}

// in place: every rank passes its own buffer; afterwards all hold root's bytes
extern "C" int dpx_comm_broadcast(void* comm, void* buf, size_t bytes, int root, dpx_stream_t stream) {
  DPX_REQUIRE(comm && (buf || !bytes), "dpx_comm_broadcast: null pointer");
  Comm* C = (Comm*)comm;
  DPX_REQUIRE(root >= 0 && root < C->world, "dpx_comm_broadcast: root %d of %d", root, C->world);
  if (!bytes) return DPX_OK;
  const int rc = rccl()->Broadcast(buf, buf, bytes, RCCL_INT8, root, C->c, (hipStream_t)stream);
  return rc ? fail("dpx_comm_broadcast", rc) : DPX_OK;
}

// recv = [world][bytes_per_rank]; send may alias recv + rank * bytes_per_rank.
// xGMI is a point-to-point mesh (7 links per GPU): the gather is issued as world - 1 direct peer sends + receives in ONE group --
// every link carries one slice each way at the same time -- instead of the library's ring, whose every step is bound by a single
// link.  DPX_COMM_ALLGATHER=ring keeps ncclAllGather (A/B).
extern "C" int dpx_comm_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, dpx_stream_t stream) {
  DPX_REQUIRE(comm && ((send && recv) || !bytes_per_rank), "dpx_comm_allgather: null pointer");
  Comm* C = (Comm*)comm;
  if (!bytes_per_rank) return DPX_OK;
  Rccl* R = rccl();
  const bool ring = dpx::tune(dpx::TUNE_COMM_ALLGATHER_RING) != 0;
  if (ring || C->world == 1) {
    const int rc = R->AllGather(send, recv, bytes_per_rank, RCCL_INT8, C->c, (hipStream_t)stream);
    return rc ? fail("dpx_comm_allgather", rc) : DPX_OK;
  }
  char* own = (char*)recv + (size_t)C->rank * bytes_per_rank;
  if ((const void*)own != send && hipMemcpyAsync(own, send, bytes_per_rank, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
    dpx::set_error("dpx_comm_allgather: local copy failed");
    return DPX_ERR_LAUNCH;
  }
  int rc = R->GroupStart();
  if (rc) return fail("dpx_comm_allgather", rc);
  for (int d = 1; d < C->world && !rc; ++d) {                 // peer at distance d: every rank talks to a different peer per step
    const int to = (C->rank + d) % C->world, from = (C->rank - d + C->world) % C->world;
    rc = R->Send(send, bytes_per_rank, RCCL_INT8, to, C->c, (hipStream_t)stream);
    if (!rc) rc = R->Recv((char*)recv + (size_t)from * bytes_per_rank, bytes_per_rank, RCCL_INT8, from, C->c, (hipStream_t)stream);
  }
  const int rc2 = R->GroupEnd();
  if (rc || rc2) return fail("dpx_comm_allgather", rc ? rc : rc2);
  return DPX_OK;
}

// root holds send = [world][bytes_per_rank]; every rank receives its slice: direct peer sends over the xGMI mesh (no ring)
extern "C" int dpx_comm_scatter(void* comm, const void* send, void* recv, size_t bytes_per_rank, int root, dpx_stream_t stream) {
  DPX_REQUIRE(comm && (recv || !bytes_per_rank), "dpx_comm_scatter: null pointer");
  Comm* C = (Comm*)comm;
  DPX_REQUIRE(root >= 0 && root < C->world && (C->rank != root || send || !bytes_per_rank), "dpx_comm_scatter: bad root / send buffer");
  if (!bytes_per_rank) return DPX_OK;
  Rccl* R = rccl();
  int rc = R->GroupStart();
  if (rc) return fail("dpx_comm_scatter", rc);
  if (C->rank == root)
    for (int r = 0; r < C->world && !rc; ++r) rc = R->Send((const char*)send + (size_t)r * bytes_per_rank, bytes_per_rank, RCCL_INT8, r, C->c, (hipStream_t)stream);
  if (!rc) rc = R->Recv(recv, bytes_per_rank, RCCL_INT8, root, C->c, (hipStream_t)stream);
  const int rc2 = R->GroupEnd();
  if (rc || rc2) return fail("dpx_comm_scatter", rc ? rc : rc2);
  return DPX_OK;
}

extern "C" int dpx_comm_rank(void* comm) { return comm ? ((Comm*)comm)->rank : -1; }
extern "C" int dpx_comm_world(void* comm) { return comm ? ((Comm*)comm)->world : 0; }
