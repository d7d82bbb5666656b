// The episode's ONE collective through the C ABI: every rank's packed class-code rows
// ([capacity][280] fp32, layout of sylph_amd/distributed.py) -> all ranks' rows in rank order on every rank, by one in-place
// ncclAllGather on the context's stream (RCCL over xGMI).  Replaces the all_gather_object of pickled dicts in
// MetaFCOSRunner._gather_class_code (sylph/runner/meta_fcos_runner.py:381-439) for hosts that do not go through torch.distributed.
//
// librccl is NOT a link-time dependency of libsylph_hip.so: the five RCCL entry points used here are resolved at first use with
// dlopen (the copy already mapped into the process -- e.g. the one PyTorch ships -- wins, so one process never runs two RCCLs).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>
#include <string>

#include "../../include/sylph_hip.h"

hipStream_t sylph_internal_stream(sylph_ctx* c);  // api_core.hip
int sylph_internal_fail(const std::string& m);     // sets sylph_last_error(), returns 1

namespace {

constexpr int CODE_ROW = 280;         // floats per packed class-code row (sylph_amd.distributed.ROW)
constexpr int UNIQUE_ID_BYTES = 128;  // NCCL_UNIQUE_ID_BYTES

// the RCCL ABI subset used here (rccl.h): ncclResult_t is an int enum with ncclSuccess = 0, ncclFloat = 7
struct NcclUniqueId { char internal[UNIQUE_ID_BYTES]; };
typedef int (*fn_get_unique_id)(NcclUniqueId*);
typedef int (*fn_comm_init_rank)(void** comm, int nranks, NcclUniqueId id, int rank);
typedef int (*fn_comm_destroy)(void* comm);
typedef int (*fn_comm_user_rank)(const void* comm, int* rank);
typedef int (*fn_comm_count)(const void* comm, int* count);
typedef int (*fn_all_gather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t s);
typedef const char* (*fn_get_error_string)(int);

struct Rccl {
  void* h = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_comm_user_rank comm_user_rank = nullptr;
  fn_comm_count comm_count = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_get_error_string get_error_string = nullptr;
  std::string err;
};

void rccl_resolve(Rccl& R) {
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names)  // a copy that is already mapped (RTLD_NOLOAD) first
    if ((R.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!R.h)
    for (const char* n : names)
      if ((R.h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!R.h) {
    const char* e = dlerror();  // ONE call: dlerror() clears the message it returns
    R.err = std::string("librccl.so not found: ") + (e ? e : "(no dlerror message)");
    return;
  }
#define SYM(field, name)                                                      \
  R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.h, name));            \
  if (!R.field) { R.err = std::string("librccl: missing symbol ") + name; R.h = nullptr; return; }
  SYM(get_unique_id, "ncclGetUniqueId")
  SYM(comm_init_rank, "ncclCommInitRank")
  SYM(comm_destroy, "ncclCommDestroy")
  SYM(comm_user_rank, "ncclCommUserRank")
  SYM(comm_count, "ncclCommCount")
  SYM(all_gather, "ncclAllGather")
  SYM(get_error_string, "ncclGetErrorString")
#undef SYM
}

// resolved once per process, whichever thread asks first (std::call_once: concurrent first calls block until it is done)
Rccl* rccl() {
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, [] { rccl_resolve(R); });
  return &R;
}

int nccl_fail(Rccl* R, const char* what, int rc) {
  return sylph_internal_fail(std::string(what) + ": " + (R->get_error_string ? R->get_error_string(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}

}  // namespace

extern "C" {

int sylph_comm_unique_id(char* id_out) {
  Rccl* R = rccl();
  if (!R->h) return sylph_internal_fail(R->err);
  if (!id_out) return sylph_internal_fail("NULL argument");
  NcclUniqueId id;
  const int rc = R->get_unique_id(&id);
  if (rc != 0) return nccl_fail(R, "ncclGetUniqueId", rc);
  memcpy(id_out, id.internal, UNIQUE_ID_BYTES);
  return 0;
}

int sylph_comm_init_rank(sylph_ctx* c, const char* id_bytes, int nranks, int rank, void** comm_out) {
  Rccl* R = rccl();
  if (!R->h) return sylph_internal_fail(R->err);
  if (!c || !id_bytes || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) return sylph_internal_fail("sylph_comm_init_rank: bad argument");
  (void)sylph_internal_stream(c);  // makes the context's device current
  NcclUniqueId id;
  memcpy(id.internal, id_bytes, UNIQUE_ID_BYTES);
  void* comm = nullptr;
  const int rc = R->comm_init_rank(&comm, nranks, id, rank);
  if (rc != 0) return nccl_fail(R, "ncclCommInitRank", rc);
  *comm_out = comm;
  return 0;
}

int sylph_comm_destroy(void* comm) {
  Rccl* R = rccl();
  if (!R->h) return sylph_internal_fail(R->err);
  if (!comm) return 0;
  const int rc = R->comm_destroy(comm);
  return rc == 0 ? 0 : nccl_fail(R, "ncclCommDestroy", rc);
}

int sylph_allgather_codes(sylph_ctx* c, void* comm, const float* local_dev, int n_local, int capacity, float* out_dev) {
  Rccl* R = rccl();
  if (!R->h) return sylph_internal_fail(R->err);
  if (!c || !comm || !out_dev || (n_local > 0 && !local_dev)) return sylph_internal_fail("sylph_allgather_codes: NULL argument");
  if (capacity < 1 || n_local < 0 || n_local > capacity)
    return sylph_internal_fail("sylph_allgather_codes: " + std::to_string(n_local) + " rows do not fit the gather block of " + std::to_string(capacity));
  int rank = 0, world = 0, rc;
  if ((rc = R->comm_user_rank(comm, &rank)) != 0) return nccl_fail(R, "ncclCommUserRank", rc);
  if ((rc = R->comm_count(comm, &world)) != 0) return nccl_fail(R, "ncclCommCount", rc);
  hipStream_t s = sylph_internal_stream(c);
  // this rank's block of the output: its rows, then zero rows (valid = 0); the gather runs in place on it
  const size_t block = (size_t)capacity * CODE_ROW;
  float* mine = out_dev + (size_t)rank * block;
  hipError_t e = hipSuccess;
  if (n_local > 0 && mine != local_dev) e = hipMemcpyAsync(mine, local_dev, (size_t)n_local * CODE_ROW * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess && n_local < capacity)
    e = hipMemsetAsync(mine + (size_t)n_local * CODE_ROW, 0, (size_t)(capacity - n_local) * CODE_ROW * sizeof(float), s);
  if (e != hipSuccess) return sylph_internal_fail(std::string("sylph_allgather_codes: ") + hipGetErrorString(e));
  rc = R->all_gather(mine, out_dev, block, /*ncclFloat*/ 7, comm, s);
  return rc == 0 ? 0 : nccl_fail(R, "ncclAllGather", rc);
}

}  // extern "C"
