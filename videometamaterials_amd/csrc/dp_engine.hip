// Data-parallel engine behind the C ABI (include/vmm_dp.h): bucketed gradient all-reduce over RCCL on ONE side stream, ordered against the
// compute stream by events only -- the host thread never waits while the backward is being enqueued.
//
// What it stands in for: DistributedDataParallel's reducer as the reference reaches it through Accelerate (main.py:31-34, vddp.py:1449,
// 1629) and the two collectives of sharded sampling (vddp.py:1506-1532, 1838-1842).  RCCL is bound with dlopen so that libvmm_hip.so
// has no link-time dependency on it (single-GPU users never load it, and a host that ships its own RCCL keeps one copy per process).
//
// xGMI is point to point (7 links per GPU): RCCL's ring / tree kernels for a 150 MB gradient buffer are link-bound, so the buckets are few
// and large (>= 16 MB by default: DataParallelTrainer.bucket_floats = 4 M floats; chosen by the host from the plan's "tail is final" marks) and every one is reduced in place, fp32, sum; the mean's
// 1 / world rides in the optimiser launch.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <vector>

#include "../../include/vmm_dp.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

template <typename F>
bool bind(void* h, const char* name, F& fn) {
  fn = reinterpret_cast<F>(dlsym(h, name));
  return fn != nullptr;
}

int load_api(const char* path, RcclApi& api, char* err, size_t nerr) {
  static_assert(sizeof(ncclUniqueId) == VMM_DP_UNIQUE_ID_BYTES, "vmm_dp.h states the id size");
  const char* p = (path && path[0]) ? path : "librccl.so.1";
  api.handle = dlopen(p, RTLD_NOW | RTLD_LOCAL);
  if (!api.handle) {
    snprintf(err, nerr, "dlopen(%s): %s", p, dlerror());
    return -2;
  }
  const bool ok = bind(api.handle, "ncclGetVersion", api.GetVersion) && bind(api.handle, "ncclGetUniqueId", api.GetUniqueId) &&
                  bind(api.handle, "ncclCommInitRank", api.CommInitRank) && bind(api.handle, "ncclCommDestroy", api.CommDestroy) &&
                  bind(api.handle, "ncclAllReduce", api.AllReduce) && bind(api.handle, "ncclBroadcast", api.Broadcast) &&
                  bind(api.handle, "ncclAllGather", api.AllGather) && bind(api.handle, "ncclGetErrorString", api.GetErrorString);
  if (!ok) {
    snprintf(err, nerr, "%s lacks an RCCL entry point: %s", p, dlerror());
    dlclose(api.handle);
    api.handle = nullptr;
    return -2;
  }
  return 0;
}

struct Bucket {
  float* ptr = nullptr;
  int64_t count = 0;
  hipEvent_t ready = nullptr;  // compute stream: the bucket's gradients are final
  hipEvent_t t0 = nullptr;     // side stream: reduction starts (timing mode)
  hipEvent_t done = nullptr;   // side stream: reduction complete
  bool pending = false;        // issued, not yet waited for by the compute stream
  bool stamped = false;        // t0 / done belong to the current window
};

}  // namespace

struct vmm_dp_engine {
  RcclApi api;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0, version = 0;
  hipStream_t side = nullptr;
  std::vector<Bucket> buckets;
  hipEvent_t win[2] = {nullptr, nullptr};
  bool win_set[2] = {false, false};
  bool timing = false;
  char err[320] = {0};
};

namespace {

int fail_hip(vmm_dp_engine* e, hipError_t r, const char* what) {
  snprintf(e->err, sizeof e->err, "%s: %s", what, hipGetErrorString(r));
  return (int)r;
}
int fail_nccl(vmm_dp_engine* e, ncclResult_t r, const char* what) {
  snprintf(e->err, sizeof e->err, "%s: %s", what, e->api.GetErrorString ? e->api.GetErrorString(r) : "?");
  return -(1000 + (int)r);
}
#define DP_HIP(call)                                         \
  do {                                                       \
    const hipError_t r_ = (call);                            \
    if (r_ != hipSuccess) return fail_hip(e, r_, #call);     \
  } while (0)
#define DP_NCCL(call)                                        \
  do {                                                       \
    const ncclResult_t r_ = (call);                          \
    if (r_ != ncclSuccess) return fail_nccl(e, r_, #call);   \
  } while (0)

void drop_buckets(vmm_dp_engine* e) {
  for (Bucket& b : e->buckets) {
    if (b.ready) (void)hipEventDestroy(b.ready);
    if (b.t0) (void)hipEventDestroy(b.t0);
    if (b.done) (void)hipEventDestroy(b.done);
  }
  e->buckets.clear();
}

bool dtype_of(int32_t dtype, ncclDataType_t& t) {
  switch (dtype) {
    case 0: t = ncclFloat32; return true;
    case 1: t = ncclFloat64; return true;
    case 2: t = ncclInt32; return true;
    case 3: t = ncclInt64; return true;
  }
  return false;
}

}  // namespace

extern "C" int vmm_dp_get_unique_id(const char* rccl_path, void* id_out) {
  if (!id_out) return -1;
  RcclApi api;
  char err[320];
  if (int r = load_api(rccl_path, api, err, sizeof err)) {
    fprintf(stderr, "vmm_dp_get_unique_id: %s\n", err);
    return r;
  }
  ncclUniqueId id;
  const ncclResult_t r = api.GetUniqueId(&id);
  if (r == ncclSuccess) memcpy(id_out, &id, sizeof id);
  dlclose(api.handle);
  return r == ncclSuccess ? 0 : -(1000 + (int)r);
}

extern "C" int vmm_dp_init(vmm_dp_engine** out, const char* rccl_path, int32_t rank, int32_t world, const void* unique_id, int32_t device) {
  if (!out || !unique_id || world < 1 || rank < 0 || rank >= world) return -1;
  *out = nullptr;
  vmm_dp_engine* e = new (std::nothrow) vmm_dp_engine();
  if (!e) return -1;
  auto bail = [&](int code) {
    fprintf(stderr, "vmm_dp_init: %s\n", e->err);
    if (e->side) (void)hipStreamDestroy(e->side);
    for (hipEvent_t w : e->win)
      if (w) (void)hipEventDestroy(w);
    if (e->api.handle) dlclose(e->api.handle);
    delete e;
    return code;
  };
  if (int r = load_api(rccl_path, e->api, e->err, sizeof e->err)) return bail(r);
  e->rank = rank;
  e->world = world;
  e->device = device;
  hipError_t h = hipSetDevice(device);
  if (h != hipSuccess) return bail(fail_hip(e, h, "hipSetDevice"));
  (void)e->api.GetVersion(&e->version);
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  const ncclResult_t n = e->api.CommInitRank(&e->comm, world, id, rank);
  if (n != ncclSuccess) return bail(fail_nccl(e, n, "ncclCommInitRank"));
  h = hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking);
  if (h != hipSuccess) return bail(fail_hip(e, h, "hipStreamCreateWithFlags"));
  for (int k = 0; k < 2; ++k) {
    h = hipEventCreate(&e->win[k]);
    if (h != hipSuccess) return bail(fail_hip(e, h, "hipEventCreate"));
  }
  *out = e;
  return 0;
}

extern "C" int vmm_dp_register_buckets(vmm_dp_engine* e, float* const* ptrs, const int64_t* counts, int32_t n) {
  if (!e || n < 0 || (n > 0 && (!ptrs || !counts))) return -1;
  for (const Bucket& b : e->buckets)
    if (b.pending) {
      snprintf(e->err, sizeof e->err, "register_buckets while reductions are pending (call vmm_dp_wait_all first)");
      return -1;
    }
  drop_buckets(e);
  e->buckets.resize(n);
  for (int i = 0; i < n; ++i) {
    Bucket& b = e->buckets[i];
    if (!ptrs[i] || counts[i] <= 0) return -1;
    b.ptr = ptrs[i];
    b.count = counts[i];
    DP_HIP(hipEventCreateWithFlags(&b.ready, hipEventDisableTiming));
    DP_HIP(hipEventCreate(&b.t0));
    DP_HIP(hipEventCreate(&b.done));
  }
  return 0;
}

extern "C" int vmm_dp_allreduce_bucket_async(vmm_dp_engine* e, int32_t i, vmm_dp_stream_t compute_stream) {
  if (!e || i < 0 || i >= (int)e->buckets.size()) return -1;
  Bucket& b = e->buckets[i];
  if (b.pending) {
    snprintf(e->err, sizeof e->err, "bucket %d issued twice without vmm_dp_wait_all", i);
    return -1;
  }
  hipStream_t cs = static_cast<hipStream_t>(compute_stream);
  DP_HIP(hipEventRecord(b.ready, cs));
  DP_HIP(hipStreamWaitEvent(e->side, b.ready, 0));
  if (e->timing) DP_HIP(hipEventRecord(b.t0, e->side));
  DP_NCCL(e->api.AllReduce(b.ptr, b.ptr, (size_t)b.count, ncclFloat32, ncclSum, e->comm, e->side));
  DP_HIP(hipEventRecord(b.done, e->side));
  b.pending = true;
  b.stamped = e->timing;
  return 0;
}

extern "C" int vmm_dp_wait_all(vmm_dp_engine* e, vmm_dp_stream_t compute_stream) {
  if (!e) return -1;
  hipStream_t cs = static_cast<hipStream_t>(compute_stream);
  for (Bucket& b : e->buckets)
    if (b.pending) {
      DP_HIP(hipStreamWaitEvent(cs, b.done, 0));
      b.pending = false;
    }
  return 0;
}

extern "C" int vmm_dp_set_timing(vmm_dp_engine* e, int32_t on) {
  if (!e) return -1;
  e->timing = on != 0;
  return 0;
}

extern "C" int vmm_dp_window_mark(vmm_dp_engine* e, int32_t which, vmm_dp_stream_t compute_stream) {
  if (!e || which < 0 || which > 1) return -1;
  if (which == 0) {
    for (Bucket& b : e->buckets) b.stamped = false;
    e->win_set[1] = false;
  }
  DP_HIP(hipEventRecord(e->win[which], static_cast<hipStream_t>(compute_stream)));
  e->win_set[which] = true;
  return 0;
}

extern "C" int vmm_dp_timing(vmm_dp_engine* e, float* out3) {
  if (!e || !out3) return -1;
  out3[0] = out3[1] = out3[2] = 0.f;
  if (!e->win_set[0] || !e->win_set[1]) {
    snprintf(e->err, sizeof e->err, "vmm_dp_timing without both window marks");
    return -1;
  }
  float win = 0.f;
  DP_HIP(hipEventElapsedTime(&win, e->win[0], e->win[1]));
  float busy = 0.f, inside = 0.f;
  for (const Bucket& b : e->buckets) {
    if (!b.stamped) continue;
    float s0 = 0.f, s1 = 0.f;
    DP_HIP(hipEventElapsedTime(&s0, e->win[0], b.t0));
    DP_HIP(hipEventElapsedTime(&s1, e->win[0], b.done));
    busy += s1 - s0;
    const float lo = s0 > 0.f ? s0 : 0.f, hi = s1 < win ? s1 : win;
    if (hi > lo) inside += hi - lo;
  }
  out3[0] = busy;
  out3[1] = inside;
  out3[2] = win;
  return 0;
}

// per bucket of the registered list: out[2 i] = ms from the window's opening mark to the start of bucket i's reduction on the side stream (its slice
// was final and the stream free), out[2 i + 1] = ms to its completion; -1 for buckets not reduced since the window opened.  After a synchronize.
extern "C" int vmm_dp_bucket_timing(vmm_dp_engine* e, float* out, int32_t n) {
  if (!e || !out || n < 0) return -1;
  if (!e->win_set[0]) {
    snprintf(e->err, sizeof e->err, "vmm_dp_bucket_timing without the opening window mark");
    return -1;
  }
  for (int i = 0; i < n; ++i) {
    out[2 * i] = out[2 * i + 1] = -1.f;
    if (i >= (int)e->buckets.size() || !e->buckets[i].stamped) continue;
    DP_HIP(hipEventElapsedTime(&out[2 * i], e->win[0], e->buckets[i].t0));
    DP_HIP(hipEventElapsedTime(&out[2 * i + 1], e->win[0], e->buckets[i].done));
  }
  return 0;
}

extern "C" int vmm_dp_allreduce(vmm_dp_engine* e, void* buf, int64_t count, int32_t dtype, int32_t op, vmm_dp_stream_t stream) {
  ncclDataType_t t;
  if (!e || !buf || count <= 0 || !dtype_of(dtype, t) || op < 0 || op > 2) return -1;
  const ncclRedOp_t o = op == 0 ? ncclSum : op == 1 ? ncclMax : ncclMin;
  DP_NCCL(e->api.AllReduce(buf, buf, (size_t)count, t, o, e->comm, static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int vmm_dp_broadcast(vmm_dp_engine* e, void* buf, int64_t bytes, int32_t root, vmm_dp_stream_t stream) {
  if (!e || !buf || bytes <= 0 || root < 0 || root >= e->world) return -1;
  DP_NCCL(e->api.Broadcast(buf, buf, (size_t)bytes, ncclInt8, root, e->comm, static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int vmm_dp_all_gather(vmm_dp_engine* e, const void* send, void* recv, int64_t bytes_per_rank, vmm_dp_stream_t stream) {
  if (!e || !send || !recv || bytes_per_rank <= 0) return -1;
  DP_NCCL(e->api.AllGather(send, recv, (size_t)bytes_per_rank, ncclInt8, e->comm, static_cast<hipStream_t>(stream)));
  return 0;
}

extern "C" int vmm_dp_rank(const vmm_dp_engine* e) { return e ? e->rank : -1; }
extern "C" int vmm_dp_world(const vmm_dp_engine* e) { return e ? e->world : -1; }
extern "C" int vmm_dp_rccl_version(const vmm_dp_engine* e) { return e ? e->version : 0; }
extern "C" const char* vmm_dp_last_error(const vmm_dp_engine* e) { return e ? e->err : "null engine"; }

extern "C" int vmm_dp_finalize(vmm_dp_engine* e) {
  if (!e) return -1;
  int rc = 0;
  if (e->side) {
    const hipError_t h = hipStreamSynchronize(e->side);
    if (h != hipSuccess) rc = (int)h;
  }
  drop_buckets(e);
  if (e->comm) {
    const ncclResult_t n = e->api.CommDestroy(e->comm);
    if (n != ncclSuccess && rc == 0) rc = -(1000 + (int)n);
  }
  if (e->side) (void)hipStreamDestroy(e->side);
  for (hipEvent_t w : e->win)
    if (w) (void)hipEventDestroy(w);
  if (e->api.handle) dlclose(e->api.handle);
  delete e;
  return rc;
}
