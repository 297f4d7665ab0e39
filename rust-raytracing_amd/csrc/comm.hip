// Multi-GPU side of the C ABI: one context per GPU (one process or thread each), the per-GPU accumulation buffers summed onto
// one rank with a single RCCL reduce over xGMI.  Replaces nothing in the reference, whose only exchange is the mpsc channel of
// finished pixels between worker threads (src/render.rs:185,192-201); the partition it stands in for is the static row split of
// src/render.rs:218-224, here a split of the sample range.
//
// RCCL is opened at run time (dlopen), so single-GPU users need no RCCL installation and the library has no link-time dependency
// on it; the four entry points used are resolved once.
#include <dlfcn.h>

#include <mutex>

#include "context.h"

namespace {
using namespace rti;

// The slice of RCCL's C API that is used (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclReduce, ncclCommDestroy, ncclGetErrorString).
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void* id) = nullptr;
  int (*CommInitRank)(void** comm, int n, rt_comm_id id, int rank) = nullptr;  // ncclUniqueId is a 128-byte struct passed by value
  int (*Reduce)(const void* send, void* recv, size_t count, int dtype, int op, int root, void* comm, hipStream_t stream) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t values of rccl.h

std::mutex g_rccl_mutex;
Rccl g_rccl;

int load_rccl(const Rccl** out) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!g_rccl.lib) {
    void* h = nullptr;
    std::string why;
    // RTAMD_RCCL_LIB names the library to open instead of the usual ones (a host with RCCL somewhere else; tests of the
    // "RCCL is not installed" path).
    const char* forced = getenv("RTAMD_RCCL_LIB");
    const char* usual[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // A process that already maps an RCCL (torch.distributed's, say) gets that one: RTLD_NOLOAD finds a library that is loaded
    // and loads nothing, so the process never ends up with two RCCL instances side by side.
    if (!forced)
      for (const char* name : usual) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (h) break;
      }
    if (!h) {
      if (forced) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!h) { const char* e = dlerror(); why = e ? e : "not found"; }
      } else {
        for (const char* name : usual) {
          h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
          if (h) break;
          const char* e = dlerror();  // one call: it hands the message over and clears it
          if (why.empty()) why = e ? e : "librccl.so.1 not found";
        }
      }
    }
    if (!h) return fail(RT_E_UNSUPPORTED, "RCCL is not available: " + (why.empty() ? std::string("librccl.so.1 not found") : why));
    Rccl r;
    r.lib = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.Reduce = reinterpret_cast<decltype(r.Reduce)>(dlsym(h, "ncclReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.Reduce || !r.CommDestroy || !r.GetErrorString) {
      dlclose(h);
      return fail(RT_E_UNSUPPORTED, "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclReduce / ncclCommDestroy / ncclGetErrorString");
    }
    g_rccl = r;
  }
  *out = &g_rccl;
  return RT_OK;
}

int rccl_fail(const Rccl* r, const char* what, int code) {
  return fail(RT_E_HIP, std::string(what) + ": " + r->GetErrorString(code));
}

// Pixel slots are four values wide (the fourth unused): the reduce carries the three colour sums only, W*H*3 values.
template <class R> __global__ void k_pack3(const R* accum4, R* packed3, uint32_t n_pixels) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pixels) return;
  packed3[(size_t)i * 3u + 0] = accum4[(size_t)i * 4u + 0];
  packed3[(size_t)i * 3u + 1] = accum4[(size_t)i * 4u + 1];
  packed3[(size_t)i * 3u + 2] = accum4[(size_t)i * 4u + 2];
}
template <class R> __global__ void k_unpack3(const R* packed3, R* accum4, uint32_t n_pixels) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pixels) return;
  accum4[(size_t)i * 4u + 0] = packed3[(size_t)i * 3u + 0];
  accum4[(size_t)i * 4u + 1] = packed3[(size_t)i * 3u + 1];
  accum4[(size_t)i * 4u + 2] = packed3[(size_t)i * 3u + 2];
}

template <class R> int reduce_impl(rt_context* c, const Rccl* r, R* accum, size_t n_pixels, int root, int dtype) {
  const size_t bytes = n_pixels * 3 * sizeof(R);
  if (c->comm_buf_bytes < bytes) {
    if (c->comm_buf) (void)hipFree(c->comm_buf);
    c->comm_buf = nullptr; c->comm_buf_bytes = 0;
    HIP_TRY(hipMalloc(&c->comm_buf, bytes));
    c->comm_buf_bytes = bytes;
  }
  R* packed = reinterpret_cast<R*>(c->comm_buf);
  const uint32_t grid = (uint32_t)((n_pixels + kBlock - 1) / kBlock);
  hipLaunchKernelGGL((k_pack3<R>), dim3(grid), dim3(kBlock), 0, c->stream, accum, packed, (uint32_t)n_pixels);
  HIP_TRY(hipGetLastError());
  int rc = r->Reduce(packed, packed, n_pixels * 3, dtype, kNcclSum, root, c->comm, c->stream);
  if (rc != 0) return rccl_fail(r, "ncclReduce", rc);
  if (c->comm_rank == root) {
    hipLaunchKernelGGL((k_unpack3<R>), dim3(grid), dim3(kBlock), 0, c->stream, packed, accum, (uint32_t)n_pixels);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RT_OK;
}

}  // namespace

extern "C" {

int rt_comm_unique_id(rt_comm_id* id) {
  if (!id) return fail(RT_E_INVALID, "id is NULL");
  const Rccl* r = nullptr;
  if (int rc = load_rccl(&r)) return rc;
  static_assert(sizeof(rt_comm_id) == 128, "ncclUniqueId is 128 bytes");
  int rc = r->GetUniqueId(id);
  if (rc != 0) return rccl_fail(r, "ncclGetUniqueId", rc);
  return RT_OK;
}

int rt_comm_init(rt_context* c, const rt_comm_id* id, int n_ranks, int rank) {
  if (!c || !id) return fail(RT_E_INVALID, "context or id is NULL");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RT_E_INVALID, "rank out of range");
  if (c->comm) return fail(RT_E_INVALID, "this context already belongs to a communicator (rt_comm_destroy first)");
  const Rccl* r = nullptr;
  if (int rc = load_rccl(&r)) return rc;
  HIP_TRY(hipSetDevice(c->device));
  void* comm = nullptr;
  int rc = r->CommInitRank(&comm, n_ranks, *id, rank);
  if (rc != 0) return rccl_fail(r, "ncclCommInitRank", rc);
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_size = n_ranks;
  return RT_OK;
}

int rt_comm_destroy(rt_context* c) {
  if (!c) return fail(RT_E_INVALID, "context is NULL");
  if (!c->comm) return RT_OK;
  const Rccl* r = nullptr;
  if (int rc = load_rccl(&r)) return rc;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  int rc = r->CommDestroy(c->comm);
  c->comm = nullptr;
  c->comm_rank = 0;
  c->comm_size = 1;
  if (c->comm_buf) (void)hipFree(c->comm_buf);
  c->comm_buf = nullptr; c->comm_buf_bytes = 0;
  if (rc != 0) return rccl_fail(r, "ncclCommDestroy", rc);
  return RT_OK;
}

int rt_reduce_accum(rt_context* c, void* accum_dev, uint32_t width, uint32_t height, int precision, int root) {
  if (!c) return fail(RT_E_INVALID, "context is NULL");
  precision &= ~RT_F64_OWN_TREE;
  if (precision != RT_PRECISION_F32 && precision != RT_PRECISION_F64) return fail(RT_E_INVALID, "unknown precision");
  if (!c->comm) return fail(RT_E_INVALID, "rt_reduce_accum before rt_comm_init");
  if (root < 0 || root >= c->comm_size) return fail(RT_E_INVALID, "root out of range");
  const size_t n = (size_t)width * height;
  if (n == 0 || n > 0x7FFFFFFFull) return fail(RT_E_INVALID, "bad image size");
  const size_t need = n * 4 * (precision == RT_PRECISION_F64 ? sizeof(double) : sizeof(float));
  if (!accum_dev) {
    if (!c->accum || c->accum_bytes < need) return fail(RT_E_INVALID, "the context holds no accumulation buffer of this size");
    accum_dev = c->accum;
  }
  const Rccl* r = nullptr;
  if (int rc = load_rccl(&r)) return rc;
  HIP_TRY(hipSetDevice(c->device));
  return precision == RT_PRECISION_F64 ? reduce_impl<double>(c, r, reinterpret_cast<double*>(accum_dev), n, root, kNcclFloat64)
                                       : reduce_impl<float>(c, r, reinterpret_cast<float*>(accum_dev), n, root, kNcclFloat32);
}

}  // extern "C"
