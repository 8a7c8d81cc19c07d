// brisk_comm.hip - the result gather of the multi-GPU batch path (BASELINE config 3) behind the C ABI: RCCL directly, no
// torch.  One process (or host thread) per GPU owns a context and a communicator; after a batch every rank hands its
// per-frame counts and fixed-size slabs of keypoints / descriptors to the root with grouped ncclSend / ncclRecv on the
// communicator's own stream (beside the next batch's kernels).  There is no collective inside detect + describe (frames are independent units,
// brisk/src/brisk-feature-detector.cc:77-85); this is the only exchange step of the path.
//
// librccl is opened at run time (dlopen at the first communicator call): the engine itself neither needs nor links it,
// and in a process that already carries an RCCL (torch ships one under the same SONAME) the loader hands back that copy
// instead of a second one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
// The RCCL entry points are looked up at run time; their declarations come from the RCCL header where the ROCm install
// has it and from the (NCCL 2.7+, stable) ABI subset below where it does not - the engine builds either way and answers
// BRISK_HIP_ERR_UNSUPPORTED when no librccl can be opened (round-4 advisor finding: a missing header broke the whole
// library's build, single-GPU users included).
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define BRISK_HAVE_RCCL_HEADER 1
#endif
#endif
#ifndef BRISK_HAVE_RCCL_HEADER
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
}
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>

#include "../../include/brisk_hip.h"
#include "brisk_common.h"
#include "brisk_kernels.h"

static_assert(BRISK_HIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace {
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
};

RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // BRISK_HIP_RCCL_LIB: the library to open instead of the default names (a site's own RCCL build; the test suite's
    // socket-based double, tests/cpp/fake_rccl.cc, which lets several ranks share the one GPU of a test box)
    const char* own = getenv("BRISK_HIP_RCCL_LIB");
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    if (own && own[0]) {
      api.lib = dlopen(own, RTLD_NOW | RTLD_LOCAL);
    } else {
      for (const char* n : names) {
        api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
      }
    }
    if (!api.lib) { api.err = std::string("librccl not found: ") + dlerror(); return; }
#define BRISK_RCCL_SYM(field, name)                                             \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name));      \
  if (!api.field) { api.err = std::string("librccl lacks ") + name; api.lib = nullptr; return; }
    BRISK_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    BRISK_RCCL_SYM(CommInitRank, "ncclCommInitRank")
    BRISK_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    BRISK_RCCL_SYM(GroupStart, "ncclGroupStart")
    BRISK_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    BRISK_RCCL_SYM(Send, "ncclSend")
    BRISK_RCCL_SYM(Recv, "ncclRecv")
    BRISK_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef BRISK_RCCL_SYM
  });
  return api.lib ? &api : nullptr;
}
}  // namespace

struct brisk_hip_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  // The transfers run on the communicator's own stream, beside the next batch's kernels (xGMI is point-to-point: ~22 MB
  // per rank and 256-frame batch take ~0.15 ms of one link against milliseconds of compute - off the critical path as
  // long as nothing waits for them).  Two slots of send slabs: the engine's result buffers are overwritten by the next
  // batch, so a batch's rows are packed into a slab on the batch's stream first; a slab is reused only after its
  // previous transfer has finished (sent[]).
  hipStream_t cs = nullptr;
  hipEvent_t packed[2] = {nullptr, nullptr}, sent[2] = {nullptr, nullptr};
  bool sent_valid[2] = {false, false};
  unsigned calls = 0;
  int* s_counts[2] = {nullptr, nullptr};
  uint8_t* s_kps[2] = {nullptr, nullptr};
  uint8_t* s_desc[2] = {nullptr, nullptr};
  size_t cap_counts = 0, cap_kps = 0, cap_desc = 0;
  std::string err;
};

// slabs out of the result buffers: counts[f] (0 for the frames this rank does not own), keypoints / descriptors of the
// first kpad rows of every frame, packed [frames_max][kpad][28] / [frames_max][kpad][strings]
__global__ void __launch_bounds__(256) k_comm_pack(const BriskFrameCounters* __restrict__ counters, const BriskKeyPoint* __restrict__ dkp,
                                                    const uint8_t* __restrict__ desc, int kp_cap, int desc_pitch, int nframes,
                                                    int frames_max, int kpad, int strings, int* __restrict__ s_counts,
                                                    uint32_t* __restrict__ s_kps, uint32_t* __restrict__ s_desc) {
  const int f = blockIdx.y;
  const int n = f < nframes ? min(counters[f].ndesc, kpad) : 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) s_counts[f] = f < nframes ? counters[f].ndesc : 0;
  const int kw = 7, dw = strings / 4;  // dwords per row
  const uint32_t* src_k = reinterpret_cast<const uint32_t*>(dkp + (long)f * kp_cap);
  uint32_t* dst_k = s_kps + (long)f * kpad * kw;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * kw; i += gridDim.x * blockDim.x) dst_k[i] = src_k[i];
  uint32_t* dst_d = s_desc + (long)f * kpad * dw;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * dw; i += gridDim.x * blockDim.x) {
    const int row = i / dw, c = i - row * dw;
    dst_d[i] = *reinterpret_cast<const uint32_t*>(desc + ((long)f * kp_cap + row) * desc_pitch + 4 * c);
  }
}

// what brisk_capi.hip knows about the last batch of a context (declared there)
int brisk_hip_internal_batch_view(brisk_hip_ctx* ctx, const BriskFrameCounters** counters, const BriskKeyPoint** dkp,
                                  const uint8_t** desc, int* kp_cap, int* desc_pitch, int* nframes, int* device, hipStream_t* stream);
int brisk_hip_internal_fail(brisk_hip_ctx* ctx, int code, const char* msg);

static int comm_fail(brisk_hip_ctx* ctx, brisk_hip_comm* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return ctx ? brisk_hip_internal_fail(ctx, code, msg.c_str()) : code;
}

extern "C" {

int brisk_hip_comm_unique_id(uint8_t* id) {
  if (!id) return BRISK_HIP_ERR_ARG;
  RcclApi* R = rccl();
  if (!R) return BRISK_HIP_ERR_UNSUPPORTED;
  ncclUniqueId u;
  if (R->GetUniqueId(&u) != ncclSuccess) return BRISK_HIP_ERR_HIP;
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return BRISK_HIP_OK;
}

int brisk_hip_comm_create(brisk_hip_ctx* ctx, int rank, int world, const uint8_t* id, brisk_hip_comm** out) {
  if (!ctx || !out || !id || world < 1 || rank < 0 || rank >= world) return BRISK_HIP_ERR_ARG;
  *out = nullptr;
  RcclApi* R = rccl();
  if (!R) return brisk_hip_internal_fail(ctx, BRISK_HIP_ERR_UNSUPPORTED, "librccl is not available in this process");
  int device = 0;
  if (brisk_hip_internal_batch_view(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &device, nullptr) != BRISK_HIP_OK)
    return BRISK_HIP_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return brisk_hip_internal_fail(ctx, BRISK_HIP_ERR_HIP, "hipSetDevice failed");
  brisk_hip_comm* c = new brisk_hip_comm();
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  const ncclResult_t rc = R->CommInitRank(&c->comm, world, u, rank);
  if (rc != ncclSuccess) {
    const std::string msg = std::string("ncclCommInitRank: ") + R->GetErrorString(rc);
    delete c;
    return brisk_hip_internal_fail(ctx, BRISK_HIP_ERR_HIP, msg.c_str());
  }
  bool ok = hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < 2 && ok; ++i)
    ok = hipEventCreateWithFlags(&c->packed[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->sent[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    brisk_hip_comm_destroy(c);
    return brisk_hip_internal_fail(ctx, BRISK_HIP_ERR_HIP, "stream / event creation for the communicator failed");
  }
  *out = c;
  return BRISK_HIP_OK;
}

void brisk_hip_comm_destroy(brisk_hip_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  RcclApi* R = rccl();
  if (c->cs) (void)hipStreamSynchronize(c->cs);
  if (R && c->comm) (void)R->CommDestroy(c->comm);
  for (int i = 0; i < 2; ++i) {
    (void)hipFree(c->s_counts[i]); (void)hipFree(c->s_kps[i]); (void)hipFree(c->s_desc[i]);
    if (c->packed[i]) (void)hipEventDestroy(c->packed[i]);
    if (c->sent[i]) (void)hipEventDestroy(c->sent[i]);
  }
  if (c->cs) (void)hipStreamDestroy(c->cs);
  delete c;
}

int brisk_hip_comm_gather_results(brisk_hip_ctx* ctx, brisk_hip_comm* c, int root, int frames_max, int kpad, int strings,
                                  int* d_counts, brisk_hip_keypoint* d_kps, uint8_t* d_desc, void* stream) {
  if (!ctx || !c) return BRISK_HIP_ERR_ARG;
  RcclApi* R = rccl();
  if (!R) return comm_fail(ctx, c, BRISK_HIP_ERR_UNSUPPORTED, "librccl is not available in this process");
  const BriskFrameCounters* counters = nullptr;
  const BriskKeyPoint* dkp = nullptr;
  const uint8_t* desc = nullptr;
  int kp_cap = 0, desc_pitch = 0, nframes = 0, device = 0;
  hipStream_t own = nullptr;
  int rc = brisk_hip_internal_batch_view(ctx, &counters, &dkp, &desc, &kp_cap, &desc_pitch, &nframes, &device, &own);
  if (rc) return rc;
  if (device != c->device) return comm_fail(ctx, c, BRISK_HIP_ERR_ARG, "communicator and context live on different devices");
  if (root < 0 || root >= c->world || frames_max < nframes || frames_max < 1 || kpad < 1 || kpad > kp_cap || strings < 4 ||
      strings % 4 || strings > desc_pitch)
    return comm_fail(ctx, c, BRISK_HIP_ERR_ARG, "gather: need frames_max >= frames of the last batch, 1 <= kpad <= keypoint capacity, strings = descriptor bytes");
  if (c->rank == root && (!d_counts || !d_kps || !d_desc)) return comm_fail(ctx, c, BRISK_HIP_ERR_ARG, "gather: the root needs destination buffers");
  if (!counters || !dkp || !desc) return comm_fail(ctx, c, BRISK_HIP_ERR_ARG, "gather: no described batch on this context");
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : own;
  if (hipSetDevice(device) != hipSuccess) return comm_fail(ctx, c, BRISK_HIP_ERR_HIP, "hipSetDevice failed");
  const size_t nb_c = (size_t)frames_max * sizeof(int), nb_k = (size_t)frames_max * kpad * sizeof(BriskKeyPoint),
               nb_d = (size_t)frames_max * kpad * strings;
  // send slabs (grown lazily; a reallocation waits for the transfers of earlier calls)
  if (nb_c > c->cap_counts || nb_k > c->cap_kps || nb_d > c->cap_desc) {
    if (hipDeviceSynchronize() != hipSuccess) return comm_fail(ctx, c, BRISK_HIP_ERR_HIP, "hipDeviceSynchronize failed");
    c->cap_counts = c->cap_kps = c->cap_desc = 0;
    for (int i = 0; i < 2; ++i) {
      (void)hipFree(c->s_counts[i]); (void)hipFree(c->s_kps[i]); (void)hipFree(c->s_desc[i]);
      c->s_counts[i] = nullptr; c->s_kps[i] = nullptr; c->s_desc[i] = nullptr;
      c->sent_valid[i] = false;
      if (hipMalloc(&c->s_counts[i], nb_c) != hipSuccess || hipMalloc(&c->s_kps[i], nb_k) != hipSuccess || hipMalloc(&c->s_desc[i], nb_d) != hipSuccess)
        return comm_fail(ctx, c, BRISK_HIP_ERR_HIP, "hipMalloc of the gather slabs failed");
    }
    c->cap_counts = nb_c; c->cap_kps = nb_k; c->cap_desc = nb_d;
  }
  const int j = (int)(c->calls & 1u);
  // the slab's previous transfer must be over before it is packed again; then pack on the batch's stream (behind the
  // batch, in front of the next one) and hand over to the communicator's stream
  if (c->sent_valid[j] && hipStreamWaitEvent(s, c->sent[j], 0) != hipSuccess) return comm_fail(ctx, c, BRISK_HIP_ERR_HIP, "hipStreamWaitEvent failed");
  // rows beyond a frame's count are never read by the root (it slices with the counts): only the counts are complete
  hipLaunchKernelGGL(k_comm_pack, dim3(8, frames_max), dim3(256), 0, s, counters, dkp, desc, kp_cap, desc_pitch, nframes, frames_max,
                     kpad, strings, c->s_counts[j], reinterpret_cast<uint32_t*>(c->s_kps[j]), reinterpret_cast<uint32_t*>(c->s_desc[j]));
  if (hipGetLastError() != hipSuccess) return comm_fail(ctx, c, BRISK_HIP_ERR_HIP, "k_comm_pack launch failed");
  if (hipEventRecord(c->packed[j], s) != hipSuccess || hipStreamWaitEvent(c->cs, c->packed[j], 0) != hipSuccess)
    return comm_fail(ctx, c, BRISK_HIP_ERR_HIP, "event hand-over to the communicator's stream failed");
  ncclResult_t nr = R->GroupStart();
  if (c->rank == root) {
    for (int r = 0; r < c->world && nr == ncclSuccess; ++r) {
      int* dc = d_counts + (size_t)r * frames_max;
      uint8_t* dk = reinterpret_cast<uint8_t*>(d_kps) + (size_t)r * nb_k;
      uint8_t* dd = d_desc + (size_t)r * nb_d;
      if (r == root) {
        if (hipMemcpyAsync(dc, c->s_counts[j], nb_c, hipMemcpyDeviceToDevice, c->cs) != hipSuccess ||
            hipMemcpyAsync(dk, c->s_kps[j], nb_k, hipMemcpyDeviceToDevice, c->cs) != hipSuccess ||
            hipMemcpyAsync(dd, c->s_desc[j], nb_d, hipMemcpyDeviceToDevice, c->cs) != hipSuccess) {
          (void)R->GroupEnd();
          return comm_fail(ctx, c, BRISK_HIP_ERR_HIP, "gather: device copy of the root's own slabs failed");
        }
        continue;
      }
      nr = R->Recv(dc, nb_c, ncclUint8, r, c->comm, c->cs);
      if (nr == ncclSuccess) nr = R->Recv(dk, nb_k, ncclUint8, r, c->comm, c->cs);
      if (nr == ncclSuccess) nr = R->Recv(dd, nb_d, ncclUint8, r, c->comm, c->cs);
    }
  } else if (nr == ncclSuccess) {
    nr = R->Send(c->s_counts[j], nb_c, ncclUint8, root, c->comm, c->cs);
    if (nr == ncclSuccess) nr = R->Send(c->s_kps[j], nb_k, ncclUint8, root, c->comm, c->cs);
    if (nr == ncclSuccess) nr = R->Send(c->s_desc[j], nb_d, ncclUint8, root, c->comm, c->cs);
  }
  const ncclResult_t ne = R->GroupEnd();
  if (nr == ncclSuccess) nr = ne;
  if (nr != ncclSuccess) return comm_fail(ctx, c, BRISK_HIP_ERR_HIP, std::string("RCCL: ") + R->GetErrorString(nr));
  if (hipEventRecord(c->sent[j], c->cs) != hipSuccess) return comm_fail(ctx, c, BRISK_HIP_ERR_HIP, "hipEventRecord failed");
  c->sent_valid[j] = true;
  c->calls++;
  return BRISK_HIP_OK;
}

int brisk_hip_comm_wait(brisk_hip_comm* c, void* stream) {
  if (!c) return BRISK_HIP_ERR_ARG;
  if (hipSetDevice(c->device) != hipSuccess) return BRISK_HIP_ERR_HIP;
  for (int i = 0; i < 2; ++i) {
    if (!c->sent_valid[i]) continue;
    const hipError_t e = stream ? hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->sent[i], 0) : hipEventSynchronize(c->sent[i]);
    if (e != hipSuccess) { c->err = std::string("brisk_hip_comm_wait: ") + hipGetErrorString(e); return BRISK_HIP_ERR_HIP; }
  }
  return BRISK_HIP_OK;
}

int brisk_hip_comm_rank(const brisk_hip_comm* c) { return c ? c->rank : -1; }
int brisk_hip_comm_world(const brisk_hip_comm* c) { return c ? c->world : 0; }

}  // extern "C"
