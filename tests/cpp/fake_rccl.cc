// fake_rccl.cc - TEST DOUBLE of the eight RCCL entry points csrc/brisk_comm.hip uses, over Unix-domain sockets and staged
// hipMemcpy: lets a world of several ranks run on the ONE GPU of a test box (RCCL itself refuses two ranks on one
// device), so that brisk_hip_comm_gather_results' peer branches - every ncclSend, every ncclRecv into rank r's slab -
// execute before the first real multi-GPU run.  Selected with BRISK_HIP_RCCL_LIB=<this library>.  It measures nothing
// and pins nothing about RCCL: point-to-point messages between two ranks arrive in the order they were sent, an operation
// is complete (host-blocking) when ncclGroupEnd / the ungrouped call returns.
// Built by tests/test_cpp_classes.py: g++ -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.cc -lamdhip64.
#include <errno.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
}

struct ncclComm {
  int rank = 0, world = 1;
  std::vector<int> fd;  // socket to every peer (-1 for self)
  int listen_fd = -1;
  std::string path;
};

namespace {
struct Op { bool send; void* buf; size_t bytes; int peer; ncclComm* comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}
bool write_all(int fd, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  while (n) {
    const ssize_t w = ::write(fd, c, n);
    if (w < 0) { if (errno == EINTR) continue; return false; }
    c += w; n -= (size_t)w;
  }
  return true;
}
bool read_all(int fd, void* p, size_t n) {
  char* c = static_cast<char*>(p);
  while (n) {
    const ssize_t r = ::read(fd, c, n);
    if (r < 0) { if (errno == EINTR) continue; return false; }
    if (r == 0) return false;
    c += r; n -= (size_t)r;
  }
  return true;
}
std::string sock_path(const ncclUniqueId& id, int rank) {
  char hex[33];
  for (int i = 0; i < 16; ++i) snprintf(hex + 2 * i, 3, "%02x", (unsigned char)id.internal[i]);
  return std::string("/tmp/brisk_fake_rccl_") + hex + "_" + std::to_string(rank);
}
ncclResult_t run_op(const Op& o) {
  if (o.peer < 0 || o.peer >= o.comm->world || o.peer == o.comm->rank) return ncclInvalidArgument;
  const int fd = o.comm->fd[(size_t)o.peer];
  std::vector<char> stage(o.bytes);
  if (o.send) {
    // stream order: what is sent is what the stream has produced by now
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (o.bytes && hipMemcpy(stage.data(), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const uint64_t n = o.bytes;
    if (!write_all(fd, &n, 8) || !write_all(fd, stage.data(), o.bytes)) return ncclSystemError;
  } else {
    uint64_t n = 0;
    if (!read_all(fd, &n, 8) || n != o.bytes) return ncclSystemError;  // (sizes of a matching send / recv pair agree)
    if (!read_all(fd, stage.data(), o.bytes)) return ncclSystemError;
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (o.bytes && hipMemcpy(o.buf, stage.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  unsigned s = (unsigned)tv.tv_usec * 2654435761u ^ (unsigned)getpid() * 40503u ^ (unsigned)tv.tv_sec;
  for (int i = 0; i < NCCL_UNIQUE_ID_BYTES; ++i) { s = s * 1664525u + 1013904223u; id->internal[i] = (char)(s >> 24); }
  return ncclSuccess;
}

// every rank listens on its own socket, connects to all lower ranks and accepts all higher ones
ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->rank = rank; c->world = nranks; c->fd.assign((size_t)nranks, -1);
  c->path = sock_path(id, rank);
  if (rank + 1 < nranks) {
    c->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un a;
    memset(&a, 0, sizeof a);
    a.sun_family = AF_UNIX;
    strncpy(a.sun_path, c->path.c_str(), sizeof a.sun_path - 1);
    unlink(c->path.c_str());
    if (c->listen_fd < 0 || bind(c->listen_fd, (sockaddr*)&a, sizeof a) != 0 || listen(c->listen_fd, nranks) != 0) { delete c; return ncclSystemError; }
  }
  for (int r = 0; r < rank; ++r) {
    const std::string p = sock_path(id, r);
    int fd = -1;
    for (int tries = 0; tries < 3000; ++tries) {  // (the lower rank may not be listening yet: up to 60 s)
      fd = socket(AF_UNIX, SOCK_STREAM, 0);
      sockaddr_un a;
      memset(&a, 0, sizeof a);
      a.sun_family = AF_UNIX;
      strncpy(a.sun_path, p.c_str(), sizeof a.sun_path - 1);
      if (fd >= 0 && connect(fd, (sockaddr*)&a, sizeof a) == 0) break;
      if (fd >= 0) close(fd);
      fd = -1;
      struct timespec ts = {0, 20 * 1000 * 1000};
      nanosleep(&ts, nullptr);
    }
    if (fd < 0) { delete c; return ncclSystemError; }
    const int32_t me = rank;
    if (!write_all(fd, &me, 4)) { delete c; return ncclSystemError; }
    c->fd[(size_t)r] = fd;
  }
  for (int k = rank + 1; k < nranks; ++k) {
    const int fd = accept(c->listen_fd, nullptr, nullptr);
    int32_t who = -1;
    if (fd < 0 || !read_all(fd, &who, 4) || who <= rank || who >= nranks || c->fd[(size_t)who] != -1) { delete c; return ncclSystemError; }
    c->fd[(size_t)who] = fd;
  }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclInvalidArgument;
  for (int fd : c->fd) if (fd >= 0) close(fd);
  if (c->listen_fd >= 0) { close(c->listen_fd); unlink(c->path.c_str()); }
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd(void) {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  // sends first (a socket buffers little: the receiving side of this double only ever receives inside a group that sends
  // nothing - the gather's pattern; a true exchange between two ranks would need a sender thread)
  ncclResult_t rc = ncclSuccess;
  for (int pass = 0; pass < 2 && rc == ncclSuccess; ++pass)
    for (const Op& o : g_ops)
      if (o.send == (pass == 0) && rc == ncclSuccess) rc = run_op(o);
  g_ops.clear();
  return rc;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  if (!c || (!buf && count)) return ncclInvalidArgument;
  const Op o{true, const_cast<void*>(buf), count * type_size(t), peer, c, s};
  if (g_depth > 0) { g_ops.push_back(o); return ncclSuccess; }
  return run_op(o);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  if (!c || (!buf && count)) return ncclInvalidArgument;
  const Op o{false, buf, count * type_size(t), peer, c, s};
  if (g_depth > 0) { g_ops.push_back(o); return ncclSuccess; }
  return run_op(o);
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake rccl)";
    case ncclSystemError: return "socket error (fake rccl)";
    case ncclInvalidArgument: return "invalid argument (fake rccl)";
    case ncclInvalidUsage: return "invalid usage (fake rccl)";
    default: return "internal error (fake rccl)";
  }
}

}  // extern "C"
