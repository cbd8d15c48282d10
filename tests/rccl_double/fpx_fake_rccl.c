/* fpx_fake_rccl.c -- TEST DOUBLE of the RCCL entry points libfpx binds at run time (fpx_api.hip rccl_bind: ncclGetUniqueId,
 * ncclCommInitRank, ncclCommDestroy, ncclReduceScatter, ncclAllGather, ncclAllReduce, ncclGroupStart / End, ncclGetErrorString), selected with
 * FPX_RCCL_LIB.  Test infrastructure only.
 *
 * Why: fpx_phase2_replica_sharded_dev (K1 on my acceptor columns -> reduce-scatter of the partial vote bitmaps ->
 * all-reduce(max) of the Nack rounds -> open + K2 on my slots) had only ever run with a world of ONE rank: the GPU boxes
 * of this build have one GPU, and RCCL refuses two ranks on one device.  With this double the SHIPPED code path runs
 * with world = 2 -- two processes, two libfpx contexts on the one GPU -- before the first multi-GPU node does.
 *
 * How: ranks are processes; a collective is stream-synchronise, device -> shared memory, barrier, reduce on the host,
 * host -> device, barrier.  Same call signatures and value conventions as rccl.h (NCCL 2.x ABI): ncclUniqueId is 128
 * opaque bytes (here: the name of a POSIX shared-memory segment), ncclSum = 0, ncclMax = 2, ncclUint8 = 1, ncclInt32 = 2,
 * ncclUint64 = 5, ncclSuccess = 0. */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
enum { SEG_BYTES = 32 << 20, HDR_BYTES = 4096, MAX_WORLD = 8 };

struct hdr {
  volatile uint32_t count, gen;
};
struct comm {
  int rank, world;
  unsigned char* base;
  size_t map_bytes;
  void* host;  /* this rank's reduction buffer */
};

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

static int barrier(struct comm* c) {
  struct hdr* h = (struct hdr*)c->base;
  const uint32_t gen = __atomic_load_n(&h->gen, __ATOMIC_ACQUIRE);
  if (__atomic_add_fetch(&h->count, 1, __ATOMIC_ACQ_REL) == (uint32_t)c->world) {
    __atomic_store_n(&h->count, 0, __ATOMIC_RELAXED);
    __atomic_add_fetch(&h->gen, 1, __ATOMIC_RELEASE);
    return 0;
  }
  const double t0 = now();
  while (__atomic_load_n(&h->gen, __ATOMIC_ACQUIRE) == gen) {
    sched_yield();
    if (now() - t0 > 120.0) return 6; /* ncclRemoteError: a rank never arrived */
  }
  return 0;
}

static unsigned char* seg(struct comm* c, int r) { return c->base + HDR_BYTES + (size_t)r * SEG_BYTES; }

static size_t type_size(int dtype) {
  switch (dtype) {
    case 1: return 1;  /* ncclUint8 */
    case 2: return 4;  /* ncclInt32 */
    case 5: return 8;  /* ncclUint64 */
    default: return 0;
  }
}

int ncclGetUniqueId(ncclUniqueId* id) {
  static unsigned counter;
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/fpxrccl_%d_%u_%llx", (int)getpid(), ++counter, (unsigned long long)(now() * 1e9));
  return 0;
}

int ncclCommInitRank(void** out, int world, ncclUniqueId id, int rank) {
  if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world || id.internal[0] != '/') return 4; /* ncclInvalidArgument */
  id.internal[127] = 0;
  struct comm* c = (struct comm*)calloc(1, sizeof(*c));
  if (!c) return 2;
  c->rank = rank, c->world = world, c->map_bytes = HDR_BYTES + (size_t)world * SEG_BYTES;
  const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) return 2; /* ncclSystemError */
  c->base = (unsigned char*)mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == MAP_FAILED) return 2;
  c->host = malloc(SEG_BYTES);
  if (!c->host) return 2;
  int rc = barrier(c); /* everybody is attached (a fresh segment reads zero: count = gen = 0) */
  if (rc) return rc;
  if (rank == 0) shm_unlink(id.internal);
  *out = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  struct comm* c = (struct comm*)comm;
  if (!c) return 4;
  munmap(c->base, c->map_bytes);
  free(c->host);
  free(c);
  return 0;
}

/* groups: this double runs every collective inside its call, so a group only checks that it is opened and closed in pairs */
static int group_depth, groups_closed;
int ncclGroupStart(void) { return ++group_depth > 8 ? 5 /* ncclInvalidUsage */ : 0; }
int ncclGroupEnd(void) {
  if (group_depth <= 0) return 5;
  --group_depth, ++groups_closed;
  return 0;
}
int fpx_fake_rccl_groups_closed(void) { return groups_closed; } /* for the tests */

const char* ncclGetErrorString(int code) { return code == 0 ? "no error" : "fpx_fake_rccl: error"; }

/* every rank's send buffer into its segment; returns after the barrier */
static int publish(struct comm* c, const void* send, size_t bytes, hipStream_t stream) {
  if (bytes > SEG_BYTES) return 4;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1; /* ncclUnhandledCudaError */
  if (hipMemcpy(seg(c, c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  return barrier(c);
}

static void reduce(void* acc, const void* x, size_t count, int dtype, int op) {
  if (dtype == 5 && op == 0) {
    for (size_t i = 0; i < count; ++i) ((uint64_t*)acc)[i] += ((const uint64_t*)x)[i];
  } else if (dtype == 2 && op == 2) {
    for (size_t i = 0; i < count; ++i)
      if (((const int32_t*)x)[i] > ((int32_t*)acc)[i]) ((int32_t*)acc)[i] = ((const int32_t*)x)[i];
  } else if (dtype == 2 && op == 0) {
    for (size_t i = 0; i < count; ++i) ((int32_t*)acc)[i] += ((const int32_t*)x)[i];
  }
}

int ncclReduceScatter(const void* send, void* recv, size_t recvcount, int dtype, int op, void* comm, hipStream_t stream) {
  struct comm* c = (struct comm*)comm;
  const size_t ts = type_size(dtype);
  if (!c || !ts || !((dtype == 5 && op == 0) || (dtype == 2 && (op == 0 || op == 2)))) return 4;
  int rc = publish(c, send, recvcount * c->world * ts, stream);
  if (rc) return rc;
  memcpy(c->host, seg(c, 0) + (size_t)c->rank * recvcount * ts, recvcount * ts);
  for (int r = 1; r < c->world; ++r) reduce(c->host, seg(c, r) + (size_t)c->rank * recvcount * ts, recvcount, dtype, op);
  if (hipMemcpy(recv, c->host, recvcount * ts, hipMemcpyHostToDevice) != hipSuccess) return 1;
  return barrier(c); /* nobody overwrites its segment before everybody has read it */
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
  struct comm* c = (struct comm*)comm;
  const size_t ts = type_size(dtype);
  if (!c || !ts || !((dtype == 5 && op == 0) || (dtype == 2 && (op == 0 || op == 2)))) return 4;
  int rc = publish(c, send, count * ts, stream);
  if (rc) return rc;
  memcpy(c->host, seg(c, 0), count * ts);
  for (int r = 1; r < c->world; ++r) reduce(c->host, seg(c, r), count, dtype, op);
  if (hipMemcpy(recv, c->host, count * ts, hipMemcpyHostToDevice) != hipSuccess) return 1;
  return barrier(c);
}

int ncclAllGather(const void* send, void* recv, size_t sendcount, int dtype, void* comm, hipStream_t stream) {
  struct comm* c = (struct comm*)comm;
  const size_t ts = type_size(dtype);
  if (!c || !ts) return 4;
  int rc = publish(c, send, sendcount * ts, stream);
  if (rc) return rc;
  for (int r = 0; r < c->world; ++r)
    if (hipMemcpy((unsigned char*)recv + (size_t)r * sendcount * ts, seg(c, r), sendcount * ts, hipMemcpyHostToDevice) != hipSuccess) return 1;
  return barrier(c);
}
